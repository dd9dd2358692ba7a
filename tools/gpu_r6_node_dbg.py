"""First policy launch of the bench workload (k_chain16, 16 rows per workgroup): six first-layer intermediates of the node phase (agg, u, q, s, g, fold)
from a -DPS_C16_DBG experiments build -- python tools/gpu_r6_node_dbg.py save|cmp file.npy with PS_LIB set."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
eng = Engine(spec, w)
eng.set_chain_impl(0); eng.set_chain_rows(16); eng.set_scene(scene)
eng.encode_scene(); eng.generate_policy(); eng.reset_rollout(); eng.sync()
rows = 1024
f = eng.lib.ps_test_c16_dbg
f.argtypes = [ctypes.c_void_p, ctypes.c_int]; f.restype = ctypes.c_int
assert f(None, rows) == 0
eng.policy_step(0); eng.sync()
out = np.zeros((6, rows, 128), np.float32)
assert f(out.ctypes.data, rows) == 0
emd = eng.get("policy_emd").copy()
eng.close()
names = ["agg", "u", "q", "s", "g", "fold"]
if sys.argv[1] == "save":
    np.save(sys.argv[2], out); np.save(sys.argv[2] + ".emd.npy", emd)
    print("saved; nonzero fractions", [(n, float((out[i] != 0).mean())) for i, n in enumerate(names)])
else:
    ref = np.load(sys.argv[2]); remd = np.load(sys.argv[2] + ".emd.npy")
    print("policy_emd max diff", float(np.abs(emd - remd).max()))
    for i, n in enumerate(names):
        d = np.abs(out[i] - ref[i])
        print(f"{n:5s}: differing {int((d > 0).sum())} of {d.size}; max {d.max():.3e}; max |ref| {np.abs(ref[i]).max():.3e}; columns of the first differing row {np.nonzero(d[np.argmax(d.max(axis=1) > 0)] > 0)[0][:8].tolist()}")
