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
# degrees of the first policy layer's set (a2p), for patterns in what differs
ge = eng.lib.ps_test_get_edges
ge.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]; ge.restype = ctypes.c_int64
_es, _ed = np.empty(1 << 22, np.int32), np.empty(1 << 22, np.int32)
_n = ge(eng.h, 4, _es.ctypes.data, _ed.ctypes.data, None, 1 << 22)
deg = np.bincount(_ed[:_n], minlength=rows)[:rows]
eng.close()
names = ["agg", "u", "q", "s", "g", "fold"]
if sys.argv[1] == "save":
    np.save(sys.argv[2], out); np.save(sys.argv[2] + ".emd.npy", emd)
    print("saved; nonzero fractions", [(n, float((out[i] != 0).mean())) for i, n in enumerate(names)])
else:
    ref = np.load(sys.argv[2]); remd = np.load(sys.argv[2] + ".emd.npy")
    print("policy_emd max diff", float(np.abs(emd - remd).max()))
    badrow = (np.abs(out[5] - ref[5]).max(axis=1) > 0)
    tiles = (deg + 15) // 16
    print("fold plane: bad rows by tile count", {int(t): (int((badrow & (tiles == t)).sum()), int((tiles == t).sum())) for t in np.unique(tiles)})
    # position in the workgroup's queue (rows by falling degree, ties by index)
    pos = np.zeros(rows, int)
    for g in range(rows // 16):
        d = deg[16 * g:16 * g + 16]
        order = sorted(range(16), key=lambda i: (-d[i], i))
        for p_, i in enumerate(order):
            pos[16 * g + i] = p_
    print("fold plane: bad rows by queue position", [(int((badrow & (pos == p_)).sum()), int((pos == p_).sum())) for p_ in range(16)])
    print("fold plane: bad rows by (tiles odd?)", [(int((badrow & ((tiles & 1) == o)).sum()), int(((tiles & 1) == o).sum())) for o in (0, 1)])
    for i, n in enumerate(names):
        d = np.abs(out[i] - ref[i])
        bad = np.argwhere(d > 0)
        if bad.size:
            rows_b, cols_b = bad[:, 0], bad[:, 1]
            print(f"        rows differing {np.unique(rows_b).size} (rows % 16 histogram {np.bincount(rows_b % 16, minlength=16).tolist()}); heads histogram {np.bincount(cols_b // 16, minlength=8).tolist()}; "
                  f"columns % 16 histogram {np.bincount(cols_b % 16, minlength=16).tolist()}")
        print(f"{n:5s}: differing {int((d > 0).sum())} of {d.size}; max {d.max():.3e}; max |ref| {np.abs(ref[i]).max():.3e}; columns of the first differing row {np.nonzero(d[np.argmax(d.max(axis=1) > 0)] > 0)[0][:8].tolist()}")
