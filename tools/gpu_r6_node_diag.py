"""First policy launch of the bench workload on k_chain16 with 16 rows per workgroup: the policy embeddings row by row against a file written by
another library (PS_LIB=... python tools/gpu_r6_node_diag.py save /tmp/ref.npy; PS_LIB=<other> python tools/gpu_r6_node_diag.py cmp /tmp/ref.npy)."""
import os, sys
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
eng.set_chain_impl(0); eng.set_chain_rows(int(os.environ.get("PS_ROWS", "16"))); eng.set_scene(scene)
outs = []
for it in range(int(os.environ.get("PS_ITERS", "6"))):
    eng.encode_scene(); eng.generate_policy(); eng.reset_rollout(); eng.policy_step(0); eng.sync()
    outs.append(eng.get("policy_emd").copy())
eng.close()
if sys.argv[1] == "save":
    np.save(sys.argv[2], outs[0])
    print("saved", outs[0].shape, "repeats equal:", all(np.array_equal(o, outs[0]) for o in outs))
else:
    ref = np.load(sys.argv[2])
    for it, o in enumerate(outs):
        d = np.abs(o.reshape(ref.shape[0], -1) - ref.reshape(ref.shape[0], -1)).max(axis=1)
        bad = np.nonzero(d > 0)[0]
        print(f"run {it}: {bad.size} of {d.size} rows differ; max {d.max():.3e}; rows % 16 histogram {np.bincount(bad % 16, minlength=16).tolist()}; "
              f"workgroups touched {np.unique(bad // 16).size}; first rows {bad[:12].tolist()}")
