"""encode_scene of the 8-scene benchmark batch a few times, latency mode (the s2s layers are split layers: k_node_pre_rt, k_edge_rows,
k_node_post_rt) -- the workload of a kernel trace / PMC pass of those kernels.  PS_ROW_IMPL: ps_set_row_impl."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
eng = Engine(spec, weights.init_weights(spec, 0))
eng.set_row_impl(int(os.environ.get("PS_ROW_IMPL", "0")))
eng.set_scene(scene)
for _ in range(int(os.environ.get("PS_ITERS", "4"))):
    eng.encode_scene(); eng.sync()
eng.close()
