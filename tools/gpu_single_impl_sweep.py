"""One configs[2] scene: rollout latency per fused-chain implementation and rows per workgroup (graph replay, median of 20)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
scene = synth.baseline_scene(spec, 2, seed=0, batch=int(os.environ.get("PS_SCENES", "1")))
ref = None
for impl, rows in ((1, 0), (2, 1), (2, 2), (2, 4), (2, 8), (3, 1), (3, 2), (3, 4)):
    eng = Engine(spec, w)
    eng.set_chain_impl(impl)
    eng.set_chain_rows(rows)
    eng.set_scene(scene)
    for _ in range(3):
        eng.rollout(); eng.sync()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); eng.rollout(); eng.sync(); ts.append(time.perf_counter() - t0)
    traj = eng.get("traj")
    if ref is None: ref = traj
    print(f"impl {impl} rows {rows}: rollout {1e3 * np.median(ts):.3f} ms  nodes {eng.graph_nodes}  max diff vs impl 1: {np.abs(traj - ref).max():.2e}", flush=True)
    eng.close()
