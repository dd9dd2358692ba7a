"""k_edge_rows (one wave per row, 12 waves per CU) against k_edge16 (16-row workgroups) in the split attention layers: bit-equality of a
whole rollout of the 8-scene benchmark batch (latency mode: the scene encoder's s2s layers are split layers), rollout / encoding time."""
import os, sys, time
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
ref = None
for impl in (2, 0):
    eng = Engine(spec, w)
    eng.set_row_impl(impl)
    eng.set_scene(scene)
    for _ in range(3):
        eng.rollout(); eng.sync()
    ts, te = [], []
    for _ in range(10):
        t0 = time.perf_counter(); eng.rollout(); eng.sync(); ts.append(time.perf_counter() - t0)
    for _ in range(10):
        t0 = time.perf_counter(); eng.encode_scene(); eng.sync(); te.append(time.perf_counter() - t0)
    eng.rollout(); eng.sync()
    out = {k: eng.get(k) for k in ("traj", "vel", "motion_pred")}
    if ref is None: ref = out
    print(f"row impl {impl}: rollout {1e3 * np.median(ts):.3f} ms, encode_scene {1e3 * np.median(te):.3f} ms, identical to k_edge16: "
          f"{all(np.array_equal(out[k], ref[k]) for k in out)}", flush=True)
    eng.close()
