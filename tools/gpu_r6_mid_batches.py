import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
for S in (10, 12, 16, 20):
    parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(S)]
    scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
                 {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
    eng = Engine(spec, w)
    ref = None
    for impl, rows in ((0, 0),):
        eng.set_chain_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
        t = eng.time_rollout(3, 20)
        tr = eng.padded("traj").copy()
        if ref is None: ref = tr
        print(f"scenes {S} ({S*128} agents) impl {impl} rows {rows:2d}: rollout {t if not hasattr(t,'__len__') else t[0]:.3f} ms  policy launch {eng.time_policy_kernel(3):.4f} ms  max |traj - rows0| {np.abs(tr-ref).max():.2e}", flush=True)
    eng.close()
