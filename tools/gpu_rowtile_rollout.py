"""The row-tile kernels inside whole rollouts (8 x configs[2] benchmark batch and one scene): scene tokens / trajectories of
ps_set_row_impl 0 (default), 11..13 (node halves forced to 1..3 row tiles per wave) against impl 1 (the staged round-3 kernels),
timings of encode / generate / replan stages; one scene also against the fp64 oracle's map tokens (seed in argv[1], default 5)."""
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
scene8 = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
              {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
eng = Engine(spec, w)
for name, scene, rows in (("8 scenes, latency mode", scene8, 0), ("8 scenes, 16 rows", scene8, 16), ("1 scene", parts[0], 0)):
    ref_tok = ref_traj = None
    for impl in (1, 0, 11, 12, 13):
        eng.set_row_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
        tok, traj = eng.get("scene_tokens"), eng.padded("traj")
        if ref_tok is None: ref_tok, ref_traj = tok, traj
        ms, st = eng.time_rollout(1, 5)
        print(f"{name:24s} row_impl {impl:2d}: rollout {ms:7.3f} ms (enc {st[0]:.3f} gen {st[1]:.3f} loop {st[2]:.3f}) | tokens vs staged max {np.abs(tok - ref_tok).max():.2e} "
              f"traj vs staged max {np.abs(traj - ref_traj).max():.2e} finite {bool(np.isfinite(traj).all())}", flush=True)
eng.set_row_impl(0)
eng.close()
