"""Parity of the exact bench.py workload (8 x configs[2] scenes, seeds 0..7, demo model) in both engine modes against the
fp64 oracle: replan 0 and per-agent closed loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
torch.set_num_threads(32)
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
t0 = time.time()
with torch.no_grad():
    o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
print("oracle fp64: %.0f s" % (time.time() - t0), flush=True)
eng = Engine(spec, w)
pm = scene["prompt_mask"].astype(bool)
for rows in (0, 4):
    eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout()
    A = eng.num_agents
    mp = eng.get("motion_pred")
    d = np.abs(eng.padded("traj") - o64["traj"].numpy())[pm].reshape(A, -1).max(1)
    print("chain_rows %d: replan-0 max %.2e | closed loop per agent: median %.2e, within 1e-4: %.4f, max %.2e" %
          (rows, np.abs(mp[0] - o64["motion_pred"][:A].numpy()).max(), np.median(d), (d < 1e-4).mean(), d.max()), flush=True)
eng.close()
