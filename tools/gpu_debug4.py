import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
for seed in (0, 1, 2):
    scene = synth.baseline_scene(spec, 4, seed=seed)
    eng = Engine(spec, w); eng.set_scene(scene)
    with torch.no_grad():
        o = orc.rollout(w, spec, scene); o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
    eng.encode_scene(); eng.generate_policy(); eng.reset_rollout()
    A = eng.num_agents
    for t in range(8):
        eng.policy_step(t)
        mp = eng.get("motion_pred")[t]
        ec = eng.get("edge_counts")[4:6]
        e = np.abs(mp - o64["motion_pred"][t*A:(t+1)*A].numpy())
        e32 = np.abs(o["motion_pred"][t*A:(t+1)*A].numpy() - o64["motion_pred"][t*A:(t+1)*A].numpy())
        print(seed, t, "edges hip", ec, "o32", o["step_edges"][t], "o64", o64["step_edges"][t], "mp err hip %.2e o32 %.2e" % (e.max(), e32.max()),
              "worst agent", int(e.reshape(A,-1).max(1).argmax()), flush=True)
    eng.close()
