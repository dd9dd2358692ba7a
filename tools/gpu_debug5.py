import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
scene = synth.make_scene(spec, 48, 256, batch=3, seed=11, goal=True, ragged=True)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    t0 = time.time()
    with torch.no_grad():
        o32 = orc.rollout(w, spec, scene)
    print("threads", nt, "oracle32 time", time.time() - t0, flush=True)
torch.set_num_threads(16)
with torch.no_grad():
    o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
eng = Engine(spec, w)
eng.set_scene(scene); eng.rollout()
traj = eng.padded("traj"); mp = eng.get("motion_pred")
pm = scene["prompt_mask"].astype(bool)
A = eng.num_agents
print("batch: traj err vs o64 per scene", [float(np.abs(traj[b] - o64["traj"][b].numpy()).max()) for b in range(3)],
      "o32 vs o64", [float(np.abs(o32["traj"][b].numpy() - o64["traj"][b].numpy()).max()) for b in range(3)])
for t in range(8):
    print(" replan", t, "mp err hip", float(np.abs(mp[t] - o64["motion_pred"][t*A:(t+1)*A].numpy()).max()),
          "o32", float(np.abs(o32["motion_pred"][t*A:(t+1)*A].numpy() - o64["motion_pred"][t*A:(t+1)*A].numpy()).max()))
for b in range(3):
    one = {k: (v[b:b + 1] if not isinstance(v, dict) else {kk: {k3: v3[b:b + 1] for k3, v3 in vv.items()} for kk, vv in v.items()}) for k, v in scene.items()}
    eng.set_scene(one); eng.rollout()
    print("single", b, "traj err vs o64", float(np.abs(eng.padded("traj")[0] - o64["traj"][b].numpy()).max()),
          "vs batch", float(np.abs(eng.padded("traj")[0] - traj[b]).max()))
