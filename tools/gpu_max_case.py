import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
torch.set_num_threads(32)
spec = SMALL_SPEC
w = weights.init_weights(spec, 0)
eng = Engine(spec, w)
for (na, nm, pts, B) in ((512, 2048, 32, 1), (400, 2160, 32, 2)):
    scene = synth.make_scene(spec, na, nm, batch=B, seed=77, goal=True, points=pts, ragged=(B > 1), square=400.0)
    t0 = time.time()
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
        o32 = orc.rollout(w, spec, scene)
    t1 = time.time()
    eng.set_scene(scene); eng.rollout()
    A = eng.num_agents
    mp = eng.get("motion_pred")
    pm = scene["prompt_mask"].astype(bool)
    d = np.abs(eng.padded("traj") - o64["traj"].numpy())[pm].reshape(A, -1).max(1)
    floor = np.abs(o32["traj"].numpy() - o64["traj"].numpy())[pm].max()
    print((na, nm, pts, B), "oracle %.1fs" % (t1 - t0), "replan0 %.2e" % np.abs(mp[0] - o64["motion_pred"][:A].numpy()).max(), "traj max %.2e floor %.2e frac<1e-4 %.3f" % (d.max(), floor, (d < 1e-4).mean()), "edges", eng.get("edge_counts"), flush=True)
try:
    eng.set_scene(synth.make_scene(spec, 513, 2048, batch=1, seed=1))
    print("NO ERROR for 2561 tokens")
except RuntimeError as ex:
    print("2561 tokens:", ex)
eng.close()
