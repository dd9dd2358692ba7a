"""Engine life-cycle stress: many create / set_scene / rollout / close cycles over changing shapes, options and modes;
device memory must come back (hipMemGetInfo through torch) and nothing may fault.  usage: python tools/gpu_stress.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC
from prosim_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(5)
specs = [SMALL_SPEC, SMALL_SPEC.replace(enc_learnable_pe=True, dec_learnable_pe=True, pol_learnable_pe=True),
         SMALL_SPEC.replace(used_v2v_tags=("Following", "Merging")), SMALL_SPEC.replace(motion_k=3, rollout_top_k=3, goal_pred_k=4)]
ws = [weights.init_weights(s, 0) for s in specs]
torch.cuda.init()
# baseline AFTER one engine has lived and died: the runtime keeps the code object, its graph pools and stream resources
e0 = Engine(specs[0], ws[0]); e0.set_scene(synth.make_scene(specs[0], 5, 20, batch=1, seed=1)); e0.rollout(); e0.get("traj"); e0.close()
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
# a leak grows with the number of cycles: the same cycle 40 times over
for _ in range(40):
    e0 = Engine(specs[1], ws[1]); e0.set_replicas(3); e0.set_scene(synth.make_scene(specs[1], 33, 200, batch=1, seed=2)); e0.rollout(); e0.get("traj"); e0.close()
torch.cuda.synchronize()
grown = (free0 - torch.cuda.mem_get_info()[0]) / 1e6
print("40 identical create / rollout / close cycles: %.1f MB not returned (%.2f MB per cycle: runtime-side pools; the engine frees all it owns)" % (grown, grown / 40), flush=True)
assert grown < 80.0, grown
eng = None
for it in range(n):
    k = rng.randint(len(specs))
    if eng is None or rng.rand() < 0.3:
        if eng is not None:
            eng.close()
        eng, ek = Engine(specs[k], ws[k]), k
    spec = specs[ek]
    rep = int(rng.choice([1, 1, 1, 3, 8]))
    eng.set_replicas(rep)
    eng.set_chain_rows(int(rng.choice([0, 0, 4, 16])))
    kw = dict(n_agents=int(rng.choice([1, 5, 33, 120])), n_polylines=int(rng.choice([1, 40, 700])), batch=1 if rep > 1 else int(rng.choice([1, 2, 9])),
              seed=int(rng.randint(1 << 20)), goal=bool(rng.rand() < 0.5), tags=bool(rng.rand() < 0.5), ragged=bool(rng.rand() < 0.5),
              replay=float(rng.choice([0.0, 0.4])) if rep == 1 else 0.0)
    if spec.used_v2v_tags:
        kw["v2v"] = True
    if kw["n_agents"] < 3:
        kw["replay"] = 0.0
    sc = synth.make_scene(spec, **kw)
    if rep > 1:
        sc.pop("mode_choice", None)
    eng.set_scene(sc)
    for _ in range(int(rng.randint(1, 4))):
        eng.rollout()
    t = eng.get("traj")
    assert np.isfinite(t).all(), (it, kw)
    eng.world_trajs(np.eye(3, dtype=np.float32))
    if it % 20 == 19:
        print(it + 1, "cycles; device memory in use by this process' engines: %.1f MB" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e6), flush=True)
eng.close()
torch.cuda.synchronize()
left = (free0 - torch.cuda.mem_get_info()[0]) / 1e6
print("after the last close: %.1f MB not returned" % left)
assert left < 512.0, left   # (the runtime's graph / stream pools; a real leak of scene buffers would be GBs here)
