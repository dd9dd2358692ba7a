"""k_chain16 (ps_set_chain_impl 0) against k_attn_chain (impl 1) and the fp64 oracle; policy-launch timings by rows per workgroup."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
torch.set_num_threads(16)

def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())

spec = SMALL_SPEC
w = weights.init_weights(spec, 0)
scene = synth.make_scene(spec, 24, 160, batch=3, seed=5, goal=True, ragged=True)
with torch.no_grad():
    o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
eng = Engine(spec, w)
res = {}
for impl in (1, 2):
    for rows in ((0,) if impl == 1 else (0, 1, 2, 4, 8, 12, 16)):
        eng.set_chain_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
        A = eng.num_agents
        mp = eng.get("motion_pred")
        res[(impl, rows)] = eng.padded("traj")
        print("small impl %d rows %2d: replan-0 err %.2e  closed-loop traj err %.2e  (vs impl1: %.2e)" % (
            impl, rows, err(mp[0], o64["motion_pred"][:A].numpy()), err(res[(impl, rows)], o64["traj"].numpy()),
            err(res[(impl, rows)], res[(1, 0)])), flush=True)
eng.close()

spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
eng = Engine(spec, w)
ref = None
for impl, rows in ((1, 0), (2, 0), (2, 2), (2, 4), (2, 8), (2, 12), (2, 16)):
    eng.set_chain_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
    traj = eng.padded("traj"); mp0 = eng.get("motion_pred")[0]
    if ref is None:
        ref = (traj, mp0)
    ms_roll, st = eng.time_rollout(1, 3)
    ms_chain = eng.time_policy_kernel(2)
    print("8x cfg2 impl %d rows %2d: policy launch %.3f ms, rollout %.2f ms (enc %.2f gen %.2f loop %.2f) | replan-0 vs impl1 %.2e, traj vs impl1 %.2e" % (
        impl, rows, ms_chain, ms_roll, st[0], st[1], st[2], err(mp0, ref[1]), err(traj, ref[0])), flush=True)
eng.close()
