"""Many engines side by side, each replaying a rollout whose searches all run on the one-launch kernel (k_radius_geo): launches of several engines
overlap on a full GPU, the case its bounded look-back wait is for.  Every replay must return the bits of the engine's first rollout; the
whole run must finish (a wave that waits forever would hang it).  usage: python tools/gpu_search_stress.py [engines] [rollouts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from prosim_amd.engine import Engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
engs, first = [], []
for k in range(K):
    if k % 3 == 0:
        spec, scene = DEMO_SPEC, synth.baseline_scene(DEMO_SPEC, 2, seed=k, batch=1)
    elif k % 3 == 1:
        spec, scene = SMALL_SPEC, synth.make_scene(SMALL_SPEC, 40 + 7 * k, 300, batch=2, seed=k, goal=True, ragged=True)
    else:
        spec, scene = SMALL_SPEC, synth.make_scene(SMALL_SPEC, 100, 60, batch=1, seed=k, tags=True, replay=0.3)
    e = Engine(spec, weights.init_weights(spec, 0))
    e.set_search_impl(int(os.environ.get("PS_SEARCH_IMPL", "0")))
    e.set_chain_impl(int(os.environ.get("PS_IMPL", "0")))
    e.set_row_impl(int(os.environ.get("PS_ROW_IMPL", "0")))
    e.set_scene(scene); e.rollout(); e.sync()
    engs.append(e); first.append((e.padded("traj").copy(), e.get("scene_tokens").copy(), e.get("policy_emd").copy(), e.get("motion_pred").copy()))
t0 = time.perf_counter()
bad = 0
for it in range(n):
    for e in engs: e.rollout()
    if it % 25 == 24 or it == n - 1:
        for k, (e, f) in enumerate(zip(engs, first)):
            e.sync()
            t = e.padded("traj")
            if not np.array_equal(t, f[0]):
                bad += 1
                tok, emd, mp = e.get("scene_tokens"), e.get("policy_emd"), e.get("motion_pred")
                nmap = tok.shape[0] - emd.shape[0]
                rep = [int(r) for r in range(mp.shape[0]) if not np.array_equal(mp[r], f[3][r])]
                print("  it %d engine %d (kind %d): map tokens differ %d, agent tokens differ %d, policy_emd rows differ %d, first differing replan %s" % (
                    it, k, k % 3, int((tok[:nmap] != f[1][:nmap]).any(-1).sum()), int((tok[nmap:] != f[1][nmap:]).any(-1).sum()),
                    int((emd != f[2]).any(-1).sum()), rep[:1]), flush=True)
dt = time.perf_counter() - t0
print("%d engines x %d rollouts side by side: %.1f s, %d mismatching checks" % (K, n, dt, bad), flush=True)
for e in engs: e.close()
sys.exit(1 if bad else 0)
