"""Single configs[2] scene (128 agents): policy-launch and rollout latency by fused-chain implementation and rows per workgroup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
scene = synth.baseline_scene(spec, 2, seed=0, batch=1)
eng = Engine(spec, w)
ref = None
for impl, rows in ((1, 0), (2, 1), (2, 2), (2, 4), (2, 8), (2, 16)):
    eng.set_chain_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
    traj = eng.padded("traj")
    if ref is None: ref = traj
    ms_roll, st = eng.time_rollout(2, 5)
    print("1 scene impl %d rows %2d: policy launch %.4f ms, rollout %.3f ms (enc %.3f gen %.3f loop %.3f) | traj vs impl1 %.2e" % (
        impl, rows, eng.time_policy_kernel(3), ms_roll, st[0], st[1], st[2], float(np.abs(traj - ref).max())), flush=True)
eng.close()
