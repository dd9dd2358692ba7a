import os, sys
sys.path.insert(0, '.')
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
eng = Engine(spec, weights.init_weights(spec, 0))
eng.set_scene(synth.baseline_scene(spec, 2, seed=0, batch=1))
eng.rollout(); eng.sync()
print("chain ms %.4f" % eng.time_policy_kernel(3), flush=True)
eng.close()
