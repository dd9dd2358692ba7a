import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
eng = Engine(spec, w)
for batch in (8, 1):
    eng.set_scene(synth.baseline_scene(spec, 2, seed=0, batch=batch))
    eng.rollout(); eng.sync()
    ms, st = eng.time_rollout(3, 20)
    print("batch", batch, "flags", os.environ.get("PS_CHAIN_FLAGS", "0"), "ms/rollout %.3f" % ms, "stages", [round(x, 3) for x in st], "chain ms %.4f" % eng.time_policy_kernel(5), flush=True)
eng.close()
