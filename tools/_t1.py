import os, sys, hashlib
sys.path.insert(0, '.')
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
eng = Engine(spec, weights.init_weights(spec, 0))
eng.set_scene(synth.baseline_scene(spec, 2, seed=0, batch=1))
eng.rollout(); eng.sync()
d = hashlib.sha256(eng.get("motion_pred").tobytes()).hexdigest()[:12]
ms, st = eng.time_rollout(2, 10)
print(os.environ.get("PS_LIB"), "chain ms %.4f rollout %.3f ms (enc %.3f gen %.3f loop %.3f) digest %s" % (eng.time_policy_kernel(3), ms, st[0], st[1], st[2], d), flush=True)
eng.close()
