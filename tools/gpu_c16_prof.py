"""In-kernel phase clocks of the policy launch (PS_CHAIN_PROF=1) on the bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
S = int(os.environ.get("PS_SCENES", "8"))
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(S)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
eng = Engine(spec, w)
eng.set_chain_impl(int(os.environ.get("PS_IMPL", "0")))
for rows in [int(r) for r in os.environ.get("PS_ROWS", "4,8,16").split(",")]:
    eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
    print("rows", rows, "policy launch ms", eng.time_policy_kernel(1), flush=True)
eng.close()
