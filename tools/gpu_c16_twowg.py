"""Timing experiment (tools build -DPS_EXPERIMENTS -DPS_C16_ABL_ONE_SLOT, wrong results by construction): the first policy launch of an
8192-row batch (one configs[2] scene x 64 replicas, 512 workgroups of 16 rows) with 8-wave workgroups (one per CU, two rounds) or,
PS_C16_NW=4, 4-wave workgroups (two per CU, one round)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
eng = Engine(spec, weights.init_weights(spec, 0))
eng.set_replicas(int(os.environ.get("PS_REPLICAS", "64")))
eng.set_chain_rows(16)
eng.set_scene(synth.baseline_scene(spec, 2, seed=0, batch=1))
eng.enable_policy_events(True)
ts = []
for _ in range(4):
    eng.rollout(); eng.sync()
    ts.append(eng.policy_event_times().copy())
ts = np.array(ts)
print("NW", os.environ.get("PS_C16_NW", "8"), "rows", eng.num_policy_agents, "first policy launch ms (per rollout):", np.round(ts[:, 0], 4), "all replans of the last:", np.round(ts[-1], 3))
eng.close()
