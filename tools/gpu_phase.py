"""In-kernel phase clocks of the policy launch (PS_CHAIN_PROF=1): mean cycles per workgroup per phase.
Phases: 0 LN + q/s/g GEMVs, 1 q~ GEMV, 2 edge setup, 3 score pass, 4 softmax, 5 (unused), 6 aggregation r~ (MFMA), 7 aggregation v,
8-9 fold, 10 to_v_r, 11 gate / to_out, 12 FFN up, 13 FFN down.  Usage: PS_CHAIN_PROF=1 [PS_CHAIN_T=84] python tools/gpu_phase.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
eng = Engine(spec, weights.init_weights(spec, 0))
eng.set_scene(synth.baseline_scene(spec, 2, seed=0, batch=8))
eng.rollout(); eng.sync()
print("chain ms %.4f" % eng.time_policy_kernel(3), flush=True)
eng.close()
