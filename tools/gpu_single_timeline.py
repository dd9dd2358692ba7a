"""One configs[2] scene (128 agents), latency mode: a few rollouts for a kernel trace (tools/prof_timeline.py prints the slice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
eng = Engine(spec, weights.init_weights(spec, 0))
if os.environ.get("PS_IMPL"): eng.set_chain_impl(int(os.environ["PS_IMPL"]))   # 1: the operand-image chains (the cross-check path)
eng.set_scene(synth.baseline_scene(spec, 2, seed=0, batch=int(os.environ.get("PS_SCENES", "1"))))
for _ in range(4):
    eng.rollout(); eng.sync()
eng.close()
