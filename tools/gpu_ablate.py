import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
eng = Engine(spec, weights.init_weights(spec, 0))
scene = synth.baseline_scene(spec, 2, seed=0, batch=int(os.environ.get('PS_BATCH', '1')))
eng.set_scene(scene); eng.rollout(); eng.sync()
ms, st = eng.time_rollout(1, 5)
print("batch", os.environ.get("PS_BATCH", "1"), "flags", os.environ.get("PS_CHAIN_FLAGS", "0"), f"rollout {ms:.3f} stages {[round(x,3) for x in st]} policy chain {eng.time_policy_kernel(3)*1e3:.1f} us")
if os.environ.get("PS_CHAIN_FLAGS", "0") == "0":
    print("agents", eng.num_agents, "edge_counts [a2a s2s p2p s2p a2p m2p cnd]", eng.get("edge_counts").tolist())
