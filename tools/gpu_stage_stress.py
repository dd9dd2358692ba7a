"""Which stage is not repeatable under load?  Three engines replay full rollouts as background load; three more (one configs[2] scene each) run the
path up to ONE stage again and again -- encode_scene, + generate_policy, + the first policy step (PS_STAGES) -- and compare its output with
the first run's bit for bit.  usage: python tools/gpu_stage_stress.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
load = []
for k in range(3):
    e = Engine(spec, w); e.set_chain_impl(int(os.environ.get("PS_LOAD_IMPL", "0"))); e.set_scene(synth.baseline_scene(spec, 2, seed=10 + k, batch=1)); e.rollout(); load.append(e)
probe = []
for k in range(3):
    e = Engine(spec, w)
    e.set_chain_impl(int(os.environ.get("PS_IMPL", "0")))
    e.set_search_impl(int(os.environ.get("PS_SEARCH_IMPL", "0")))
    e.set_scene(synth.baseline_scene(spec, 2, seed=3 + k, batch=1))
    probe.append(e)
def run_stage(e, stage):
    if stage == "encode":
        e.encode_scene(); return e.get("scene_tokens")
    if stage == "generate":   # (on a fresh encoding every time: the stage is not idempotent on its own output buffers)
        e.encode_scene(); e.generate_policy(); return e.get("policy_emd")
    e.encode_scene(); e.generate_policy(); e.reset_rollout(); e.policy_step(0); return e.get("motion_pred")[0]
for stage in os.environ.get("PS_STAGES", "encode").split(","):
    ref = [run_stage(e, stage).copy() for e in probe]
    bad = 0
    for it in range(n):
        if not os.environ.get("PS_NOLOAD"):
            for e in load: e.rollout()
        for k, e in enumerate(probe):
            out = run_stage(e, stage)
            if not np.array_equal(out, ref[k]):
                bad += 1
                d = out != ref[k]
                rows = np.nonzero(d.reshape(d.shape[0], -1).any(-1))[0]
                if bad <= 6:
                    print("  %s it %d engine %d: %d rows differ (first %s), max abs diff %.3e" % (stage, it, k, len(rows), rows[:6].tolist(), float(np.abs(out - ref[k]).max())), flush=True)
    print("%s: %d of %d runs differ from the first" % (stage, bad, 3 * n), flush=True)
for e in load + probe: e.close()
