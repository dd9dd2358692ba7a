"""The REFERENCE's own Python (ProSim.forward(batch, 'val'), imported from /root/reference through oracle/ref_harness.py with the
torch_cluster / torch_geometric stand-ins) timed on one BASELINE configs[2] scene in the build container, and the oracle beside it
on the same cores -- context for bench.py's cpu_baseline (the reference cannot travel to the GPU box).  Writes
profiles/r04_reference_cpu_time.json.  Runs only where /root/reference exists."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from oracle import prosim_oracle as orc
import gen_golden as gg
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
scene = synth.baseline_scene(spec, 2, seed=0, batch=1)
threads = min(8, os.cpu_count() or 8)
torch.set_num_threads(threads)
ts_ref, ts_orc = [], []
for _ in range(3):
    t0 = time.perf_counter(); gg.run_reference(spec, w, scene); ts_ref.append(time.perf_counter() - t0)
with torch.no_grad():
    for _ in range(3):
        t0 = time.perf_counter(); orc.rollout(w, spec, scene); ts_orc.append(time.perf_counter() - t0)
A = int(scene["prompt_mask"].sum())
out = {"what": "one BASELINE configs[2] scene (128 agents, 1024 polylines, goal prompts), 80-step closed-loop rollout, torch fp32 on the BUILD CONTAINER's host cores",
       "threads": threads, "cpu_count": os.cpu_count(), "torch": torch.__version__,
       "reference_plus_standins": {"seconds_per_rollout": [round(t, 3) for t in ts_ref], "best": min(ts_ref),
                                   "agent_steps_per_s": A * spec.max_steps / min(ts_ref),
                                   "note": "model construction + state_dict load + forward (tests/gen_golden.py:run_reference); the first run also imports the reference"},
       "oracle_port": {"seconds_per_rollout": [round(t, 3) for t in ts_orc], "best": min(ts_orc), "agent_steps_per_s": A * spec.max_steps / min(ts_orc)}}
with open(os.path.join(ROOT, "profiles", "r04_reference_cpu_time.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
