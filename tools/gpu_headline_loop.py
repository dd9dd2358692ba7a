"""The headline loop of bench.py alone (4 engines x 8 configs[2] scenes, throughput mode, rollouts in turn) -- for a kernel trace
whose CU-time table (tools/prof_cu_time.py) holds this workload only.  PS_IMPL=3: the fast_encoder variant; PS_ROW_IMPL: ps_set_row_impl (1 = the staged row kernels, 11..13 = forced row tiles per wave); PS_SEARCH_IMPL: ps_set_search_impl (1 = count / fill / record launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
S, n_fl, steps = int(os.environ.get("PS_SCENES", "8")), int(os.environ.get("PS_INFLIGHT", "4")), int(os.environ.get("PS_STEPS", "24"))
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(S)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
engines = [Engine(spec, w) for _ in range(n_fl)]
for e in engines:
    e.set_row_impl(int(os.environ.get("PS_ROW_IMPL", "0")))
    e.set_search_impl(int(os.environ.get("PS_SEARCH_IMPL", "0")))
    e.set_chain_impl(int(os.environ.get("PS_IMPL", "0")))
    e.set_chain_rows(int(os.environ.get("PS_ROWS", "16")))
    e.set_scene(scene)
for k in range(2 * n_fl):
    engines[k % n_fl].rollout()
for e in engines:
    e.sync()
t0 = time.perf_counter()
for k in range(steps):
    engines[k % n_fl].rollout()
for e in engines:
    e.sync()
dt = time.perf_counter() - t0
print("%.3f ms per step, %.2f M agent-steps/s" % (1e3 * dt / steps, engines[0].num_agents * spec.max_steps * steps / dt / 1e6))
for e in engines:
    e.close()
