"""Does the pipelined loop slow down as it runs?  Per-chunk step times of tools/gpu_headline_loop.py's loop over a few seconds, with the
shader clock / power rocm-smi reports between chunks (a power- or clock-limited steady state shows as a step time that rises after the
first tens of milliseconds)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
engines = [Engine(spec, w) for _ in range(4)]
for e in engines:
    e.set_chain_rows(16); e.set_scene(scene); e.rollout(); e.sync()
def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in o.splitlines() if ("sclk" in l or "Power" in l or "junction" in l.lower()) and "GPU[0]" in l]
        return " | ".join(k.split(":", 1)[-1].strip()[:60] for k in keep[:4])
    except Exception as ex:
        return f"(rocm-smi: {ex})"
print("idle:", smi(), flush=True)
k = 0
for chunk, n in enumerate([8, 8, 16, 32, 64, 128, 256, 512]):
    for e in engines: e.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        engines[k % 4].rollout(); k += 1
    for e in engines: e.sync()
    dt = time.perf_counter() - t0
    print(f"chunk of {n:4d} steps from cold pipeline: {1e3 * dt / n:.3f} ms per step", flush=True)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:   # ~6 s without a drain, sampling the clocks while it runs
    for _ in range(64):
        engines[k % 4].rollout(); k += 1
    n += 64
    if n % 512 == 0:
        for e in engines: e.sync()
        print(f"  sustained, {n} steps in: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per step; {smi()}", flush=True)
for e in engines: e.sync()
for e in engines: e.close()
