"""Throughput with K engines (K streams) sharing one GPU, 8 scenes total."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
for total, K in ((8, 1), (8, 2), (8, 4), (8, 8), (16, 2), (16, 4), (32, 4)):
    per = total // K
    engs = []
    for k in range(K):
        e = Engine(spec, w)
        e.set_scene(synth.baseline_scene(spec, 2, seed=k, batch=per))
        engs.append(e)
    for _ in range(3):
        for e in engs: e.rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        for e in engs: e.rollout()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{total} scenes as {K} engines x {per}: {dt*1e3:.2f} ms per round -> {total*128*80/dt/1e6:.2f} M agent-steps/s", flush=True)
    for e in engs: e.close()
