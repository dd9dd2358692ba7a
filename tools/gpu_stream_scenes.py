"""Throughput over a STREAM of different scene batches (what a Sim-Agents style sweep does), against the repeated-
rollout figure bench.py reports: ps_set_scene (host preprocessing + uploads) and the first, eager rollout of a new
batch are outside bench.py's timed region."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
eng = Engine(spec, weights.init_weights(spec, 0))
batches = [synth.baseline_scene(spec, 2, seed=100 + i, batch=8) for i in range(6)]
eng.set_scene(batches[0]); eng.rollout(); eng.sync()
for rep in range(2):
    t_set = t_first = t_second = 0.0
    for sc in batches:
        t0 = time.perf_counter(); eng.set_scene(sc); t1 = time.perf_counter()
        eng.rollout(); eng.sync(); t2 = time.perf_counter()
        eng.rollout(); eng.sync(); t3 = time.perf_counter()
        t_set += t1 - t0; t_first += t2 - t1; t_second += t3 - t2
    n = len(batches)
    print("per 8-scene batch: set_scene %.2f ms, first rollout %.2f ms, repeated rollout %.2f ms -> stream %.2f M agent-steps/s vs repeated %.2f M"
          % (1e3 * t_set / n, 1e3 * t_first / n, 1e3 * t_second / n, 8 * 128 * 80 / ((t_set + t_first) / n) / 1e6, 8 * 128 * 80 / (t_second / n) / 1e6), flush=True)
eng.close()

# ---- two engines ping-pong: the host prepares batch k+1 (set_scene on the other engine: its own buffers and stream)
# while the GPU still runs batch k; results of batch k are read when its engine comes up again
engs = [Engine(spec, weights.init_weights(spec, 0)) for _ in range(2)]
for e in engs:
    e.set_scene(batches[0]); e.rollout(); e.sync()
for rep in range(2):
    t0 = time.perf_counter()
    pending = [None, None]
    n = 0
    for k, sc in enumerate(batches * 2):
        e = engs[k & 1]
        if pending[k & 1] is not None:
            e.sync(); _ = e.get("traj"); n += 1          # results of the batch this engine ran two steps ago
        e.set_scene(sc)
        e.rollout()
        pending[k & 1] = k
    for i in range(2):
        if pending[i] is not None:
            engs[i].sync(); _ = engs[i].get("traj"); n += 1
    dt = time.perf_counter() - t0
    print("ping-pong (2 engines), results read back: %.2f ms per 8-scene batch -> stream %.2f M agent-steps/s" % (1e3 * dt / n, n * 8 * 128 * 80 / dt / 1e6), flush=True)
for e in engs:
    e.close()
