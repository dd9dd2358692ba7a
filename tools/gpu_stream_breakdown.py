"""Host-side cost of serving ONE new batch (8 x configs[2] scenes), piece by piece, and the pipeline's throughput by depth:
Engine.set_scene (numpy casts + ps_set_scene + condition setters), rollout() (signature check or capture + instantiate + graph
launch), prefetch, sync, padded() read-back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
from prosim_amd.stream import RolloutPipeline

spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
S = 8
batches = [synth.baseline_scene(spec, 2, seed=1000 + i, batch=S) for i in range(6)]
eng = Engine(spec, w)
eng.set_chain_rows(16)
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc.setdefault(name, []).append(1e3 * (t1 - t0)); return t1
for it in range(3):
    for b in batches:
        t = time.perf_counter()
        eng.set_scene(b); t = tick("set_scene (host)", t)
        eng.rollout(); t = tick("rollout() call (host)", t)
        eng.sync(); t = tick("sync (device time left)", t)
        eng.padded("traj"); eng.padded("vel"); t = tick("padded traj+vel", t)
print("graph captures / reuses:", eng.graph_stats())
for k, v in acc.items():
    print("%-28s first %.3f ms   median of the rest %.3f ms" % (k, v[0], float(np.median(v[1:]))))
eng.close()
A = S * 128
for depth in (1, 2, 3, 4, 5):
    with RolloutPipeline(spec, w, depth=depth) as pipe:
        for _ in pipe.run(batches[:depth]):
            pass
        t0 = time.perf_counter()
        n = sum(1 for _ in pipe.run(batches * 3))
        dt = time.perf_counter() - t0
        print("pipeline depth %d: %.3f ms per batch, %.2f M agent-steps/s, captures/reuses per engine %s" % (
            depth, 1e3 * dt / n, n * A * spec.max_steps / dt / 1e6, [e.graph_stats() for e in pipe.engines]), flush=True)
