"""RolloutPipeline over a stream of NEW 8-scene batches: throughput by depth (set_scene + capture + rollout + read-back)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.stream import RolloutPipeline
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
batches = [synth.baseline_scene(spec, 2, seed=100 + i, batch=8) for i in range(8)]
for depth in (1, 2, 3, 4):
    with RolloutPipeline(spec, w, depth=depth) as pipe:
        for _ in pipe.run(batches[:depth]):
            pass
        t0 = time.perf_counter()
        n = sum(1 for _ in pipe.run(batches * 3))
        dt = time.perf_counter() - t0
    print("depth %d: %.2f ms per 8-scene batch -> %.2f M agent-steps/s" % (depth, 1e3 * dt / n, n * 8 * 128 * 80 / dt / 1e6), flush=True)
