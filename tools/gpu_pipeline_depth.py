"""RolloutPipeline throughput over a stream of NEW 8-scene batches by depth (engines in flight), several repeats per depth, beside
the resident loop of bench.py (the same batch rolled out again and again on `depth` engines in turn)."""
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
A = S * 128
batches = [synth.baseline_scene(spec, 2, seed=1000 + i, batch=S) for i in range(6)]
depths = [int(d) for d in os.environ.get("PS_DEPTHS", "1,2,3,4,5,6").split(",")]
reps = int(os.environ.get("PS_REPS", "3"))
nb = int(os.environ.get("PS_BATCHES", "60"))
for depth in depths:
    with RolloutPipeline(spec, w, depth=depth) as pipe:
        for _ in pipe.run(batches[:depth]):
            pass
        vals = []
        for r in range(reps):
            t0 = time.perf_counter()
            n = sum(1 for _ in pipe.run((batches * ((nb + 5) // 6))[:nb]))
            dt = time.perf_counter() - t0
            vals.append(n * A * spec.max_steps / dt / 1e6)
        # resident: the engines keep their last batch; rollouts issued round-robin, an engine is synchronised before it is reused
        engs = pipe.engines
        for e in engs:
            e.rollout()
        for e in engs:
            e.sync()
        res = []
        for r in range(reps):
            t0 = time.perf_counter()
            for i in range(nb):
                e = engs[i % depth]
                e.sync()
                e.rollout()
            for e in engs:
                e.sync()
            res.append(nb * A * spec.max_steps / (time.perf_counter() - t0) / 1e6)
        print("depth %d: streaming %s M agent-steps/s | resident %s M" % (depth, " ".join("%.2f" % v for v in vals), " ".join("%.2f" % v for v in res)), flush=True)
