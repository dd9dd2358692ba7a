"""prosim_amd.stream.RolloutPipeline: batches pipelined over several engines give the results of one engine, in order."""
import numpy as np
import pytest

from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC

pytestmark = pytest.mark.gpu


def test_pipeline_equals_single_engine_and_keeps_order():
    from prosim_amd.engine import Engine
    from prosim_amd.stream import RolloutPipeline
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    scenes = [synth.make_scene(spec, 6 + 5 * (i % 3), 20 + 17 * i, batch=1 + i % 2, seed=60 + i, goal=bool(i % 2), ragged=bool(i % 3 == 0),
                               replay=0.4 if i % 4 == 1 else 0.0) for i in range(7)]
    # one engine in the mode the pipeline runs its engines in: latency mode at depth 1, throughput mode (16 rows per
    # workgroup, every fused chain on k_chain16) beyond -- two kernels, two summation orders
    wants = {}
    for rows in (0, 16):
        eng = Engine(spec, w)
        eng.set_chain_rows(rows)
        wants[rows] = []
        for sc in scenes:
            eng.set_scene(sc)
            eng.rollout()
            wants[rows].append((eng.padded("traj"), eng.get("motion_pred")))
        eng.close()
    for a, b in zip(wants[0], wants[16]):
        assert np.abs(a[1][0] - b[1][0]).max() < 2e-5                      # replan 0: the modes agree to fp32 rounding
    for depth in (1, 2, 3):
        want = wants[0 if depth == 1 else 16]
        with RolloutPipeline(spec, w, depth=depth, outputs=("traj", "motion_pred")) as pipe:
            got = list(pipe.run(scenes))
            assert [i for i, _ in got] == list(range(len(scenes)))
            for (i, out), (traj, mp) in zip(got, want):
                assert np.array_equal(out["traj"], traj) and np.array_equal(out["motion_pred"], mp), (depth, i)
            # tickets keep counting across run() calls; a full pipeline (queue batches per engine) refuses another batch
            t0 = pipe.submit(scenes[0])
            for _ in range(pipe.capacity - 1):
                pipe.submit(scenes[1])
            with pytest.raises(RuntimeError, match="collect it"):
                pipe.submit(scenes[2])
            assert np.array_equal(pipe.collect(t0)["traj"], want[0][0])
            with pytest.raises(RuntimeError, match="not in flight"):
                pipe.collect(t0)
            if depth > 1:   # (t0 + 1 + depth is the second batch of the next engine: t0 + 1 is ahead of it)
                with pytest.raises(RuntimeError, match="not the oldest"):
                    pipe.collect(t0 + 1 + depth)
            for k in range(1, pipe.capacity):   # every queued batch -- uploaded behind a rollout in flight on its engine -- is scenes[1]'s result
                assert np.array_equal(pipe.collect(t0 + k)["traj"], want[1][0]), (depth, k)
