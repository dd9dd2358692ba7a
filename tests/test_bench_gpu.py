"""bench.py end to end on the GPU box: the driver's command lines, the one JSON line, and the multi-rank code path
(process group + the RCCL gather at metric-compute time) taken with a single rank (PS_BENCH_FORCE_DIST)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(extra_env, args):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                  # ONE JSON line
    return json.loads(lines[0])


def test_bench_line_single_and_forced_distributed():
    plain = _run({}, ["--steps", "8", "--warmup", "4", "--no-cpu-baseline"])
    assert REQUIRED <= set(plain) and plain["n_gpus"] == 1 and plain["steps"] == 8 and plain["warmup"] == 4
    assert plain["data"] == "synthetic" and plain["scaling"] == "weak" and plain["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(plain["roofline"])
    assert plain["value"] > 1e6 and abs(plain["value"] - 8 * 128 * 80 / (plain["ms_per_step"] * 1e-3)) < 1e-3 * plain["value"]
    dist = _run({"PS_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29578"}, ["--steps", "8", "--warmup", "4", "--no-cpu-baseline"])
    # same scenes, same engine: the gathered metrics equal the local ones, and the gather costs a few per cent at most
    # (warmup 4 = one pass of every in-flight engine, so no first-use cost lands in the 8 timed steps)
    assert dist["rollout_metrics"] == plain["rollout_metrics"] and dist["rollout_metrics"]["scenes"] == 8
    assert dist["value"] > 0.9 * plain["value"]
