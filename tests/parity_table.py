"""Closed-loop parity numbers, collected by the -m gpu tests and written as ONE JSON table: gpurun_out/r02_parity.json on
the GPU box (merged back by gpurun), copied to profiles/r02_parity.json for the record.  Per entry: replan-0 max error
(open loop), per-agent closed-loop max error of the trajectories: median / 99th percentile / max, the fraction of agents
within 1e-4, and -- where the test computed it -- the fp32 floor (fp32 oracle against the fp64 oracle on the same scene)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ROWS = {}


def per_agent(err_per_agent: np.ndarray) -> dict:
    d = np.asarray(err_per_agent, np.float64)
    return dict(agents=int(d.size), median=float(np.median(d)), p99=float(np.percentile(d, 99)), max=float(d.max()),
                within_1e4=float((d < 1e-4).mean()))


def record(name: str, **fields):
    _ROWS[name] = fields
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "r02_parity.json")
    table = {}
    if os.path.exists(path):
        try:
            with open(path) as f:
                table = json.load(f)
        except Exception:
            table = {}
    table[name] = fields
    with open(path, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
