"""Closed-loop parity numbers, collected by the -m gpu tests and written as ONE JSON table: gpurun_out/r06_parity.json on
the GPU box (merged back by gpurun), copied to profiles/r06_parity.json for the record.  Per entry: replan-0 max error
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
                within_1e4=float((d < 1e-4).mean()), outside_1e4=[int(i) for i in np.nonzero(d >= 1e-4)[0]])


# Agents known to sit behind a +-pi cut of the reference's math on a given workload (DESIGN.md "branch cuts": an fp32 run -- the
# reference's own included -- lands on the other side of one wrap_angle / atan2 discontinuity and stays ~1e-3 away for the rest
# of that agent's rollout).  The closed-loop gates allow THESE agents outside the 1e-4 band and fail when a new one leaves it.
KNOWN_CUTS = os.path.join(ROOT, "tests", "golden", "known_cut_agents.json")


def known_cut_agents(workload: str) -> set:
    if not os.path.exists(KNOWN_CUTS):
        return set()
    with open(KNOWN_CUTS) as f:
        return set(json.load(f).get(workload, []))


PARITY_TABLE = "r06_parity.json"   # the table of this round: gpurun_out/ on the GPU box, profiles/ for the record


def closed_loop_gate(workload: str, d: np.ndarray, tol: float = 1e-4):
    """The closed-loop bar at what is measured: every agent outside the 1e-4 band is on the committed list of that workload's cut
    agents, AND at most 0.5 % of the agents are outside -- a bound of its own, which the list cannot widen (round 4's form,
    max(0.5 %, len(list)), could never fail once the first assertion held: ADVICE round 4) --, median below 2e-5 on the timed
    workload (3e-5 elsewhere), nobody beyond 2e-3.  No workload needs an exception today: the two listed agents (cfg3 seed 0 #25,
    no-truncation cfg4 #204) are 1 of 256 each."""
    d = np.asarray(d, np.float64)
    outside = set(int(i) for i in np.nonzero(d >= tol)[0])
    known = known_cut_agents(workload)
    new = outside - known
    assert not new, f"{workload}: agents {sorted(new)} left the {tol:g} band (errors {[float(d[i]) for i in sorted(new)]}); known cut agents: {sorted(known)}"
    assert len(outside) <= int(0.005 * d.size), (workload, sorted(outside), float((d < tol).mean()))
    assert len(known) <= max(1, int(0.005 * d.size)), (workload, sorted(known))   # the list itself stays within the bar it excuses
    # median: 2e-5 on the timed workload (measured 1.5e-5); the 64-agent config sits at 2.5e-5 (its fp32 oracle: 2.4e-5)
    med_bar = 2e-5 if workload.startswith("bench_workload") else 3e-5
    assert np.median(d) < med_bar and d.max() < 2e-3, (workload, float(np.median(d)), float(d.max()))


def record(name: str, **fields):
    _ROWS[name] = fields
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, PARITY_TABLE)
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
