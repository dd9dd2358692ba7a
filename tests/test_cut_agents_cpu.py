"""The committed cut-agent lists against the fp64 oracle's own margins (CPU; both files are fixtures: known_cut_agents.json is what
the -m gpu gates allow outside the 1e-4 band, near_cut_rows.json what oracle/cut_margin.py finds -- tests/gen_golden.py near_cut)."""
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_known_cut_agents_are_rows_the_oracle_puts_at_a_cut():
    with open(os.path.join(GOLD, "known_cut_agents.json")) as f:
        known = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    with open(os.path.join(GOLD, "near_cut_rows.json")) as f:
        near = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    assert set(known) == set(near)
    for workload, agents in known.items():
        primary = [a for a in agents if str(a) in near[workload]]
        # every list is led by a row whose own generator / policy edge lies within 1e-5 rad of a +-pi cut in the fp64 oracle; at most
        # one more agent of the same scene rides on it (the dense 256-agent no-truncation scene: agent 79 at 2.4e-4 beside 204)
        assert len(agents) == 0 or primary, (workload, agents, near[workload])
        assert len(agents) - len(primary) <= 1, (workload, agents, primary)
        assert all(m < 1e-5 for m in near[workload].values())
    assert near["baseline_configs/cfg4_seed0"]["254"] < 3e-6 and near["no_truncation/cfg4"]["204"] < 1e-6


def test_every_listed_cut_agent_is_outside_the_band_in_the_latest_committed_parity_table():
    """A listed agent that the engine holds inside 1e-4 is a stale entry: the gate would let it drift out again unnoticed."""
    import glob
    import re
    tables = sorted(glob.glob(os.path.join(os.path.dirname(GOLD), "..", "profiles", "r*_parity.json")),
                    key=lambda p: int(re.search(r"r(\d+)_parity", p).group(1)))
    assert tables, "no profiles/rNN_parity.json"
    with open(tables[-1]) as f:
        table = json.load(f)
    with open(os.path.join(GOLD, "known_cut_agents.json")) as f:
        known = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    for workload, agents in known.items():
        assert workload in table, (workload, os.path.basename(tables[-1]))
        outside = set(table[workload]["outside_1e4"])
        assert set(agents) == outside, f"{workload}: listed {sorted(agents)}, outside the band in {os.path.basename(tables[-1])}: {sorted(outside)}"
    with open(os.path.join(GOLD, "near_cut_rows.json")) as f:
        near = json.load(f)
    for workload, agents in known.items():   #  every listed agent is a row the oracle puts at a cut -- or rides on one (at most one per list, checked above)
        assert sum(str(a) in near[workload] for a in agents) >= min(1, len(agents)), (workload, agents)
