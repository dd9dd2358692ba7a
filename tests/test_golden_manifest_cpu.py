"""Fixture hygiene: every file under tests/golden/ is what tests/gen_golden.py last wrote
(MANIFEST.sha256, written at the end of every run of that script), nothing is missing and nothing is unlisted; and the
derived formatter fixture was cut from the committed agent table (the `z` column: round 3 shipped a table that predated it)."""
import hashlib
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _manifest():
    out = {}
    with open(os.path.join(GOLD, "MANIFEST.sha256")) as f:
        for line in f:
            h, rel = line.rstrip("\n").split("  ", 1)
            out[rel] = h
    return out


def test_every_fixture_matches_the_manifest():
    want = _manifest()
    have = {}
    for dirpath, _, files in os.walk(GOLD):
        for fn in files:
            if fn == "MANIFEST.sha256":
                continue
            path = os.path.join(dirpath, fn)
            with open(path, "rb") as f:
                have[os.path.relpath(path, GOLD).replace(os.sep, "/")] = hashlib.sha256(f.read()).hexdigest()
    assert sorted(set(want) - set(have)) == [], "listed in MANIFEST.sha256 but missing"
    assert sorted(set(have) - set(want)) == [], "under tests/golden/ but not in MANIFEST.sha256 (run tests/gen_golden.py)"
    changed = sorted(k for k in want if want[k] != have[k])
    assert changed == [], ("fixtures differ from what gen_golden.py last wrote: " +
                           "; ".join(f"{k} (owner: {_owner(k)})" for k in changed))


def _owner(rel: str) -> str:
    """Which command rewrites a fixture -- and with it the manifest (the failure should say where to look)."""
    if rel.endswith(".json") and not rel.startswith("oracle_cache/"):
        return ("hand-edited list: re-run `python tests/gen_golden.py manifest` after editing"
                if rel == "known_cut_agents.json" else "python tests/gen_golden.py near_cut")
    if rel.startswith("oracle_cache/"):
        return "python tests/gen_golden.py oracle_cache"
    if rel.startswith("ref_standins_"):
        return "python tests/gen_golden.py only " + rel[len("ref_standins_"):-len(".npz")]
    if rel.startswith("ref_format_") or rel.startswith("demo_"):
        return "python tests/gen_golden.py format | tracks | map"
    return "python tests/gen_golden.py (the branch that writes it; no argument = the pure primitives)"


def test_agent_tables_carry_every_column_the_script_writes():
    for scene in ("scene_0", "scene_1"):
        g = np.load(os.path.join(GOLD, f"demo_{scene}_agent_table.npz"))
        for c in ("agent_id", "scene_ts", "origin", "x", "y", "z", "vx", "vy", "ax", "ay", "heading", "length", "width"):
            assert c in g.files, (scene, c)
