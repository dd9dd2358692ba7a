"""prosim_amd/formatting.py on the demo_dataset agent table (tests/golden/demo_scene_0_agent_table.npz, made by
tests/gen_golden.py from the reference's sample data): the invariants get_center_obs / local_map_to_sym_coord imply,
and that the oracle rolls the formatted scene out."""
import os

import numpy as np
import torch

from oracle import prosim_oracle as orc
from prosim_amd import formatting as fmt, weights
from prosim_amd.spec import SMALL_SPEC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "golden", "demo_scene_0_agent_table.npz")


def _tracks():
    g = np.load(TABLE)
    return fmt.tracks_from_table({k: g[k] for k in g.files if k != "origin"})


def test_tracks_table_roundtrip():
    g = np.load(TABLE)
    tr = _tracks()
    assert tr["x"].shape == (55, 91) and len(tr["agent_ids"]) == 55
    assert int(np.isfinite(tr["x"]).sum()) == len(g["scene_ts"])          # every row lands in exactly one cell
    i, t = list(tr["agent_ids"]).index(str(g["agent_id"][100])), int(g["scene_ts"][100])
    assert tr["y"][i, t] == np.float64(g["y"][100])


def test_history_is_in_each_agents_own_frame():
    spec, tr, t0 = SMALL_SPEC, _tracks(), 10
    sc = fmt.scene_from_tracks(spec, tr, t0)
    N = sc["obs_input"].shape[1]
    present = np.isfinite(tr["x"][:, t0])
    assert N == int(present.sum()) and sc["prompt_mask"].all()
    o, m = sc["obs_input"][0], sc["obs_mask"][0]
    # the current step is the frame origin: (0, 0), heading 0
    assert np.abs(o[:, -1, 0:2]).max() == 0 and np.abs(o[:, -1, 2]).max() == 0 and np.abs(o[:, -1, 3] - 1).max() == 0
    # missing steps are NaN and masked, present ones finite
    assert (np.isfinite(o) == m).all() and m[:, -1, :8].all() and (~m[:, 0, :8]).any()
    # undo the frame: R(h0) p + p0 gives back the table
    sel = np.nonzero(present)[0]
    h0, p0 = sc["obs_head"][0].astype(np.float64), sc["obs_pos"][0].astype(np.float64)
    for j in range(spec.hist_steps):
        ok = m[:, j, 0]
        c, s = np.cos(h0), np.sin(h0)
        wx = o[:, j, 0] * c - o[:, j, 1] * s + p0[:, 0]
        wy = o[:, j, 0] * s + o[:, j, 1] * c + p0[:, 1]
        t = t0 - spec.hist_steps + 1 + j
        assert np.abs(wx[ok] - tr["x"][sel, t][ok]).max() < 1e-3 and np.abs(wy[ok] - tr["y"][sel, t][ok]).max() < 1e-3
    # speed is frame-invariant; sin^2 + cos^2 = 1; the prompt carries the local velocity and the extent
    sp_local = np.hypot(o[:, -1, 4], o[:, -1, 5])
    sp_world = np.hypot(tr["vx"][sel, t0], tr["vy"][sel, t0])
    assert np.abs(sp_local - sp_world).max() < 1e-4
    assert np.abs(o[..., 2][m[..., 2]] ** 2 + o[..., 3][m[..., 3]] ** 2 - 1).max() < 1e-5
    assert np.array_equal(sc["prompt"][0, :, 0:2], np.nan_to_num(o[:, -1, 4:6])) and (sc["prompt"][0, :, 2:4] > 0).all()
    assert (o[:, :, 13:13 + spec.hist_steps] == np.eye(spec.hist_steps)).all()


def test_polylines_are_in_their_midpoint_tangent_frame():
    spec, tr = SMALL_SPEC, _tracks()
    lanes = fmt.lanes_from_tracks(tr)
    mp = fmt.polylines_to_map(spec, lanes)
    inp, msk = mp["map_input"][0], mp["map_mask"][0]
    assert inp.shape[0] == len(lanes) > 20 and msk[:, 0].all()
    n = msk.sum(1)
    first, last = inp[:, 0, 0:2], inp[np.arange(len(n)), n - 1, 2:4]
    # first start and last end are mirror images on the local x axis (midpoint at the origin, tangent along +x)
    assert np.abs(first + last).max() < 1e-3 and np.abs(last[:, 1]).max() < 1e-3 and (last[:, 0] > 0).all()
    d = inp[..., 9:11][msk]
    assert np.abs(np.linalg.norm(d, axis=-1) - 1).max() < 1e-4
    # back in the scene frame the first vertex is the lane's first vertex
    c, s = np.cos(mp["map_head"][0]), np.sin(mp["map_head"][0])
    wx = first[:, 0] * c - first[:, 1] * s + mp["map_pos"][0, :, 0]
    assert np.abs(wx - np.array([l[0, 0] for l in lanes])).max() < 1e-2


def test_oracle_rolls_out_the_demo_scene():
    """BASELINE configs[0] (plumbing): a demo_dataset scene, 16 agents, 20 steps, on the CPU path."""
    spec = SMALL_SPEC.replace(max_steps=20)
    sc = fmt.scene_from_tracks(spec, _tracks(), 10, max_agents=16)
    sc.pop("agent_ids")
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o = orc.rollout(w, spec, sc)
    assert o["traj"].shape == (1, 16, 20, 4) and torch.isfinite(o["traj"]).all()
