"""prosim_amd/vecmap.py: the protobuf wire decoder of the demo cache's vector maps and the lane-vector pipeline of
prosim/dataset/data_utils.py:156-255 + format_utils.py:150-263, on the reference's own sample map
(tests/golden/demo_waymo_train_1_map.pb, a data file copied by tests/gen_golden.py) and on hand-built messages."""
import os
import struct

import numpy as np
import pytest
import torch

from oracle import prosim_oracle as orc
from prosim_amd import formatting as fmt, vecmap as vm, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pb():
    with open(os.path.join(GOLD, "demo_waymo_train_1_map.pb"), "rb") as f:
        return f.read()


def _tracks():
    g = np.load(os.path.join(GOLD, "demo_scene_1_agent_table.npz"))
    tr = fmt.tracks_from_table({k: g[k] for k in g.files if k != "origin"})
    return tr, g["origin"].astype(np.float64), g


# ---- a minimal protobuf WRITER (test side only) to build messages of the layout the decoder documents ----
def _vi(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _vi(field << 3 | 2) + _vi(len(payload)) + payload


def _zz(vals):
    return b"".join(_vi(((int(v) << 1) ^ (int(v) >> 31)) & 0xFFFFFFFF) for v in vals)


def _pt(p):
    return b"".join(_vi(f << 3 | 1) + struct.pack("<d", v) for f, v in zip((1, 2, 3), p))


def _pl(mm, head=None):
    mm = np.asarray(mm, np.int64)
    d = np.diff(mm, axis=0, prepend=0)
    b = _ld(1, _zz(d[:, 0])) + _ld(2, _zz(d[:, 1])) + _ld(3, _zz(d[:, 2]))
    if head is not None:
        b += _ld(4, np.asarray(head, "<f8").tobytes())
    return b


def test_wire_round_trip_bit_exact():
    rng = np.random.default_rng(0)
    origin = np.array([1234.5, -987.25, 12.0])
    lanes_mm = [np.cumsum(rng.integers(-70000, 70000, size=(n, 3)), axis=0) for n in (2, 7, 300)]
    heads = [rng.uniform(-3, 3, len(l)) for l in lanes_mm]
    msg = _ld(1, b"env:map_7")
    for i, (l, h) in enumerate(zip(lanes_mm, heads)):
        lane = _ld(1, _pl(l, h)) + (_ld(2, _pl(l + 1500)) if i else b"") + _ld(3, _pl(l - 1500)) + _ld(4, b"9") + _ld(5, b"10_1") + _ld(5, b"11")
        msg += _ld(2, _ld(1, f"{i}_x".encode()) + _ld(2, lane))
    msg += _ld(2, _ld(1, b"cw") + _ld(4, _ld(1, _pl(lanes_mm[0]))))            # a crosswalk: counted, not a lane
    msg += _ld(3, _pt(origin + 5)) + _ld(4, _pt(origin)) + _ld(5, _pt(origin))
    m = vm.decode_vector_map(msg)
    assert m["name"] == "env:map_7" and len(m["lanes"]) == 3 and len(m["others"]["ped_crosswalk"]) == 1
    assert (m["origin"] == origin).all() and (m["max_pt"] == origin + 5).all()
    for i, (l, h, d) in enumerate(zip(lanes_mm, heads, m["lanes"])):
        assert d["id"] == f"{i}_x" and d["entry"] == ["9"] and d["exit"] == ["10_1", "11"] and d["adj_left"] == []
        assert np.array_equal(d["center"][:, :3], l / 1000.0 + origin) and np.array_equal(d["center"][:, 3], h)   # exact mm
        assert (d["left"] is None) == (i == 0) and np.array_equal(d["right"][:, :3], (l - 1500) / 1000.0 + origin)
        assert np.isnan(d["right"][:, 3]).all()


@pytest.mark.parametrize("bad", ["truncate", "group", "ragged", "no_origin"])
def test_malformed_messages_raise(bad):
    pl = _pl([[0, 0, 0], [1000, 0, 0]])
    lane = _ld(2, _ld(1, b"a") + _ld(2, _ld(1, pl)))
    org = _ld(5, _pt([0, 0, 0]))
    if bad == "truncate":
        msg = (lane + org)[:-3]
    elif bad == "group":
        msg = lane + org + _vi(7 << 3 | 3)
    elif bad == "ragged":
        msg = _ld(2, _ld(1, b"a") + _ld(2, _ld(1, _ld(1, _zz([1, 2, 3])) + _ld(2, _zz([1, 2]))))) + org
    else:
        msg = lane
    with pytest.raises(vm.WireError):
        vm.decode_vector_map(msg)


def test_demo_map_decodes_and_agrees_with_its_own_redundancy():
    m = vm.decode_vector_map(_pb())
    L = m["lanes"]
    assert m["name"] == "waymo_train:waymo_train_1" and len(L) == 144 and len(m["others"]["ped_crosswalk"]) == 8
    pts = np.concatenate([l[k][:, :3] for l in L for k in ("center", "left", "right") if l[k] is not None]
                         + [q[:, :3] for q in m["others"]["ped_crosswalk"]])
    # the stored extent (written by trajdata from the un-rounded vertices) against the decoded one
    assert np.abs(pts.max(0) - m["max_pt"]).max() < 0.1 and np.abs(pts.min(0) - m["min_pt"]).max() < 0.1
    assert (m["origin"] == m["min_pt"]).all()
    # stored per-vertex headings against atan2 of the decoded vertex deltas; left/right boundaries on their sides
    dh, side = [], []
    for l in L:
        c = l["center"]
        assert np.isfinite(c).all()
        if len(c) > 2:
            h = np.arctan2(np.diff(c[:, 1]), np.diff(c[:, 0]))
            dh.append(np.abs(np.angle(np.exp(1j * (h - c[:-1, 3])))))
        for k, sgn in (("left", 1), ("right", -1)):
            e = l[k]
            if e is None or len(c) < 2:
                continue
            i = min(len(c) // 2, len(c) - 2)
            t = c[i + 1, :2] - c[i, :2]
            v = e[np.argmin(np.linalg.norm(e[:, :2] - c[i, :2], axis=1)), :2] - c[i, :2]
            side.append(sgn * (t[0] * v[1] - t[1] * v[0]) > 0)
    dh = np.concatenate(dh)
    assert np.median(dh) < 5e-3 and np.quantile(dh, 0.99) < 0.1
    assert np.mean(side) > 0.98 and len(side) > 80
    # lane graph ids refer to lanes of the same map
    ids = {l["id"] for l in L}
    assert all(set(l["entry"]) | set(l["exit"]) | set(l["adj_left"]) | set(l["adj_right"]) <= ids for l in L)


def _lane(xy, left=None, right=None, lid="a"):
    f = lambda p: None if p is None else np.concatenate([np.asarray(p, float), np.zeros((len(p), 2))], 1)
    return {"id": lid, "center": f(xy), "left": f(left), "right": f(right)}


def test_vector_lanes_sampling_clipping_chunking():
    x = np.arange(0.0, 45.0)                                                  # 45 centre vertices, 1 m apart, along +x
    ctr = np.stack([x, np.zeros_like(x)], -1)
    lane = _lane(ctr, left=ctr + [0, 2], right=ctr[:3] - [0, 2])
    v = vm.vector_lanes([lane], frame=(0.0, 0.0, 0.0), tls={"a": 2.0})
    # centre: 45 points -> chunks [0:20], [20:40], [40:45] -> 19 + 19 + 4 segments (no segment across a cut);
    # left edge: every 4th -> 12 points <= 20 -> ONE chunk that drops its last vertex -> 10 segments;
    # right edge: 3 points <= rate 4 -> not subsampled, 3 points -> [0:2] -> 1 segment.   Order: centre, left, right.
    assert v.shape == (5, 19, 6) and v.dtype == np.float32
    assert ((v[..., 4] > 0).sum(1) == [19, 19, 4, 10, 1]).all()
    assert (v[0, :, 0] == np.arange(19)).all() and (v[0, :, 2] == np.arange(1, 20)).all() and v[1, 0, 0] == 20 and v[2, 0, 0] == 40
    assert (v[3, :10, 0] == 4 * np.arange(10)).all() and (v[3, :10, 2] == 4 * np.arange(1, 11)).all() and (v[3, :10, 1] == 2).all()
    assert (np.unique(v[..., 4]) == [0, 1, 2, 3]).all() and (v[..., 5][v[..., 4] > 0] == 2).all() and (v[..., 5][v[..., 4] == 0] == 0).all()
    # the centre agent's frame: rotate the world by -heading about the agent
    r = vm.vector_lanes([lane], frame=(10.0, 0.0, np.pi / 2))
    assert np.allclose(r[0, 0, :4], [0, 10, 0, 9], atol=1e-5) and (r[..., 5][r[..., 4] > 0] == vm.TLS_NO_DATA).all()
    # clipping to the square |x|, |y| < range; a lane with fewer than two points left contributes nothing
    c = vm.vector_lanes([lane], frame=(0.0, 0.0, 0.0), map_range=10.0, include=("center",))
    assert c.shape[0] == 1 and (c[0, :, 4] > 0).sum() == 8                    # x = 0..9 inside -> 10 points -> [0:9] -> 8 segments
    far = vm.vector_lanes([_lane(ctr + [1000.0, 0.0])], frame=(0.0, 0.0, 0.0))
    assert far.shape == (1, 19, 6) and not far.any()


def test_local_vector_map_range_and_cap():
    rng = np.random.default_rng(1)
    M, P = 40, 19
    full = np.zeros((M, P, 6), np.float32)
    n = rng.integers(1, P + 1, M)
    base = rng.uniform(-300, 300, (M, 2)).astype(np.float32)
    for m in range(M):
        full[m, :n[m], 0:2] = base[m] + rng.uniform(-1, 1, (n[m], 2))
        full[m, :n[m], 2:4] = full[m, :n[m], 0:2] + 1
        full[m, :n[m], 4] = 1
    pos = np.stack([full[m, :n[m], :2].mean(0) for m in range(M)])
    d = np.linalg.norm(pos, axis=-1)
    vec, mask = vm.local_vector_map(full, max_points=64)
    near = d < 200
    assert np.array_equal(vec[:near.sum()], full[near]) and not vec[near.sum():].any()
    assert np.array_equal(mask[:near.sum()], full[near][..., 4] > 0) and not mask[near.sum():].any()
    vec8, mask8 = vm.local_vector_map(full, max_points=8)
    order = np.argsort(d[near], kind="stable")[:8]
    assert np.array_equal(vec8, full[near][order])
    assert np.array_equal(mask8, full[near][:8, :, 4] > 0)                    # the reference's mask: taken before the re-ordering


def test_vectors_to_map_frames():
    full = vm.vector_lanes(vm.decode_vector_map(_pb())["lanes"], frame=(429.3, 540.0, 0.3))
    vec, mask = vm.local_vector_map(full)
    mp = vm.vectors_to_map(DEMO_SPEC, vec, mask)
    inp, msk = mp["map_input"][0], mp["map_mask"][0]
    M = inp.shape[0]
    assert 100 < M < 2048 and msk[:, 0].all() and inp.shape[1:] == (19, 11)
    n = msk.sum(1)
    first, last = inp[:, 0, 0:2], inp[np.arange(M), n - 1, 2:4]
    assert np.abs(first + last).max() < 1e-3 and np.abs(last[:, 1]).max() < 1e-3 and (last[:, 0] > 0).all()
    assert np.abs(np.linalg.norm(inp[..., 9:11][msk], axis=-1) - 1).max() < 1e-4
    assert (inp[..., 6:9][msk].sum(-1) == 1).all() and set(np.unique(inp[..., 4][msk])) == {1.0, 2.0, 3.0}
    # back in the scene frame the first start point is the chunk's first vertex
    c, s = np.cos(mp["map_head"][0]), np.sin(mp["map_head"][0])
    wx = first[:, 0] * c - first[:, 1] * s + mp["map_pos"][0, :, 0]
    wy = first[:, 0] * s + first[:, 1] * c + mp["map_pos"][0, :, 1]
    assert np.abs(wx - vec[:M, 0, 0]).max() < 1e-3 and np.abs(wy - vec[:M, 0, 1]).max() < 1e-3


def demo_scene_real_lanes(spec, t0=10, max_agents=16, scene="scene_1"):
    """BASELINE configs[0] on a demo scene's real lanes and agent types, at step t0 in the frame of its ego.  scene_1: a
    small map (144 lanes, no traffic-light records); scene_0: 1056 lanes with traffic-light records -- more chunks in
    range than DATASET.FORMAT.MAP.MAX_POINTS (2048), so the closest 2048 are kept (format_utils.py:168-176)."""
    import lzma
    g = np.load(os.path.join(GOLD, f"demo_{scene}_agent_table.npz"))
    tr = fmt.tracks_from_table({k: g[k] for k in g.files if k != "origin"})
    origin = g["origin"].astype(np.float64)
    f = fmt.ego_frame(tr, t0)
    ego = list(tr["agent_ids"]).index("ego")
    z = float(g["z"][(g["agent_id"] == "ego") & (g["scene_ts"] == t0)][0]) if "z" in g.files else None
    tl = np.load(os.path.join(GOLD, f"demo_{scene}_tls_table.npz"))
    world = np.array([f[0] + origin[0], f[1] + origin[1], f[2]])                # the table is re-centred, the map is not
    if scene == "scene_1":
        pb = _pb()
    else:
        with open(os.path.join(GOLD, "demo_waymo_train_0_map.pb.xz"), "rb") as fh:
            pb = lzma.decompress(fh.read())
    mp = vm.map_for_scene(spec, pb, world, center_z=z, tls=vm.tls_at(tl["lane_id"], tl["scene_ts"], tl["status"], t0))
    present = np.isfinite(tr["x"][:, t0])
    order = [ego] + [i for i in np.nonzero(present)[0] if i != ego]           # the centred agent is agent 0 (format_utils.py:229)
    # the agents' types from the cache's scene metadata (vehicle 1, pedestrian 2, bicycle 3), read without trajdata
    meta = fmt.agent_types_from_scene_metadata(os.path.join(GOLD, f"demo_{scene}_metadata.dill"))
    types = np.array([meta[a] for a in tr["agent_ids"]], np.int64)
    sc = fmt.scene_from_tracks(spec, tr, t0, agents=order, max_agents=max_agents, frame=f, map_fields=mp, agent_types=types)
    sc.pop("agent_ids")
    return sc


def test_big_demo_map_traffic_lights_and_the_2048_chunk_cap():
    sc = demo_scene_real_lanes(DEMO_SPEC.replace(max_steps=20), scene="scene_0")
    inp, msk = sc["map_input"][0], sc["map_mask"][0]
    assert inp.shape == (2048, 19, 11)                                         # 3066 chunks in range: the closest 2048 stay
    tl = inp[..., 5][msk]
    assert set(np.unique(tl)) == {-1.0, 0.0, 1.0, 2.0}                         # no record / unknown / green / red at this step
    # closest first: the kept chunks are ordered by the distance of their mean start point (get_local_vec_map :174-176)
    d = np.linalg.norm(sc["map_pos"][0], axis=-1)
    assert np.median(d[:200]) < np.median(d[-200:]) and d.max() < 230.0
    # the reference takes the point mask from the chunks BEFORE that re-ordering (:168-171): kept as it is there, so some
    # rows' masks belong to another chunk -- a mask bit over a padding segment carries the frame-shifted zero row
    own = inp[..., 4] > 0
    assert (own != msk).any() and msk.any(1).all()


def demo_rollout_batch(spec, t0=10, n_policy=12, scene="scene_1", conditions=True):
    """A whole 80-step rollout input from the demo cache: the ego and the next agents that stay in the scene for all 80
    steps are policy agents, every other agent that is there at some replan replays its log -- some leave, some enter."""
    import lzma
    g = np.load(os.path.join(GOLD, f"demo_{scene}_agent_table.npz"))
    tr = fmt.tracks_from_table({k: g[k] for k in g.files if k != "origin"})
    origin = g["origin"].astype(np.float64)
    f = fmt.ego_frame(tr, t0)
    ego = list(tr["agent_ids"]).index("ego")
    pres = np.isfinite(tr["x"]) & np.isfinite(tr["heading"])
    ever = [i for i in range(pres.shape[0]) if any(pres[i, t0 + t] for t in spec.all_t_indices)]
    stay = [i for i in ever if pres[i, t0:t0 + spec.max_steps + 1].all()]
    policy = [ego] + [i for i in stay if i != ego][:n_policy - 1]
    replay = [i for i in ever if i not in policy]
    tl = np.load(os.path.join(GOLD, f"demo_{scene}_tls_table.npz"))
    world = np.array([f[0] + origin[0], f[1] + origin[1], f[2]])
    if scene == "scene_1":
        pb = _pb()
    else:
        with open(os.path.join(GOLD, "demo_waymo_train_0_map.pb.xz"), "rb") as fh:
            pb = lzma.decompress(fh.read())
    mp = vm.map_for_scene(spec, pb, world, tls=vm.tls_at(tl["lane_id"], tl["scene_ts"], tl["status"], t0))
    meta = fmt.agent_types_from_scene_metadata(os.path.join(GOLD, f"demo_{scene}_metadata.dill"))
    types = np.array([meta[a] for a in tr["agent_ids"]], np.int64)
    sc = fmt.rollout_batch_from_tracks(spec, tr, t0, policy, replay, frame=f, map_fields=mp, agent_types=types)
    sc.pop("agent_ids")
    if conditions:   # the log-derived goal and drag-point prompts of the policy agents
        sc["cond"] = fmt.conditions_from_tracks(spec, tr, t0, policy + replay, sc["prompt_mask"][0])
    return sc, tr, policy, replay


def test_rollout_batch_from_tracks_log_replay_on_real_data():
    spec = SMALL_SPEC
    sc, tr, policy, replay = demo_rollout_batch(spec)
    R, N = spec.n_replans, len(policy) + len(replay)
    assert sc["fut_obs_input"].shape == (R - 1, 1, N, spec.hist_steps, spec.obs_dim) and sc["prompt_mask"].sum() == len(policy)
    seen0 = sc["obs_mask"][0].all(-1).any(-1)
    seen = sc["fut_obs_mask"][:, 0].all(-1).any(-1)                            # [R - 1, N]
    assert seen0[:len(policy)].all() and seen[:, :len(policy)].all()           # policy agents are listed in every frame
    enters = ~seen0 & seen.any(0)
    leaves = seen0 & ~seen[-1]
    assert enters.sum() >= 5 and leaves.sum() >= 5                             # the real log has both kinds
    # a frame is the agent's own 11-step window ending at that replan: its last step sits at the origin of its frame,
    # and its pose in the scene frame is where the table says the agent is at t0 + 10 k
    k, j = 2, int(np.nonzero(seen[2])[0][-1])
    assert np.abs(sc["fut_obs_input"][k, 0, j, -1, 0:2]).max() == 0
    row = (policy + replay)[j]
    f = fmt.ego_frame(tr, 10)
    dx, dy = tr["x"][row, 10 + spec.all_t_indices[k + 1]] - f[0], tr["y"][row, 10 + spec.all_t_indices[k + 1]] - f[1]
    c, s_ = np.cos(-f[2]), np.sin(-f[2])
    assert np.allclose(sc["fut_obs_pos"][k, 0, j], [dx * c - dy * s_, dx * s_ + dy * c], atol=1e-3)
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o = orc.rollout(w, spec, sc)
    assert o["traj"].shape == (1, N, spec.max_steps, 4) and torch.isfinite(o["traj"][0, :len(policy)]).all()
    # the log-derived conditions: the goal is the end of the agent's logged future in its frame at t0, the drag points its path
    cg, cd = sc["cond"]["goal"], sc["cond"]["drag_point"]
    assert cg["mask"][0, :len(policy)].all() and not cg["mask"][0, len(policy):].any() and (cg["input"][0, :len(policy), 2] == spec.max_steps).all()
    assert cd["input"].shape == (1, N, spec.max_steps // 5, 2) and cd["mask"][0, :len(policy)].all()
    row1 = policy[1]
    gx, gy = tr["x"][row1, 90] - tr["x"][row1, 10], tr["y"][row1, 90] - tr["y"][row1, 10]
    c1, s1 = np.cos(-tr["heading"][row1, 10]), np.sin(-tr["heading"][row1, 10])
    assert np.allclose(cg["input"][0, 1, :2], [gx * c1 - gy * s1, gx * s1 + gy * c1], atol=1e-3)
    # the metric's ground truth from the same table: local targets per replan, gaps where an agent has left the log
    gt = fmt.pair_targets_from_tracks(spec, tr, 10, policy + replay)
    assert gt["tgt"].shape == (1, R, N, spec.target_steps, 5) and gt["mask"][0, :, :len(policy)].all()
    assert not gt["mask"][0, -1, len(policy):].all() and np.isnan(gt["tgt"][0][~gt["mask"][0]]).all()
    # chained back to the scene frame a policy agent's targets ARE its logged path: the first target step is the log's
    # next state in the agent's frame at t0
    j, row = 1, policy[1]
    dx, dy = tr["x"][row, 11] - tr["x"][row, 10], tr["y"][row, 11] - tr["y"][row, 10]
    c, s_ = np.cos(-tr["heading"][row, 10]), np.sin(-tr["heading"][row, 10])
    assert np.allclose(gt["tgt"][0, 0, j, 0, :2], [dx * c - dy * s_, dx * s_ + dy * c], atol=1e-4)


def test_scene_metadata_is_read_without_trajdata():
    meta = fmt.agent_types_from_scene_metadata(os.path.join(GOLD, "demo_scene_1_metadata.dill"))
    tr, _, _ = _tracks()
    assert set(meta) == set(tr["agent_ids"].tolist()) and meta["ego"] == 1
    vals = np.array(list(meta.values()))
    assert (vals == 1).sum() == 50 and (vals == 2).sum() == 2 and (vals == 3).sum() == 6     # vehicles, pedestrians, bicycles
    # the types line up with the tracks they name: pedestrians are the short ones, vehicles the long ones
    ln = {a: float(np.nanmedian(tr["length"][i])) for i, a in enumerate(tr["agent_ids"])}
    assert max(ln[a] for a, t in meta.items() if t == 2) < 1.5 < np.median([ln[a] for a, t in meta.items() if t == 1])


def test_real_lane_scene_rolls_out_and_is_frame_invariant():
    spec = SMALL_SPEC.replace(max_steps=20)
    sc = demo_scene_real_lanes(spec, max_agents=24)
    assert set(np.unique(sc["agent_type"])) >= {1, 3}                              # more than one type among the first 24
    assert sc["map_input"].shape[1] > 300 and np.abs(sc["obs_pos"][0, 0]).max() < 1e-4 and abs(sc["obs_head"][0, 0]) < 1e-6
    # agents drive on the lanes: every moving agent is within a lane width of some centre-line chunk
    d = np.linalg.norm(sc["obs_pos"][0][:, None] - sc["map_pos"][0][None], axis=-1).min(1)
    assert np.median(d) < 6.0
    w = weights.init_weights(spec, 0)
    # the model sees relative poses only: the same scene reported in another frame rolls out the same local motion
    th, off = 0.7, np.array([30.0, -12.0], np.float32)
    rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]], np.float32)
    sc2 = dict(sc)
    for k in ("obs", "map"):
        sc2[k + "_pos"] = sc[k + "_pos"] @ rot.T + off
        sc2[k + "_head"] = sc[k + "_head"] + np.float32(th)
    with torch.no_grad():
        a = orc.rollout(w, spec, sc, dtype=torch.float64)
        b = orc.rollout(w, spec, sc2, dtype=torch.float64)
    assert a["traj"].shape == (1, 24, 20, 4) and torch.isfinite(a["traj"]).all()
    # (rows of replan 0: later replans start from poses that already carry the float32 rounding of the moved frame)
    assert (a["motion_pred"][:24] - b["motion_pred"][:24]).abs().max() < 2e-4
