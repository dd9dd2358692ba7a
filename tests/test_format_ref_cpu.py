"""prosim_amd/formatting.py and prosim_amd/vecmap.py against outputs of the REFERENCE's own formatters
(tests/golden/ref_format_scene_{0,1}.npz, written by tests/gen_golden.py:gen_format from prosim/dataset/format_utils.py,
data_utils.py and prompt_utils.py run on a duck-typed SceneBatch of the demo cache's agent table and on lane objects of the
map that vecmap.py decodes; trajdata's StateTensor / arr_utils helpers are builder stand-ins: 'ref + trajdata stand-ins').
Same inputs (gen_golden.format_inputs), so every difference is the formatter's."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden_cases import format_inputs, GOLD  # noqa: E402
from prosim_amd import formatting as fmt, vecmap as vm  # noqa: E402
from prosim_amd.spec import DEMO_SPEC  # noqa: E402

T0 = 10


def _fix(scene):
    return np.load(os.path.join(GOLD, f"ref_format_{scene}.npz"))


def _same(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    return float(np.nanmax(np.abs(a - b))) < tol if a.size else True


@pytest.mark.parametrize("scene", ["scene_1", "scene_0"])
def test_center_obs_and_prompt_vs_reference(scene):
    """get_center_obs (format_utils.py:357-447) + AgentStatusGenerator.prompt_for_scene_batch (prompt_utils.py:111-150)."""
    g, inp = _fix(scene), format_inputs(scene, T0)
    tr, order = inp["tracks"], inp["order"]
    ids = [str(tr["agent_ids"][i]) for i in order]
    rows = [order[ids.index(a)] for a in g["obs_ids"]]                       # the reference lists the agents present at t0
    sc = fmt.scene_from_tracks(DEMO_SPEC, tr, T0, agents=rows, agent_types=inp["types"], map_polylines=[np.zeros((2, 2)) + [[0, 0], [1, 0]]])
    assert list(sc["agent_ids"]) == list(g["obs_ids"])
    assert np.array_equal(sc["obs_mask"], g["obs_mask"])
    assert _same(np.where(sc["obs_mask"], sc["obs_input"], np.nan), np.where(g["obs_mask"], g["obs_input"], np.nan), 2e-5)
    assert _same(sc["obs_pos"], g["obs_pos"], 1e-5) and _same(sc["obs_head"], g["obs_head"], 1e-6)
    # the prompt rows of the reference are its target agents (present at t0), in batch order
    sel = [list(sc["agent_ids"]).index(a) for a in g["prompt_ids"]]
    assert _same(sc["prompt"][0][sel], g["prompt"][0], 2e-5)
    assert _same(sc["obs_pos"][0][sel], g["prompt_pos"][0], 1e-5) and _same(sc["obs_head"][0][sel], g["prompt_head"][0, :, 0], 1e-6)
    assert np.array_equal(sc["agent_type"][0][sel], g["prompt_type"][0])


@pytest.mark.parametrize("scene", ["scene_1", "scene_0"])
def test_future_obs_frames_vs_reference(scene):
    """get_future_obs (format_utils.py:667-687, FUTURE_OBS_TYPE 'latest'): per replan the agents in the scene then, in their own frames."""
    g, inp = _fix(scene), format_inputs(scene, T0)
    tr, order = inp["tracks"], inp["order"]
    ids = [str(tr["agent_ids"][i]) for i in order]
    for k in g["fut_keys"]:
        want_ids = list(g[f"fut{k}_ids"])
        rows = [order[ids.index(a)] for a in want_ids]
        # target agents stay listed even when they have left (NaN origin): keep_absent keeps their (masked) slots
        sc = fmt.scene_from_tracks(DEMO_SPEC, tr, T0 + int(k), agents=rows, agent_types=inp["types"], keep_absent=True,
                                   map_polylines=[np.zeros((2, 2)) + [[0, 0], [1, 0]]])
        m_ref = g[f"fut{k}_mask"]
        gone = ~np.isfinite(tr["x"][rows, T0 + int(k)])
        # (an agent without a state at that step: the reference leaves NaN history and all-False state mask but sets the
        # extent / type / time columns valid; the engine takes 'every feature valid' as the point mask, so both agree on it)
        assert np.array_equal(sc["obs_mask"][0][~gone], m_ref[0][~gone])
        assert not sc["obs_mask"][0][gone].all(-1).any() and not m_ref[0][gone].all(-1).any()
        a = np.where(sc["obs_mask"], sc["obs_input"], np.nan)[0][~gone]
        b = np.where(m_ref, g[f"fut{k}_input"], np.nan)[0][~gone]
        assert _same(a, b, 2e-5)
        assert _same(sc["obs_pos"][0][~gone], g[f"fut{k}_pos"][0][~gone], 1e-5)
        assert _same(sc["obs_head"][0][~gone], g[f"fut{k}_head"][0][~gone], 1e-6)


@pytest.mark.parametrize("scene", ["scene_1", "scene_0"])
def test_pair_targets_and_conditions_vs_reference(scene):
    """get_local_io_pairs_T_step_batch (format_utils.py:498-616): the metric's targets, the goal, full_traj_xy (what the
    goal / drag-point conditions are cut from, condition_utils.py:126-175, :401-447)."""
    g, inp = _fix(scene), format_inputs(scene, T0)
    tr, order = inp["tracks"], inp["order"]
    ids = [str(tr["agent_ids"][i]) for i in order]
    rows = [order[ids.index(a)] for a in g["io_names"]]
    pt = fmt.pair_targets_from_tracks(DEMO_SPEC, tr, T0, rows)
    assert np.array_equal(pt["mask"], g["io_mask"])
    assert _same(np.where(pt["mask"][..., None, None], pt["tgt"], np.nan), np.where(g["io_mask"][..., None, None], g["io_tgt"], np.nan), 2e-5)
    cd = fmt.conditions_from_tracks(DEMO_SPEC, tr, T0, rows, np.ones(len(rows), bool))
    goal_ref = g["io_goal"][0, 0]                                             # replan 0: the last logged future position, agent frame at t0
    ok = g["io_mask"][0, 0] & cd["goal"]["mask"][0]
    assert ok.sum() >= len(rows) - 2
    assert _same(cd["goal"]["input"][0, ok, :2], goal_ref[ok], 2e-5)
    # drag points: every 5th step of full_traj_xy (condition_utils.py:415-430 takes it at POINT_SAMPLE_RATE 5, padded to 80 steps with NaN)
    dp_ref = g["io_full_traj_xy"][0][:, ::5]
    assert cd["drag_point"]["input"].shape[2] == dp_ref.shape[1] == 16
    assert _same(cd["drag_point"]["input"][0][cd["drag_point"]["mask"][0]], dp_ref[cd["drag_point"]["mask"][0]], 2e-5)


@pytest.mark.parametrize("scene", ["scene_1", "scene_0"])
def test_vector_lanes_and_map_frames_vs_reference(scene):
    """_get_vectorized_lanes_from_vector_map (data_utils.py:155-271), get_local_vec_map + local_map_to_sym_coord +
    get_center_vec_init_map (format_utils.py:150-263) on the same decoded lanes."""
    g, inp = _fix(scene), format_inputs(scene, T0)
    full = vm.vector_lanes(inp["lanes"], inp["world"], center_z=inp["z"], tls=inp["tls"])
    assert full.shape == g["vector_lane"].shape
    assert np.abs(full - g["vector_lane"]).max() < 2e-3                      # float32 of coordinates up to 200 m, transformed in float64 on both sides
    vec, mask = vm.local_vector_map(g["vector_lane"])
    mp = vm.vectors_to_map(DEMO_SPEC, vec, mask, drop_padding=False)
    assert np.array_equal(mp["map_mask"], g["map_mask"])
    m = g["map_mask"]
    # With more than MAX_POINTS chunks in range the reference keeps the closest by torch.argsort, which is not a stable
    # sort: chunks at EQUAL distance (the left edge of one lane lying on the right edge of its neighbour) may come out in
    # either order.  Rows are therefore compared as a multiset: every row of ours (whole content, also under the mask bits
    # that the reference takes from the unsorted chunks) must be a row of the reference's, and row by row wherever no tie
    # is involved.
    full = vm.vectors_to_map(DEMO_SPEC, vec, np.ones_like(mask), drop_padding=False)
    ours = np.concatenate([full["map_input"][0].reshape(2048, -1), full["map_pos"][0], full["map_head"][0][:, None]], -1)
    ref = np.concatenate([g["map_input"][0].reshape(2048, -1), g["map_pos"].reshape(2048, 2), g["map_head"].reshape(2048, 1)], -1)
    same = np.abs(ours - ref).max(-1) < 1e-4
    assert same.mean() > 0.9
    bad = np.nonzero(~same)[0]
    free, cut = set(bad.tolist()), 0
    for i in bad:   # every row of ours that is not in place sits at the position of a tied neighbour in the reference
        cand = np.array(sorted(free))
        d = np.abs(ref[cand] - ours[i]).max(-1)
        if d.min() >= 1e-4:   # a tie cut in two by the MAX_POINTS limit: the other edge of the same geometry was kept
            geo = np.array([c for c in range(ours.shape[1]) if c >= 209 or c % 11 in (0, 1, 2, 3, 9, 10)])
            d = np.abs(ref[cand][:, geo] - ours[i][geo]).max(-1)
            cut += 1
        assert d.min() < 1e-4, (i, d.min())
        free.discard(int(cand[int(np.argmin(d))]))
    assert cut <= 2
    if scene == "scene_1":   # (fewer chunks than MAX_POINTS: no sort, no ties)
        assert same.all()
    used = m.any(-1)[0] & same
    assert np.abs(np.where(m[..., None], mp["map_input"] - g["map_input"], 0)[0][same]).max() < 1e-4
    # (local_map_to_sym_coord leaves a broadcast axis in: position [B, M, 1, 2], heading [B, M, 1])
    assert np.abs((mp["map_pos"][0] - g["map_pos"].reshape(-1, 2))[used]).max() < 1e-5
    assert np.abs((mp["map_head"][0] - g["map_head"].reshape(-1))[used]).max() < 1e-6


def test_scene_metadata_pickle_is_allow_listed(tmp_path):
    """A cache directory is user input: the Scene unpickler refuses every global that a Scene does not need."""
    import os as _os
    import pickle

    class Evil:
        def __reduce__(self):
            return (_os.system, ("true",))

    p = tmp_path / "evil.dill"
    p.write_bytes(pickle.dumps(Evil()))
    with pytest.raises(pickle.UnpicklingError):
        fmt.agent_types_from_scene_metadata(str(p))
    assert len(fmt.agent_types_from_scene_metadata(os.path.join(GOLD, "demo_scene_1_metadata.dill"))) > 10
