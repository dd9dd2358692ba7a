"""Agent tracks -> the rollout path's input batch, without trajdata.

The reference builds ``init_obs / init_map / prompt`` from trajdata ``SceneBatch`` objects
(``prosim/dataset/format_utils.py``); trajdata is a third-party fork that is absent from the tree,
so this module restates the arithmetic of those formatters on plain arrays:

* ``tracks_from_table``     -- the demo cache's per-scene Arrow table (``agent_data_dt0.10.feather``:
                               agent_id, scene_ts, x, y, vx, vy, ax, ay, heading, length, width) -> NaN-padded
                               ``[agent][t]`` arrays;
* ``scene_from_tracks``     -- ``get_center_obs`` (format_utils.py:357-447): the 11-step history of every agent
                               in ITS OWN frame at the current step (offset by the pose at t0, rotated by
                               -heading), NaN + mask for missing steps, extent / type / time one-hots, and the
                               status prompt ``AgentStatusGenerator.prompt_for_scene_batch``
                               (prompt_utils.py:111-150): [v_local(2), extent(2), type one-hot(3)];
* ``rollout_batch_from_tracks`` -- the above plus ``get_future_obs`` (format_utils.py:667-687): the observation frames of the later
                               replans for policy and log-replay agents (agents that leave or enter keep their slots);
* ``pair_targets_from_tracks`` -- ``get_local_io_pairs_T_step_batch`` (format_utils.py:498-616): the metric's ground truth, per replan
                               the next target_steps logged states in the agent's frame at that replan, NaN gaps kept;
* ``conditions_from_tracks`` -- the log-derived goal and drag-point conditions (condition_utils.py:126-175, :401-447);
* ``agent_types_from_scene_metadata`` -- the cache's pickled ``Scene`` -> agent id -> type, read without trajdata;
* ``polylines_to_map``      -- ``local_map_to_sym_coord`` + ``get_center_vec_init_map`` (format_utils.py:184-263):
                               per-polyline frame = midpoint / tangent of (first start, last valid end), segments in
                               that frame, type one-hot, unit direction;
* ``lanes_from_tracks``     -- NOT in the reference: a stand-in that draws lane polylines along the paths the agents
                               drove, for tables that come without a map.  The demo scenes' real lanes (trajdata VectorMap
                               protobufs) are decoded by ``prosim_amd/vecmap.py`` and passed in as ``map_fields``.

``transform_to_frame_offset_rot`` (trajdata, absent) is restated from its name and call site: positions are
offset and rotated into the frame, velocities / accelerations are rotated, headings are made relative --
parity unpinned, like the rest of the trajdata boundary.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from .spec import ModelSpec

TRACK_COLS = ("x", "y", "vx", "vy", "ax", "ay", "heading", "length", "width")


def tracks_from_table(cols: Dict[str, np.ndarray], n_steps: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Columns of the per-scene agent table -> ``{'agent_ids': [N], name: [N, T] float64 (NaN = absent)}``.
    Agents are ordered by id (string order, as the cache stores them)."""
    ids = np.asarray(cols["agent_id"]).astype(str)
    ts = np.asarray(cols["scene_ts"]).astype(np.int64)
    uniq = sorted(set(ids.tolist()))
    index = {a: i for i, a in enumerate(uniq)}
    T = int(n_steps if n_steps is not None else ts.max() + 1)
    rows = np.array([index[a] for a in ids])
    keep = ts < T
    out = {"agent_ids": np.array(uniq)}
    for c in TRACK_COLS:
        arr = np.full((len(uniq), T), np.nan, np.float64)
        arr[rows[keep], ts[keep]] = np.asarray(cols[c], np.float64)[keep]
        out[c] = arr
    return out


def agent_types_from_scene_metadata(path: str) -> Dict[str, int]:
    """The cache's ``scene_metadata_dt*.dill`` (a pickled trajdata ``Scene``) -> {agent id: AgentType value} WITHOUT trajdata:
    the pickle is read with structural stand-ins for every ``trajdata.*`` class (they keep their state dict, an enum keeps
    its value).  trajdata's AgentType: 0 unknown, 1 vehicle, 2 pedestrian, 3 bicycle, 4 motorcycle; the reference's model
    types are 1 vehicle, 2 pedestrian, 3 cyclist (DATASET.USE_PED_CYCLIST; prompt / observation one-hots over 1..3)."""
    import pickle

    class _Stub:
        def __init__(self, *a, **k):
            self._args = a
            self.__dict__.update(k)

        def __setstate__(self, st):
            if isinstance(st, dict):
                self.__dict__.update(st)
            else:
                self._state = st

    # Globals a Scene pickle of the cache needs besides the trajdata classes (measured over the 16 demo scenes).  Anything
    # else -- dill's _create_function / _import_module, os, builtins.eval ... -- is refused: a cache directory is user input.
    allowed = {("collections", "defaultdict"), ("collections", "OrderedDict"), ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"),
               ("pathlib", "Path"), ("numpy", "dtype"), ("numpy", "ndarray"), ("numpy.core.multiarray", "scalar"),
               ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"),
               ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple")}

    def _load_type(name):   # dill._dill._load_type: a builtin TYPE by name; only plain containers / scalars are honoured
        import builtins
        if name not in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "str", "bool", "bytes", "NoneType", "type", "PartialType"):
            raise pickle.UnpicklingError(f"scene metadata pickle asks for type {name!r}: refused")
        if name == "PartialType":   # the Scene holds defaultdict(partial(const_lambda, ...)); it can only wrap what find_class lets through
            import functools
            return functools.partial
        return type(None) if name == "NoneType" else getattr(builtins, name)

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module.split(".")[0] == "trajdata":
                return type(name, (_Stub,), {"__module__": module})
            if (module, name) == ("dill._dill", "_load_type"):            # (the cache is written with dill)
                return _load_type
            if (module, name) in allowed:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"scene metadata pickle references {module}.{name}: refused (not on the allow-list)")

    with open(path, "rb") as f:
        scene = _Unpickler(f).load()
    out = {}
    for a in scene.__dict__["agents"]:
        t = a.__dict__["type"]
        out[str(a.__dict__["name"])] = int(t._args[0]) if getattr(t, "_args", None) else int(getattr(t, "_state", 0) or 0)
    return out


def _rotate(x, y, ang):
    c, s = np.cos(ang), np.sin(ang)
    return x * c - y * s, x * s + y * c


def scene_from_tracks(spec: ModelSpec, tracks: Dict[str, np.ndarray], t0: int,
                      map_polylines: Optional[Sequence[np.ndarray]] = None, agent_types: Optional[np.ndarray] = None,
                      max_agents: Optional[int] = None, points: int = 19,
                      agents: Optional[Sequence[int]] = None, frame: Optional[Sequence[float]] = None,
                      map_fields: Optional[Dict[str, np.ndarray]] = None, keep_absent: bool = False) -> Dict[str, np.ndarray]:
    """One scene (batch 1) at current step ``t0``.  Agents whose state at ``t0`` is not finite are dropped
    (get_center_obs skips non-target agents with a NaN origin, :383-388).  ``agents``: row indices of the track
    table to consider (default: all), ``max_agents``: keep the first so many of those that are present.
    ``frame`` = (x, y, heading) of the centre agent at ``t0`` (``ego_frame``): agent poses are reported in that frame, as
    the scene-centric batch of the reference is (its obs positions are relative to the centred agent); default: the
    table's own frame.  ``map_fields``: the map_* entries made by ``prosim_amd.vecmap`` in the SAME frame (the scene's
    real lanes); without them (and without ``map_polylines``) lanes are drawn along the driven paths.
    ``keep_absent``: keep the slots of ``agents`` that are not in the scene at ``t0`` (fully masked rows) instead of dropping
    them -- the frames of ``rollout_batch_from_tracks`` need the same slots at every replan."""
    H = spec.hist_steps
    if t0 < 0 or t0 >= tracks["x"].shape[1]:
        raise ValueError("t0 outside the track table")
    valid0 = np.isfinite(tracks["x"][:, t0]) & np.isfinite(tracks["y"][:, t0]) & np.isfinite(tracks["heading"][:, t0])
    sel = np.nonzero(valid0)[0] if agents is None else np.array([i for i in agents if valid0[i] or keep_absent], np.int64)
    if max_agents is not None:
        sel = sel[:max_agents]
    N = len(sel)
    if N == 0:
        raise ValueError("no agent is present at t0")
    f32 = np.float32
    lo = t0 - H + 1
    win = np.arange(lo, t0 + 1)
    inside = win >= 0

    def window(name):
        w = np.full((N, H), np.nan)
        w[:, inside] = tracks[name][sel][:, win[inside]]
        return w

    x, y, h = window("x"), window("y"), window("heading")
    vx, vy, ax, ay = window("vx"), window("vy"), window("ax"), window("ay")
    x0, y0, h0 = x[:, -1:], y[:, -1:], h[:, -1:]
    rx, ry = _rotate(x - x0, y - y0, -h0)
    rvx, rvy = _rotate(vx, vy, -h0)
    rax, ray = _rotate(ax, ay, -h0)
    rh = h - h0
    obs = np.full((1, N, H, spec.obs_dim), np.nan, np.float64)
    obs[0, :, :, 0], obs[0, :, :, 1] = rx, ry
    obs[0, :, :, 2], obs[0, :, :, 3] = np.sin(rh), np.cos(rh)
    obs[0, :, :, 4], obs[0, :, :, 5] = rvx, rvy
    obs[0, :, :, 6], obs[0, :, :, 7] = rax, ray
    # extent: the largest finite (length, width) the agent ever reports (get_all_agent_data :318-322)
    ext = np.stack([np.nanmax(np.where(np.isfinite(tracks[c][sel]), tracks[c][sel], -1.0), axis=1) for c in ("length", "width")], -1)
    obs[0, :, :, 8:10] = ext[:, None, :]
    types = np.ones(N, np.int64) if agent_types is None else np.asarray(agent_types, np.int64)[sel]
    for tid in (1, 2, 3):
        obs[0, :, :, 10 + tid - 1] = (types == tid)[:, None]
    obs[0, :, :, 13:13 + H] = np.eye(H)
    mask = np.isfinite(obs)
    prompt = np.zeros((1, N, spec.prompt_dim), f32)
    prompt[0, :, 0], prompt[0, :, 1] = np.nan_to_num(rvx[:, -1]), np.nan_to_num(rvy[:, -1])
    # (the prompt carries the extent AT t0 -- prompt_utils.py:68 reads agent_hist_extent[..., -1, :2] --, the observation the
    # largest one the agent ever reports; found by the reference-made fixture, tests/test_format_ref_cpu.py)
    prompt[0, :, 2:4] = np.nan_to_num(np.stack([tracks["length"][sel, t0], tracks["width"][sel, t0]], -1))
    for tid in (1, 2, 3):
        prompt[0, :, 4 + tid - 1] = types == tid
    px, py, ph = x0[:, 0], y0[:, 0], h0[:, 0]
    if frame is not None:
        px, py = _rotate(px - frame[0], py - frame[1], -frame[2])
        ph = ph - frame[2]
    if keep_absent:   # an absent agent has no pose: keep the arrays finite, its mask says it is not there
        px, py, ph = np.nan_to_num(px), np.nan_to_num(py), np.nan_to_num(ph)
    scene = dict(obs_input=obs.astype(f32), obs_mask=mask, obs_pos=np.stack([px, py], -1)[None].astype(f32),
                 obs_head=ph[None].astype(f32), prompt=prompt, prompt_mask=valid0[sel][None].copy(),
                 agent_type=types[None], agent_ids=tracks["agent_ids"][sel])
    if map_fields is not None:
        scene.update({k: map_fields[k] for k in ("map_input", "map_mask", "map_pos", "map_head")})
        return scene
    if map_polylines is None:
        if frame is not None:
            raise ValueError("lanes drawn along the tracks are in the table's frame: pass frame=None or real lanes")
        map_polylines = lanes_from_tracks(tracks, points=points)
    scene.update(polylines_to_map(spec, map_polylines, points=points))
    return scene


def rollout_batch_from_tracks(spec: ModelSpec, tracks: Dict[str, np.ndarray], t0: int, policy: Sequence[int],
                              replay: Sequence[int] = (), **kw) -> Dict[str, np.ndarray]:
    """The whole input of a closed-loop rollout from a track table: ``scene_from_tracks`` at ``t0`` plus the later
    replans' observation frames ``fut_obs_*`` [R - 1, 1, N, ...] (``get_future_obs``, format_utils.py:667-687, FUTURE_OBS_TYPE
    'latest': the history window that ends at step t0 + t of every agent, in its own frame there -- ``get_center_obs`` again).
    ``policy``: table rows the simulation drives (present at ``t0``); ``replay``: rows that follow their log -- they may
    leave (masked frames from then on) or enter later (no history at ``t0``, listed from their first frame on).  ``kw`` goes
    to ``scene_from_tracks`` (frame, map_fields, agent_types)."""
    rows = list(policy) + [r for r in replay if r not in set(policy)]
    sc = scene_from_tracks(spec, tracks, t0, agents=rows, keep_absent=True, **kw)
    n_pol = len(list(policy))
    if not sc["prompt_mask"][0, :n_pol].all():
        raise ValueError("a policy agent is not in the scene at t0")
    sc["prompt_mask"][0, n_pol:] = False
    T = tracks["x"].shape[1]
    frames = []
    for t in spec.all_t_indices[1:]:
        if t0 + t >= T:
            raise ValueError("the track table ends before the last replan")
        frames.append(scene_from_tracks(spec, tracks, t0 + t, agents=rows, keep_absent=True, **kw))
    if frames:
        sc["fut_obs_input"] = np.stack([f["obs_input"] for f in frames])
        sc["fut_obs_mask"] = np.stack([f["obs_mask"] for f in frames])
        sc["fut_obs_pos"] = np.stack([f["obs_pos"] for f in frames])
        sc["fut_obs_head"] = np.stack([f["obs_head"] for f in frames])
    return sc


def pair_targets_from_tracks(spec: ModelSpec, tracks: Dict[str, np.ndarray], t0: int, rows: Sequence[int]) -> Dict[str, np.ndarray]:
    """The ground truth of the validation metric from a track table: ``io_pairs_batch['tgt' | 'mask']``
    (``get_local_io_pairs_T_step_batch``, format_utils.py:498-616, TAIL_PADDING, SAMPLE_RATE = replan_freq).  Per replan t
    and agent: the next ``target_steps`` logged states (x, y, h, xd, yd) in the frame of the agent's logged state at
    t0 + t (offset, rotate by -heading; velocities rotated, heading relative), NaN where the log has no state; the pair is
    valid iff that centre state is complete and the target is not all NaN.  Returns tgt [1, R, N, S, 5] float32 and
    mask [1, R, N]."""
    rows = np.asarray(list(rows), np.int64)
    R, S, N, T = spec.n_replans, spec.target_steps, len(rows), tracks["x"].shape[1]
    tgt = np.full((1, R, N, S, 5), np.nan, np.float64)
    mask = np.zeros((1, R, N), bool)
    cols = ("x", "y", "heading", "vx", "vy", "ax", "ay")
    for r, t in enumerate(spec.all_t_indices):
        tc = t0 + t
        if tc >= T:
            continue
        ctr = {c: tracks[c][rows, tc] for c in cols}
        ok = np.all([np.isfinite(ctr[c]) for c in cols], 0)                    # HISTORY.ELEMENTS x,y,s,c,xd,yd,xdd,ydd all there
        steps = np.arange(tc + 1, tc + 1 + S)
        inside = steps < T
        fut = {c: np.full((N, S), np.nan) for c in ("x", "y", "heading", "vx", "vy")}
        for c in fut:
            fut[c][:, inside] = tracks[c][rows][:, steps[inside]]
        lx, ly = _rotate(fut["x"] - ctr["x"][:, None], fut["y"] - ctr["y"][:, None], -ctr["heading"][:, None])
        lvx, lvy = _rotate(fut["vx"], fut["vy"], -ctr["heading"][:, None])
        lh = fut["heading"] - ctr["heading"][:, None]
        lh = (lh + np.pi) % (2 * np.pi) - np.pi
        tg = np.stack([lx, ly, lh, lvx, lvy], -1)
        valid = ok & ~np.isnan(tg).all(-1).all(-1)
        tgt[0, r, valid] = tg[valid]
        mask[0, r] = valid
    return dict(tgt=tgt.astype(np.float32), mask=mask)


def conditions_from_tracks(spec: ModelSpec, tracks: Dict[str, np.ndarray], t0: int, rows: Sequence[int],
                           policy_mask: np.ndarray, drag_rate: int = 5) -> Dict[str, Dict[str, np.ndarray]]:
    """The log-derived prompt conditions of the reference, for the slots ``rows`` (one condition row per slot, valid on the
    policy slots): ``goal`` (get_goal_condition_batch, condition_utils.py:126-175: the last logged position of the agent's
    future in its frame at ``t0`` -- GOAL.LOCAL --, and the number of future steps it is there) and ``drag_point``
    (get_drag_points_condition_batch :401-447: every ``drag_rate``-th step of the agent's future path in that frame;
    the reference then keeps a random consecutive subset and adds noise -- training-time randomisation, left out here:
    all valid points, no noise).  Futures span ``max_steps`` steps (the ROLLOUT split)."""
    rows = np.asarray(list(rows), np.int64)
    N, T = len(rows), tracks["x"].shape[1]
    pm = np.asarray(policy_mask, bool).reshape(N)
    steps = np.arange(t0 + 1, t0 + 1 + spec.max_steps)                       # a table that ends early: NaN steps, as the
    inside = steps < T                                                       # reference pads full_traj_xy (format_utils.py:625-633)
    fx, fy = np.full((N, len(steps)), np.nan), np.full((N, len(steps)), np.nan)
    fx[:, inside], fy[:, inside] = tracks["x"][rows][:, steps[inside]], tracks["y"][rows][:, steps[inside]]
    x0, y0, h0 = tracks["x"][rows, t0][:, None], tracks["y"][rows, t0][:, None], tracks["heading"][rows, t0][:, None]
    lx, ly = _rotate(fx - x0, fy - y0, -h0)                                   # [N, F] the future path in the agent's frame at t0
    ok = np.isfinite(lx) & np.isfinite(ly)
    fut_len = np.where(ok.any(1), ok.shape[1] - np.argmax(ok[:, ::-1], 1), 0)  # steps up to the last logged one
    goal = np.zeros((1, N, 3), np.float32)
    gmask = pm & (fut_len > 0)
    idx = np.clip(fut_len - 1, 0, None)
    goal[0, :, 0] = np.where(gmask, np.nan_to_num(lx[np.arange(N), idx]), 0.0)
    goal[0, :, 1] = np.where(gmask, np.nan_to_num(ly[np.arange(N), idx]), 0.0)
    goal[0, :, 2] = fut_len
    pidx = np.arange(N, dtype=np.int64)[None, :, None]
    dpts = np.stack([lx, ly], -1)[:, ::drag_rate][None].astype(np.float32)    # [1, N, Td, 2], NaN where the log has no state
    dmask = pm & np.isfinite(dpts[0]).all(-1).any(-1)
    dpts[0, ~dmask] = np.nan
    return {"goal": dict(input=goal, mask=gmask[None], prompt_idx=pidx.copy()),
            "drag_point": dict(input=dpts, mask=dmask[None], prompt_idx=pidx.copy())}


def tracks_in_frame(tracks: Dict[str, np.ndarray], frame: Sequence[float]) -> Dict[str, np.ndarray]:
    """The track table with positions, velocities, accelerations and headings expressed in ``frame`` = (x, y, heading) --
    what a scene-centric trajdata batch holds (``standardize_data``: everything relative to the centred agent at t0)."""
    out = dict(tracks)
    out["x"], out["y"] = _rotate(tracks["x"] - frame[0], tracks["y"] - frame[1], -frame[2])
    out["vx"], out["vy"] = _rotate(tracks["vx"], tracks["vy"], -frame[2])
    out["ax"], out["ay"] = _rotate(tracks["ax"], tracks["ay"], -frame[2])
    out["heading"] = (tracks["heading"] - frame[2] + np.pi) % (2 * np.pi) - np.pi
    return out


def ego_frame(tracks: Dict[str, np.ndarray], t0: int, agent_id: str = "ego") -> np.ndarray:
    """(x, y, heading) of the centre agent at ``t0`` -- the frame trajdata centres a scene batch on (the reference reads
    it as centered_agent_from_world_tf, data_utils.py:141-142)."""
    i = list(tracks["agent_ids"]).index(agent_id)
    f = np.array([tracks["x"][i, t0], tracks["y"][i, t0], tracks["heading"][i, t0]], np.float64)
    if not np.isfinite(f).all():
        raise ValueError("the centre agent is absent at t0")
    return f


def polylines_to_map(spec: ModelSpec, polylines: Sequence[np.ndarray], points: int = 19, lane_type: int = 1) -> Dict[str, np.ndarray]:
    """``polylines``: arrays ``[n_i + 1, 2]`` of scene-frame vertices (n_i <= points segments each).
    Returns map_input [1, M, P, 11], map_mask [1, M, P], map_pos [1, M, 2], map_head [1, M]."""
    M, P = len(polylines), points
    f32 = np.float32
    inp = np.zeros((1, M, P, spec.map_dim), np.float64)
    msk = np.zeros((1, M, P), bool)
    pos = np.zeros((1, M, 2), np.float64)
    head = np.zeros((1, M), np.float64)
    for m, pl in enumerate(polylines):
        pl = np.asarray(pl, np.float64)
        n = min(len(pl) - 1, P)
        if n < 1:
            raise ValueError("a polyline needs at least two vertices")
        start, end = pl[:n], pl[1:n + 1]
        s0, e1 = start[0], end[-1]                       # local_map_to_sym_coord: first start, last valid end
        hd = np.arctan2(e1[1] - s0[1], e1[0] - s0[0])
        ctr = (s0 + e1) / 2
        for k, pts in ((0, start), (2, end)):
            lx, ly = _rotate(pts[:, 0] - ctr[0], pts[:, 1] - ctr[1], -hd)
            inp[0, m, :n, k], inp[0, m, :n, k + 1] = lx, ly
        inp[0, m, :n, 4] = lane_type
        inp[0, m, :n, 5] = 0.0                           # traffic-light state: unknown
        for tid in (1, 2, 3):
            inp[0, m, :n, 6 + tid - 1] = float(lane_type == tid)
        d = inp[0, m, :n, 2:4] - inp[0, m, :n, 0:2]
        inp[0, m, :n, 9:11] = d / np.clip(np.linalg.norm(d, axis=-1, keepdims=True), 1e-6, None)
        msk[0, m, :n] = True
        pos[0, m], head[0, m] = ctr, hd
    return dict(map_input=inp.astype(f32), map_mask=msk, map_pos=pos.astype(f32), map_head=head.astype(f32))


def lanes_from_tracks(tracks: Dict[str, np.ndarray], points: int = 19, min_len: float = 5.0, stride: int = 2):
    """Lane-centre polylines along the paths the agents drove (a stand-in for the VectorMap lanes, see the module
    docstring): every agent path is cut into polylines of ``points`` segments of ``stride`` steps each."""
    out = []
    for i in range(tracks["x"].shape[0]):
        ok = np.isfinite(tracks["x"][i]) & np.isfinite(tracks["y"][i])
        p = np.stack([tracks["x"][i][ok], tracks["y"][i][ok]], -1)[::stride]
        for a in range(0, max(len(p) - 1, 0), points):
            seg = p[a:a + points + 1]
            if len(seg) >= 2 and np.linalg.norm(seg[-1] - seg[0]) >= min_len:
                out.append(seg)
    if not out:
        raise ValueError("no agent moved far enough to draw a lane")
    return out
