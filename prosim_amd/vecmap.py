"""The demo cache's vector maps (``trajdata_cache/<env>/maps/<map>.pb``) -> the rollout path's lane vectors, without
trajdata and without protobuf.

trajdata (a third-party fork, absent from the reference tree and from this image) stores each map as one protobuf
message and hands the reference ``VectorMap`` lane objects; the reference then samples, clips and chunks the lane
polylines into ``batch.extras['vector_lane']`` (prosim/dataset/data_utils.py:156-255) and the formatter keeps the
closest ``MAX_POINTS`` of them in their own midpoint/tangent frames (prosim/dataset/format_utils.py:150-263).

* ``decode_vector_map``   -- a hand-written protobuf WIRE decoder.  No ``.proto`` ships with the reference; the message
                             layout below was read off the wire format of the demo maps and is cross-checked against
                             redundancy the files themselves carry (tests/test_vecmap_cpu.py): the stored per-vertex
                             headings equal atan2 of the decoded deltas, the decoded extent equals the stored
                             min/max points, left/right boundaries lie on the left/right of the centre line::

                                 VectorizedMap { 1: name  2: repeated MapElement  3: max_pt  4: min_pt  5: shifted_origin }
                                 MapElement    { 1: id (bytes)  2: RoadLane | 3: RoadArea | 4: PedCrosswalk | 5: PedWalkway }
                                 RoadLane      { 1: center  2: left_boundary  3: right_boundary  (Polyline)
                                                 4: entry  5: exit  6: adjacent_left  7: adjacent_right  (repeated bytes) }
                                 Polyline      { 1: dx_mm  2: dy_mm  3: dz_mm  (packed sint32: the first value is the vertex
                                                 relative to shifted_origin in mm, the rest are vertex-to-vertex deltas)
                                                 4: h_rad (packed double) }
                                 Point         { 1: x  2: y  3: z  (double) }

* ``vector_lanes``        -- ``_get_vectorized_lanes_from_vector_map`` (data_utils.py:156-255) with COLLATE_MODE 'lane':
                             lanes near the centre agent, centre line every CENTER_SAMPLE_RATE-th vertex and the edges
                             every EDGE_SAMPLE_RATE-th, in the centre agent's frame, clipped to the MAP.RANGE square,
                             cut into chunks of MAX_LANE_POINTS vertices -> ``[M, MAX_LANE_POINTS - 1, 6]`` segment rows
                             (x0, y0, x1, y1, line type, traffic-light status; type 0 = padding).
* ``local_vector_map``    -- ``get_local_vec_map`` (format_utils.py:150-182): the chunks whose mean start point lies
                             within LOCAL_RANGE of the centre agent, at most MAX_POINTS of them (the closest).
* ``vectors_to_map``      -- ``local_map_to_sym_coord`` + ``get_center_vec_init_map`` (format_utils.py:184-263): every
                             chunk in its own frame, type one-hot, unit direction.

What stays unpinned is trajdata's side of the boundary (which lanes ``get_lanes_within`` returns and in which order,
the traffic-light enum); it is restated from the call sites and marked where used.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .spec import ModelSpec

# RoadLaneType (data_utils.py:23-26)
LINE_TYPES = {"center": 1.0, "left_edge": 2.0, "right_edge": 3.0}
# trajdata's TrafficLightStatus for a lane without a record at this step (unpinned: restated from the enum's name)
TLS_NO_DATA = -1.0


class WireError(ValueError):
    """The bytes are not a well-formed protobuf message of the layout above."""


def _varint(b: bytes, i: int) -> Tuple[int, int]:
    r = s = 0
    while True:
        if i >= len(b) or s > 63:
            raise WireError("truncated or over-long varint")
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _fields(b: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message: varints as ints, fixed/length-delimited as bytes."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v, i = b[i:i + 8], i + 8
        elif w == 5:
            v, i = b[i:i + 4], i + 4
        elif w == 2:
            ln, i = _varint(b, i)
            v, i = b[i:i + ln], i + ln
        else:
            raise WireError(f"wire type {w} (groups) is not used by this format")
        if i > n:
            raise WireError("field runs past the end of its message")
        yield f, w, v


def _packed_sint32(b: bytes) -> np.ndarray:
    """Packed zigzag varints -> int64.  Vectorised: a varint ends at every byte without the continuation bit."""
    a = np.frombuffer(b, np.uint8)
    if a.size == 0:
        return np.zeros(0, np.int64)
    ends = np.nonzero(a < 0x80)[0]
    if ends.size == 0 or ends[-1] != a.size - 1:
        raise WireError("packed varints end mid-value")
    starts = np.concatenate([[0], ends[:-1] + 1])
    if (ends - starts).max() > 9:
        raise WireError("over-long varint")
    pos = np.arange(a.size) - np.repeat(starts, ends - starts + 1)            # byte index inside its varint
    vals = np.add.reduceat((a & 0x7F).astype(np.uint64) << (7 * pos).astype(np.uint64), starts)
    vals = vals & np.uint64(0xFFFFFFFF)                                       # sint32: negative values arrive sign-extended
    return (vals >> np.uint64(1)).astype(np.int64) ^ -(vals & np.uint64(1)).astype(np.int64)


def _point(b: bytes) -> np.ndarray:
    p = np.zeros(3, np.float64)
    for f, w, v in _fields(b):
        if w == 1 and 1 <= f <= 3:
            p[f - 1] = struct.unpack("<d", v)[0]
    return p


def _polyline(b: bytes, origin: np.ndarray) -> np.ndarray:
    """-> [n, 4] float64 (x, y, z, heading); heading NaN when the polyline stores none."""
    d = {1: None, 2: None, 3: None}
    h = None
    for f, w, v in _fields(b):
        if w != 2:
            raise WireError("polyline fields are packed")
        if f in d:
            d[f] = _packed_sint32(v)
        elif f == 4:
            if len(v) % 8:
                raise WireError("h_rad is not a whole number of doubles")
            h = np.frombuffer(v, "<f8")
    n = 0 if d[1] is None else len(d[1])
    if d[2] is None or len(d[2]) != n:
        raise WireError("dx_mm and dy_mm differ in length")
    out = np.full((n, 4), np.nan, np.float64)
    out[:, 0] = np.cumsum(d[1]) / 1000.0 + origin[0]
    out[:, 1] = np.cumsum(d[2]) / 1000.0 + origin[1]
    out[:, 2] = (np.cumsum(d[3]) / 1000.0 if d[3] is not None and len(d[3]) == n else np.zeros(n)) + origin[2]
    if h is not None:
        if len(h) != n:
            raise WireError("h_rad and dx_mm differ in length")
        out[:, 3] = h
    return out


def decode_vector_map(data: bytes) -> Dict[str, object]:
    """One ``maps/<name>.pb`` -> ``{'name', 'max_pt' [3], 'min_pt' [3], 'origin' [3], 'lanes': [ {id, center [n,4],
    left [m,4] | None, right [k,4] | None, entry, exit, adj_left, adj_right (lists of ids)} ], 'others': {kind: [polygon
    [n,4]]}}``.
    World-frame metres.  Only road lanes are expanded (the path's INCLUDE_TYPES are lane lines, data_utils.py:47-52:
    areas, crosswalks and walkways are switched off)."""
    top = list(_fields(data))
    pts = {f: _point(v) for f, w, v in top if f in (3, 4, 5) and w == 2}
    if 5 not in pts:
        raise WireError("the map has no shifted_origin")
    origin = pts[5]
    name = b"".join(v for f, w, v in top if f == 1 and w == 2).decode("utf-8", "replace")
    lanes: List[Dict[str, object]] = []
    others: Dict[str, List[np.ndarray]] = {}
    kinds = {3: "road_area", 4: "ped_crosswalk", 5: "ped_walkway"}
    for f, w, v in top:
        if f != 2 or w != 2:
            continue
        eid, body, kind = "", None, None
        for ef, ew, ev in _fields(v):
            if ef == 1 and ew == 2:
                eid = ev.decode("utf-8", "replace")
            elif ef in (2, 3, 4, 5) and ew == 2:
                kind, body = ef, ev
        if kind != 2:
            if kind is not None:                                              # kept for the extent cross-check only
                others.setdefault(kinds[kind], []).extend(_polyline(pv, origin) for pf, pw, pv in _fields(body) if pw == 2 and pf == 1)
            continue
        lane = {"id": eid, "center": None, "left": None, "right": None, "entry": [], "exit": [], "adj_left": [], "adj_right": []}
        for lf, lw, lv in _fields(body):
            if lw != 2:
                continue
            if lf in (1, 2, 3):
                lane[("center", "left", "right")[lf - 1]] = _polyline(lv, origin)
            elif lf in (4, 5, 6, 7):
                lane[("entry", "exit", "adj_left", "adj_right")[lf - 4]].append(lv.decode("utf-8", "replace"))
        if lane["center"] is None:
            raise WireError(f"lane {eid!r} has no centre line")
        lanes.append(lane)
    return {"name": name, "max_pt": pts.get(3), "min_pt": pts.get(4), "origin": origin, "lanes": lanes, "others": others}


def _to_frame(xy: np.ndarray, frame: Sequence[float]) -> np.ndarray:
    """World xy -> the frame (x, y, heading): offset, then rotate by -heading (transform_coords_np with the centred
    agent's agent_from_world_tf, data_utils.py:213)."""
    c, s = np.cos(-frame[2]), np.sin(-frame[2])
    d = xy - np.asarray(frame[:2], np.float64)
    return np.stack([d[:, 0] * c - d[:, 1] * s, d[:, 0] * s + d[:, 1] * c], -1)


def vector_lanes(lanes: Sequence[Dict[str, object]], frame: Sequence[float], center_z: Optional[float] = None,
                 tls: Optional[Dict[str, float]] = None, map_range: float = 200.0, center_sample_rate: int = 1,
                 edge_sample_rate: int = 4, max_lane_points: int = 20,
                 include: Sequence[str] = ("center", "right_edge", "left_edge")) -> np.ndarray:
    """``frame``: (x, y, heading) of the centre agent at the current step (world).  ``tls``: lane id -> traffic-light
    status at the current step.  Defaults = the demo config (no_text.yaml:134-139; MAP.RANGE.ROLLOUT default.py:233).
    Returns float32 ``[M, max_lane_points - 1, 6]``; with no lane in range one all-padding chunk, as the reference
    (data_utils.py:252-253 -- it hard-codes 39 rows there, the rows this returns follow max_lane_points)."""
    lane_dist = np.sqrt(2.0) * map_range                                      # data_utils.py:172
    q = np.array([frame[0], frame[1], 0.0 if center_z is None else center_z])
    nd = 2 if center_z is None else 3
    out = []
    for lane in lanes:                                                        # file order (unpinned: get_lanes_within's order)
        ctr = lane["center"]
        if not (np.linalg.norm(ctr[:, :nd] - q[:nd], axis=-1) <= lane_dist).any():
            continue
        t = float(TLS_NO_DATA if tls is None else tls.get(lane["id"], TLS_NO_DATA))
        pts = {"center": ctr[:, :2], "left_edge": None if lane["left"] is None else lane["left"][:, :2],
               "right_edge": None if lane["right"] is None else lane["right"][:, :2]}
        for k, v in pts.items():                                              # dict order of the reference: centre, left, right
            if k not in include or v is None:
                continue
            rate = edge_sample_rate if "edge" in k else center_sample_rate
            if v.shape[0] > rate:
                v = v[::rate]
            v = _to_frame(v, frame)
            v = v[(np.abs(v[:, 0]) < map_range) & (np.abs(v[:, 1]) < map_range)]
            n = v.shape[0]
            if n < 2:
                continue
            if n > max_lane_points:
                cuts = list(range(0, n, max_lane_points))
                if cuts[-1] != n:
                    cuts.append(n)
            else:
                cuts = [0, n - 1]                                             # as written there: a short line loses its last vertex
            for a, b in zip(cuts[:-1], cuts[1:]):
                ch = v[a:b]
                m = len(ch) - 1
                if m < 1:
                    continue
                row = np.zeros((max_lane_points - 1, 6))
                row[:m, 0:2], row[:m, 2:4] = ch[:-1], ch[1:]
                row[:m, 4], row[:m, 5] = LINE_TYPES[k], t
                out.append(row)
    if not out:
        return np.zeros((1, max_lane_points - 1, 6), np.float32)
    return np.stack(out).astype(np.float32)


def local_vector_map(full_vec: np.ndarray, local_pos: Sequence[float] = (0.0, 0.0), local_range: float = 200.0,
                     max_points: int = 2048) -> Tuple[np.ndarray, np.ndarray]:
    """``get_local_vec_map`` (format_utils.py:150-182) for [M, P, 6] chunks -> (local_vec [max_points, P, 6],
    local_mask [max_points, P]).  As there, the mask is taken from the first max_points chunks in range BEFORE they are
    re-ordered by distance (it only matters when more than max_points chunks are in range)."""
    full_vec = np.asarray(full_vec, np.float32)
    valid = full_vec[..., 4] > 0
    cnt = np.maximum(valid.sum(1), 1)
    position = full_vec[..., :2].sum(1) / cnt[:, None].astype(np.float32)
    dist = np.linalg.norm(position - np.asarray(local_pos, np.float32), axis=-1)
    near = dist < local_range
    vec, d = full_vec[near], dist[near]
    P = full_vec.shape[1]
    mask = np.zeros((max_points, P), bool)
    p = min(max_points, len(vec))
    mask[:p] = vec[:p, :, 4] > 0
    if len(vec) > max_points:
        vec = vec[np.argsort(d, kind="stable")[:max_points]]
    else:
        vec = np.concatenate([vec, np.zeros((max_points - len(vec), P, full_vec.shape[2]), np.float32)])
    return vec, mask


def vectors_to_map(spec: ModelSpec, local_vec: np.ndarray, local_mask: np.ndarray, drop_padding: bool = True) -> Dict[str, np.ndarray]:
    """``local_map_to_sym_coord`` + the type / direction features of ``get_center_vec_init_map`` (format_utils.py:184-263)
    -> ``map_input [1, M, P, 11], map_mask [1, M, P], map_pos [1, M, 2], map_head [1, M]``.  Chunks are located by their
    OWN validity (type > 0), the returned mask is ``local_mask``.  ``drop_padding`` removes the all-masked rows behind
    the last used chunk (the engine takes ragged token counts; the reference pads to MAX_POINTS and masks)."""
    v = np.array(local_vec, np.float32)
    M, P = v.shape[:2]
    cnt = (v[..., 4] > 0).sum(1)
    start = v[:, 0, :2].copy()
    end = v[np.arange(M), cnt - 1, 2:4].copy()                                # cnt 0 -> the last row, as indexing with -1 does there
    head = np.arctan2(end[:, 1] - start[:, 1], end[:, 0] - start[:, 0]).astype(np.float32)
    pos = ((start + end) / 2).astype(np.float32)
    c, s = np.cos(-head)[:, None], np.sin(-head)[:, None]
    for k in (0, 2):
        x, y = v[..., k] - pos[:, None, 0], v[..., k + 1] - pos[:, None, 1]
        v[..., k], v[..., k + 1] = c * x - s * y, c * y + s * x
    onehot = np.stack([(v[..., 4] == t) for t in (1, 2, 3)], -1).astype(np.float32)
    diff = v[..., 2:4] - v[..., 0:2]
    direc = diff / np.clip(np.linalg.norm(diff, axis=-1, keepdims=True), 1e-6, None)
    inp = np.concatenate([v, onehot, direc], -1)
    if inp.shape[-1] != spec.map_dim:
        raise ValueError("map feature width does not match the spec")
    mask = np.asarray(local_mask, bool)
    if drop_padding:
        used = np.nonzero(mask.any(1) | (cnt > 0))[0]
        keep = slice(0, int(used[-1]) + 1 if len(used) else 1)
        inp, mask, pos, head = inp[keep], mask[keep], pos[keep], head[keep]
    inp = np.where(mask[..., None], inp, 0.0).astype(np.float32)              # masked entries are never read; keep them clean
    return dict(map_input=inp[None], map_mask=mask[None], map_pos=pos[None], map_head=head[None])


def tls_at(lane_ids: Sequence[str], scene_ts: Sequence[int], status: Sequence[int], t: int) -> Dict[str, float]:
    """The cache's ``tls_data_dt*.feather`` columns -> lane id -> status at step ``t``."""
    lane_ids, scene_ts, status = np.asarray(lane_ids).astype(str), np.asarray(scene_ts), np.asarray(status)
    sel = scene_ts == t
    return {a: float(b) for a, b in zip(lane_ids[sel].tolist(), status[sel].tolist())}


def map_for_scene(spec: ModelSpec, pb: bytes, frame: Sequence[float], center_z: Optional[float] = None,
                  tls: Optional[Dict[str, float]] = None, max_points: int = 2048, **lane_kw) -> Dict[str, np.ndarray]:
    """decode -> vector_lanes -> local_vector_map -> vectors_to_map for one scene, in the frame of its centre agent."""
    vm = decode_vector_map(pb)
    full = vector_lanes(vm["lanes"], frame, center_z=center_z, tls=tls, **lane_kw)
    vec, mask = local_vector_map(full, max_points=max_points)
    return vectors_to_map(spec, vec, mask)
