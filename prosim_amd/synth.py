"""Seeded synthetic scenes in the rollout path's input layout.

Layouts follow the reference's batch formatters (dataset/format_utils.py:221-263 map,
:357-447 history, prompt_utils.py:111-150 prompt, condition_utils.py:126-222 conditions):

  map_input  [B, M, P, 11]  per polyline-local segments: x0,y0,x1,y1, type, tl, type one-hot(3), dir(2)
  map_mask   [B, M, P]      bool
  map_pos    [B, M, 2], map_head [B, M]          polyline frame in the scene frame
  obs_input  [B, N, 11, 24] ego-relative history: x,y,sin,cos,vx,vy,ax,ay, extent(2), type one-hot(3), time one-hot(11)
  obs_mask   [B, N, 11, 24] bool
  obs_pos    [B, N, 2], obs_head [B, N]          agent pose at the last history step, scene frame
  prompt     [B, N, 7]  v_local(2), extent(2), type one-hot(3);  prompt_mask [B, N];  agent_type [B, N] in 1..3
  cond       {'goal': {input [B,C,3]=(gx,gy,t), mask [B,C], prompt_idx [B,C,1]},
              'drag_point': {input [B,C,T,2] (NaN = no point), mask, prompt_idx},
              'v_action_tag': {input [B,C,3]=(tag_id,t0,t1), mask, prompt_idx}}
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .spec import ModelSpec


def make_scene(spec: ModelSpec, n_agents: int, n_polylines: int, batch: int = 1, seed: int = 0,
               square: float = 200.0, points: int = 19, goal: bool = False, tags: bool = False, dup_tags: bool = False,
               ragged: bool = False, clustered: bool = False, replay: float = 0.0, drag: bool = False, v2v: bool = False, v2v_reverse: bool = False,
               enter: float = 0.0) -> Dict[str, np.ndarray]:
    """One batch of ``batch`` scenes.  ``ragged``: later scenes in the batch get fewer agents /
    polylines / points, and some history steps are masked (NaN), to exercise the mask paths.
    ``clustered``: agents are placed along polylines (realistic density) instead of uniformly.
    ``replay`` > 0: that fraction of the observed agents is NOT policy-controlled (prompt_mask False, history valid):
    they replay a log -- ``fut_obs_input / fut_obs_mask / fut_obs_pos / fut_obs_head`` [R-1, B, N, ...] carry their
    observation and pose at every later replan (straight-line motion here; a few drop out of the log), as the
    reference's ``batch.extras['fut_obs'][t]`` does (traj_sam.py:221-270).
    ``enter`` > 0 (needs ``replay``): that fraction of the log-replay agents is NOT in the scene at the initial step
    (no valid history point in ``obs_mask``) and enters at a later replan: masked frames before, valid frames from
    then on -- get_center_obs lists an agent only while its state is finite (format_utils.py:383-388)."""
    rng = np.random.RandomState(1234 + seed)
    B, N, M, P, H = batch, n_agents, n_polylines, points, spec.hist_steps
    f32 = np.float32
    half = square / 2

    map_pos = rng.uniform(-half, half, (B, M, 2)).astype(f32)
    map_head = rng.uniform(-np.pi, np.pi, (B, M)).astype(f32)
    map_input = np.zeros((B, M, P, spec.map_dim), f32)
    seg = rng.uniform(-5, 5, (B, M, P, 4)).astype(f32)
    map_input[..., :4] = seg
    map_input[..., 4] = 1.0               # lane centre type
    map_input[..., 5] = 0.0               # traffic light state
    map_input[..., 6] = 1.0               # one-hot(type == 1)
    d = seg[..., 2:4] - seg[..., 0:2]
    map_input[..., 9:11] = d / np.clip(np.linalg.norm(d, axis=-1, keepdims=True), 1e-6, None)
    map_mask = np.ones((B, M, P), bool)

    if clustered:
        pick = rng.randint(0, M, (B, N))
        obs_pos = (np.take_along_axis(map_pos, pick[..., None].repeat(2, -1), 1)
                   + rng.uniform(-3, 3, (B, N, 2))).astype(f32)
    else:
        obs_pos = rng.uniform(-half, half, (B, N, 2)).astype(f32)
    obs_head = rng.uniform(-np.pi, np.pi, (B, N)).astype(f32)
    obs_input = np.zeros((B, N, H, spec.obs_dim), f32)
    obs_input[..., 0:2] = rng.uniform(-2, 2, (B, N, H, 2))
    dth = rng.uniform(-0.1, 0.1, (B, N, H))
    obs_input[..., 2] = np.sin(dth)
    obs_input[..., 3] = np.cos(dth)
    obs_input[..., 4:8] = rng.uniform(-1, 1, (B, N, H, 4))
    # last history step is the agent's own frame origin
    obs_input[:, :, -1, 0:2] = 0.0
    obs_input[:, :, -1, 2] = 0.0
    obs_input[:, :, -1, 3] = 1.0
    agent_type = rng.randint(1, 4, (B, N)).astype(np.int64) if ragged else np.ones((B, N), np.int64)
    extent = np.array([4.5, 2.0], f32)
    obs_input[..., 8:10] = extent
    for tid in (1, 2, 3):
        obs_input[..., 10 + tid - 1] = (agent_type == tid)[..., None]
    obs_input[..., 13:13 + H] = np.eye(H, dtype=f32)
    obs_mask = np.ones((B, N, H, spec.obs_dim), bool)

    prompt_mask = np.ones((B, N), bool)
    if ragged:
        for b in range(B):
            n_b = min(N, max(2, N - 3 * b - (1 if b == 0 else 0)))
            m_b = min(M, max(4, M - 5 * b - 2))
            prompt_mask[b, n_b:] = False
            obs_mask[b, n_b:] = False
            map_mask[b, m_b:] = False
            # ragged polylines: trailing points invalid
            npts = rng.randint(1, P + 1, M)
            for m in range(M):
                map_mask[b, m, npts[m]:] = False
            # early history steps missing for some agents (NaN-padded, as get_center_obs does)
            miss = rng.randint(0, H // 2, N)
            for n in range(n_b):
                obs_mask[b, n, :miss[n], :8] = False
        obs_input = np.where(obs_mask, obs_input, np.nan).astype(f32)
        map_input = np.where(map_mask[..., None], map_input, 0.0).astype(f32)

    prompt = np.zeros((B, N, spec.prompt_dim), f32)
    prompt[..., 0:2] = np.nan_to_num(obs_input[:, :, -1, 4:6])
    prompt[..., 2:4] = extent
    for tid in (1, 2, 3):
        prompt[..., 4 + tid - 1] = agent_type == tid

    scene = dict(map_input=map_input, map_mask=map_mask, map_pos=map_pos, map_head=map_head,
                 obs_input=obs_input, obs_mask=obs_mask, obs_pos=obs_pos, obs_head=obs_head,
                 prompt=prompt, prompt_mask=prompt_mask, agent_type=agent_type)
    if replay > 0:
        observed = prompt_mask.copy()
        logged = observed & (rng.rand(B, N) < replay)
        logged[:, 0] = False                                   # at least one policy agent per scene
        prompt_mask = observed & ~logged
        scene["prompt_mask"] = prompt_mask
        scene["prompt"] = np.where(prompt_mask[..., None], prompt, 0.0).astype(f32)
        R = spec.n_replans
        speed = rng.uniform(0.0, 8.0, (B, N)).astype(f32)
        fi = np.repeat(obs_input[None], R - 1, 0).copy()
        fm = np.repeat(obs_mask[None], R - 1, 0).copy()
        fp = np.zeros((R - 1, B, N, 2), f32)
        fh = np.zeros((R - 1, B, N), f32)
        for r in range(1, R):
            tsec = r * spec.replan_freq * spec.dt
            fp[r - 1] = obs_pos + (speed * tsec)[..., None] * np.stack([np.cos(obs_head), np.sin(obs_head)], -1)
            fh[r - 1] = obs_head + 0.02 * r
            # ego-relative history of a straight constant-speed drive, in the frame of the pose at that replan
            steps = (np.arange(H) - (H - 1)) * spec.dt
            fi[r - 1, ..., 0] = speed[..., None] * steps
            fi[r - 1, ..., 1] = 0.0
            fi[r - 1, ..., 2], fi[r - 1, ..., 3] = 0.0, 1.0
            fi[r - 1, ..., 4], fi[r - 1, ..., 5] = speed[..., None], 0.0
            fi[r - 1, ..., 6:8] = 0.0
            gone = logged & (rng.rand(B, N) < 0.15)            # dropped from the log at this replan: all points masked
            fm[r - 1][gone] = False
            fm[r - 1][~observed] = False
        if enter > 0:
            late = logged & (rng.rand(B, N) < enter)
            first = rng.randint(1, max(2, R - 1), (B, N))              # replan index of the first frame that lists the agent
            for r in range(1, R):
                fm[r - 1][late & (first > r)] = False
            obs_mask = obs_mask.copy()
            obs_mask[late] = False
            scene["obs_mask"] = obs_mask
            scene["obs_input"] = np.where(obs_mask, obs_input, np.nan).astype(f32)
        fi = np.where(fm, fi, np.nan).astype(f32)
        scene.update(fut_obs_input=fi, fut_obs_mask=fm, fut_obs_pos=fp, fut_obs_head=fh)
    cond = {}
    if goal:
        g = np.zeros((B, N, 3), f32)
        g[..., 0:2] = rng.uniform(-50, 50, (B, N, 2))
        g[..., 2] = float(spec.max_steps)
        gm = prompt_mask.copy()
        if ragged:
            gm &= rng.rand(B, N) < 0.7
        cond["goal"] = dict(input=g, mask=gm, prompt_idx=np.tile(np.arange(N)[None, :, None], (B, 1, 1)).astype(np.int64))
    if tags:
        tg = np.zeros((B, N, 3), f32)
        tg[..., 0] = rng.randint(0, 11, (B, N))
        tg[..., 1] = 0.0
        tg[..., 2] = float(spec.max_steps)
        tm = prompt_mask.copy()
        if ragged:
            tm &= rng.rand(B, N) < 0.6
        pidx = np.tile(np.arange(N)[None, :, None], (B, 1, 1)).astype(np.int64)
        if dup_tags:
            # a second tag entry per prompt over the second half of the horizon: the SAME tag for every other prompt (the reference
            # writes its edge matrix by assignment, condition_attns.py:155-166: of two entries of one tag on one prompt the later
            # one survives), another tag for the rest (both count in the mean pool)
            tg[..., 2] = float(spec.max_steps // 2)
            tg2 = tg.copy()
            tg2[..., 1], tg2[..., 2] = float(spec.max_steps // 2), float(spec.max_steps)
            other = (np.arange(N)[None, :] % 2 == 1)
            tg2[..., 0] = np.where(other, (tg[..., 0] + 1 + rng.randint(0, 10, (B, N))) % 11, tg[..., 0])
            tm2 = tm & (rng.rand(B, N) < 0.8)
            tg, tm, pidx = np.concatenate([tg, tg2], 1), np.concatenate([tm, tm2], 1), np.concatenate([pidx, pidx], 1)
        cond["v_action_tag"] = dict(input=tg, mask=tm, prompt_idx=pidx)
    if drag:
        # drag points (condition_utils.py:401-447): every 5th step of a path in the agent's start frame, a
        # consecutive subset of the points kept, the others NaN; a condition without any point is masked off
        Td = (spec.max_steps + 4) // 5
        t_ = np.arange(Td, dtype=f32)[None, None, :]
        v = rng.uniform(2.0, 12.0, (B, N, 1)).astype(f32)
        curv = rng.uniform(-0.02, 0.02, (B, N, 1)).astype(f32)
        dp = np.stack([v * 0.5 * t_, curv * (v * 0.5 * t_) ** 2], -1) + rng.normal(0, 0.1, (B, N, Td, 2)).astype(f32)
        lo = rng.randint(0, Td - 5, (B, N, 1))
        hi = lo + rng.randint(5, Td, (B, N, 1))
        keep = (np.arange(Td)[None, None, :] >= lo) & (np.arange(Td)[None, None, :] < hi)
        dm = prompt_mask.copy()
        if ragged:
            dm &= rng.rand(B, N) < 0.5
        keep &= dm[..., None]
        dp = np.where(keep[..., None], dp, np.nan).astype(f32)
        cond["drag_point"] = dict(input=dp, mask=dm, prompt_idx=np.tile(np.arange(N)[None, :, None], (B, 1, 1)).astype(np.int64))
    if v2v:
        # binary tags (condition_utils: 'v2v_tag'): per scene up to N pairs of distinct policy agents, a tag of the spec's
        # used V2V tags, a time span; every (tag, s, t) at most once (the reference writes its edge matrix by assignment)
        from .spec import V2V_TAGS
        if not spec.used_v2v_tags:
            raise ValueError("v2v conditions need spec.used_v2v_tags")
        C = N
        vi = np.zeros((B, C, 3), f32)
        vi[..., 0] = -1.0
        vm = np.zeros((B, C), bool)
        vp = np.zeros((B, C, 2), np.int64)
        for b in range(B):
            pol = np.nonzero(prompt_mask[b])[0]
            seen = set()
            if len(pol) < 2:
                continue
            for c in range(C):
                s_, t_ = rng.choice(pol, 2, replace=False)
                tag = V2V_TAGS.index(spec.used_v2v_tags[rng.randint(len(spec.used_v2v_tags))])
                if (tag, int(s_), int(t_)) in seen or (tag, int(t_), int(s_)) in seen:
                    continue
                seen.add((tag, int(s_), int(t_)))
                t0 = float(rng.randint(0, spec.max_steps // 2))
                vi[b, c] = (tag, t0, t0 + float(rng.randint(5, spec.max_steps // 2)))
                vp[b, c] = (s_, t_)
                vm[b, c] = (rng.rand() < 0.7) if ragged else True
        if v2v_reverse:
            # the corner case of the reference's two assignment passes (condition_attns.py:155-162): the SAME tag on a pair in both
            # directions, (s, t) and (t, s) -- the later pass's target halves overwrite the source halves on both edges
            for b in range(B):
                free = [c for c in range(C) if not vm[b, c] and vi[b, c, 0] < 0]
                have = [c for c in range(C) if vm[b, c]]
                for c_src, c_new in zip(have[:3], free):
                    vi[b, c_new] = vi[b, c_src]
                    vi[b, c_new, 1] += 3.0          # (another time span: the two entries' embeddings differ)
                    vp[b, c_new] = vp[b, c_src][::-1]
                    vm[b, c_new] = True
        cond["v2v_tag"] = dict(input=vi, mask=vm, prompt_idx=vp)
    if cond:
        scene["cond"] = cond
    return scene


# The BASELINE.json configs (index -> generator kwargs).  Config 0's real demo_dataset scene is built by
# prosim_amd/formatting.py from the sample agent table and prosim_amd/vecmap.py from the cache's VectorMap protobuf
# (tests/test_vecmap_cpu.py::demo_scene_real_lanes); this entry is its synthetic twin of the same shape; config 4 (Waymo-val dense scene, unobtainable
# offline) is a 256-agent scene in a 100 m square.  Both substitutions are stated in DESIGN.md.
BASELINE_CONFIGS = [
    dict(name="cfg0_16a_128p", n_agents=16, n_polylines=128, batch=1),
    dict(name="cfg1_64a_512p", n_agents=64, n_polylines=512, batch=1),
    dict(name="cfg2_128a_1024p_goal", n_agents=128, n_polylines=1024, batch=1, goal=True),
    dict(name="cfg3_8x128a_1024p_goal", n_agents=128, n_polylines=1024, batch=8, goal=True),
    dict(name="cfg4_256a_1024p_goal_tags_dense", n_agents=256, n_polylines=1024, batch=1, goal=True, tags=True,
         square=100.0),
]


def baseline_scene(spec: ModelSpec, idx: int, seed: int = 0, batch: Optional[int] = None) -> Dict[str, np.ndarray]:
    kw = dict(BASELINE_CONFIGS[idx])
    kw.pop("name")
    if batch is not None:
        kw["batch"] = batch
    return make_scene(spec, seed=seed, **kw)


def make_pair_metric_inputs(seed: int, B: int = 2, N: int = 6, R: int = 8, K: int = 3, S: int = 10, D: int = 5):
    """Seeded inputs of the rollout metric: per-(scene, replan, agent) local targets with the gaps a real log has
    (agents that leave: trailing NaN steps; agents without any future at a replan: mask False or all-NaN; one
    coordinate missing), K-mode predictions and mode probabilities."""
    rng = np.random.RandomState(4321 + seed)
    tgt = np.cumsum(rng.uniform(-0.5, 1.5, (B, R, N, S, D)), axis=3).astype(np.float32)
    tgt[..., 2] = rng.uniform(-0.3, 0.3, (B, R, N, S))                # per-step heading offsets stay small
    mask = np.ones((B, R, N), bool)
    for b in range(B):
        tgt[b, 3:, 1] = np.nan                                         # agent 1 leaves after replan 2 ...
        mask[b, 3:, 1] = False
        tgt[b, 2, 1, 4:] = np.nan                                      # ... and its last logged replan is cut short
        tgt[b, 5, 2] = np.nan                                          # a NaN target whose mask says valid
        mask[b, 6, 3] = False                                          # a masked pair whose target is finite
        tgt[b, 1, 4, 7:, 0] = np.nan                                   # only x missing on the last steps
    n_agents = [N, N - 2][:B] + [N] * max(0, B - 2)
    pairs = [(b, t, n) for b in range(B) for t in range(R) for n in range(n_agents[b]) if mask[b, t, n]]
    bidx, tidx, nidx = (np.array(v, np.int64) for v in zip(*pairs))
    pair_tgt = tgt[bidx, tidx, nidx]
    pred = (np.nan_to_num(pair_tgt)[:, None] + rng.normal(0, 0.4, (len(pairs), K, S, D))).astype(np.float32)
    prob = rng.uniform(0, 1, (len(pairs), K)).astype(np.float32)
    return dict(tgt=tgt, mask=mask, motion_pred=pred, motion_prob=prob, bidx=bidx, tidx=tidx, nidx=nidx)
