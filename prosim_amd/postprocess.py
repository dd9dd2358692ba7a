"""Row (f) "next" helpers around the engine: M-replica fan-out and world-frame output.

* ``replicate_scene`` -- the reference batches M replicas of one scene on the batch dim for parallel
  rollouts (``replica_batch_for_parallel_rollout``, prosim/rollout/gpu_utils.py:59-123); with TOP_K = 1
  the replicas are identical, so this is the throughput shape of the Sim-Agents workload (32 replicas).
* ``trajs_to_world`` -- ``obtain_rollout_trajs_in_world`` (gpu_utils.py:230-281) with
  ``batch_nd_transform_points_pt`` / ``batch_nd_transform_angles_pt`` (rollout/utils.py:347-392): rotate the
  agent-init-frame rollout by the initial heading, translate by the initial position.

NOTE on frames: the rollout loop itself adds the agent-frame xy to ``init_pos`` WITHOUT rotating
(``a_pos['position'] = init_pos + traj[..., :2]``, traj_sam.py:213) -- the engine mirrors that bit for bit.
``trajs_to_world`` is the separate output transform of the Waymo packaging path, which does rotate.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def replicate_scene(scene: Dict[str, np.ndarray], m: int) -> Dict[str, np.ndarray]:
    """Tile a 1-scene batch M times on the batch dim (gpu_utils.py:59-123)."""
    if scene["prompt_mask"].shape[0] != 1:
        raise ValueError("replicate_scene expects a single-scene batch")
    def rep(v):
        if isinstance(v, dict):
            return {k: rep(x) for k, x in v.items()}
        a = np.asarray(v)
        if a.ndim and a.shape[0] == 1:
            return np.repeat(a, m, axis=0)
        if a.ndim >= 2 and a.shape[1] == 1 and a.shape[0] != 1:      # fut_obs_input [R-1, B, ...]
            return np.repeat(a, m, axis=1)
        return a
    return {k: rep(v) for k, v in scene.items()}


def trajs_to_world(traj: np.ndarray, init_pos: np.ndarray, init_heading: np.ndarray) -> Dict[str, np.ndarray]:
    """traj [..., S, 4] (x, y, sin, cos in the agent-init frame), init_pos [..., 2], init_heading [...]
    -> world-frame xy [..., S, 2] and heading [..., S] (gpu_utils.py:230-281)."""
    h0 = np.asarray(init_heading, np.float64)[..., None]
    c, s = np.cos(h0), np.sin(h0)
    x, y = traj[..., 0].astype(np.float64), traj[..., 1].astype(np.float64)
    xy = np.stack([x * c - y * s, x * s + y * c], axis=-1) + np.asarray(init_pos, np.float64)[..., None, :]
    heading = np.arctan2(traj[..., 2], traj[..., 3]).astype(np.float64) + h0
    heading = (heading + np.pi) % (2 * np.pi) - np.pi
    return dict(xy=xy.astype(np.float32), heading=heading.astype(np.float32))
