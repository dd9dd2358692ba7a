"""ORACLE (test infrastructure) -- CPU restatement of the reference's world-frame output step.  Imported by tests/,
__graft_entry__.smoke() and tests/gen_golden.py only; the product path is the device kernel k_world_traj.

``trajs_in_world`` follows ``obtain_rollout_trajs_in_world`` (prosim/rollout/gpu_utils.py:255-267):

* ``batch_rotate_2D(trajs[:, :, :2], init_heads) + init_pos[:, None]``      (models/utils/geometry.py:19-22)
* ``wrap_angle(arctan2(trajs[..., 2], trajs[..., 3]) + init_heads)``          (models/utils/geometry.py:13-17)
* ``batch_nd_transform_points_pt(xy, tf)`` / ``batch_nd_transform_angles_pt(h, tf)`` with ``angle_wrap``
                                                                            (rollout/utils.py:272-283, :347-392)

Pinned by tests/golden/ref_world_trajs.npz, made by tests/gen_golden.py from the reference's own function
(oracle/ref_harness.load_world_output).  ``replicate_rows`` is the row order of the M-replica batch
(replica_batch_for_parallel_rollout, gpu_utils.py:59-123: every per-agent tensor ``.repeat(M, ...)`` -> replica-major).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def trajs_in_world(traj, init_pos, init_head, center_to_world=None, dtype=torch.float32) -> torch.Tensor:
    """traj [n, T, 4] (x, y, sin, cos in the agent-init frame), init_pos [n, 2], init_head [n] or [n, 1],
    center_to_world [3, 3] or None -> [n, T, 3] (x, y, heading) in the world frame."""
    traj = torch.as_tensor(np.asarray(traj), dtype=dtype)
    pos = torch.as_tensor(np.asarray(init_pos), dtype=dtype)
    th = torch.as_tensor(np.asarray(init_head), dtype=dtype).reshape(-1, 1)
    tf = torch.eye(3, dtype=dtype) if center_to_world is None else torch.as_tensor(np.asarray(center_to_world), dtype=dtype)
    x, y = traj[..., 0], traj[..., 1]
    xc = torch.stack([x * torch.cos(th) - y * torch.sin(th), y * torch.cos(th) + x * torch.sin(th)], dim=-1) + pos[:, None]
    hs = torch.arctan2(traj[..., 2], traj[..., 3])
    hc = -math.pi + (hs + th + math.pi) % (2 * math.pi)
    mt = tf.transpose(-1, -2)
    xw = (xc[..., None, :] @ mt[None, :2, :2]).squeeze(-2) + mt[-1:, :2]
    rot = torch.arctan2(tf[1, 0], tf[0, 0])
    hw = (hc + rot + np.pi) % (2 * np.pi) - np.pi
    return torch.cat([xw, hw[..., None]], dim=-1)


def replicate_rows(a: np.ndarray, m: int) -> np.ndarray:
    """Per-agent rows [n, ...] of one scene -> the M-replica batch's rows [m * n, ...] (replica-major)."""
    return np.concatenate([np.asarray(a)] * m, axis=0)
