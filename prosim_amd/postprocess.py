"""The reference-shaped M-replica batch, for callers that want the reference's layout and for the parity tests.

``replicate_scene`` tiles a ONE-scene batch M times on the batch dim -- what ``replica_batch_for_parallel_rollout``
(prosim/rollout/gpu_utils.py:59-123) does to every tensor before ``rollout_batch``.  The engine does not need it:
``Engine.set_replicas(M)`` (``ps_set_replicas``) rolls M replicas out from the one-scene batch, computes the shared
prefix once and keeps the map tokens once; ``tests/test_replicas_gpu.py`` checks the two routes
against each other.  The world-frame output step (``obtain_rollout_trajs_in_world``, :230-281) is the device kernel
behind ``Engine.world_trajs`` (``ps_world_trajs``).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

# entries whose batch dim is axis 1 ([R - 1, B, N, ...]: the per-replan frames of batch.extras['fut_obs'])
_FRAME_KEYS = ("fut_obs_input", "fut_obs_mask", "fut_obs_pos", "fut_obs_head")
# entries without a batch dim
_PLAIN_KEYS = ("mode_choice",)


def replicate_scene(scene: Dict[str, np.ndarray], m: int) -> Dict[str, np.ndarray]:
    """Tile a 1-scene batch M times on the batch dim (gpu_utils.py:59-123).  ``mode_choice`` is per replica and is not
    tiled: hand the [R, M, N] table over separately."""
    if scene["prompt_mask"].shape[0] != 1:
        raise ValueError("replicate_scene expects a single-scene batch")

    def rep(key, v):
        if isinstance(v, dict):
            return {k: rep(k, x) for k, x in v.items()}
        if v is None or key in _PLAIN_KEYS:
            return v
        a = np.asarray(v)
        if key in _FRAME_KEYS:
            if a.shape[1] != 1:
                raise ValueError(f"{key}: expected [R - 1, 1, N, ...]")
            return np.repeat(a, m, axis=1)
        if a.ndim == 0:
            return a
        if a.shape[0] != 1:
            raise ValueError(f"{key}: expected a batch dim of 1, got {a.shape}")
        return np.repeat(a, m, axis=0)

    return {k: rep(k, v) for k, v in scene.items() if k != "mode_choice"}
