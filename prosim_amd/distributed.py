"""Scene sharding + the one collective of the rollout path.

Scenes are independent (every graph op of the model is batch-segmented, e.g. act_decoder.py:250,
attn_fusion.py:107), so the path shards by scene with NO data-path collective: scene i runs on rank
``i % world`` (mirrors rollout/callbacks.py:76,247).  The only exchange is the metric reduction
after a rollout: the per-agent (ADE, FDE) vector each rank computed on its device
(``ps_rollout_metric``) is all-gathered -- RCCL over xGMI on the GPU box (backend "nccl"), gloo in the
CPU tests.  The payload is a few KB, latency-bound; link bandwidth is irrelevant.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """Scene indices owned by ``rank`` (i % world == rank, rollout/callbacks.py:76)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return [i for i in range(n_scenes) if i % world == rank]


def gather_scene_metrics(local: torch.Tensor, scene_ids: Sequence[int], n_scenes: int, max_agents: int) -> torch.Tensor:
    """All-gather per-scene metric rows.  ``local`` [n_local, max_agents, M] (NaN-padded rows for missing
    agents) for the scenes in ``scene_ids``; returns [n_scenes, max_agents, M] on every rank, in scene order.
    Uneven shards are padded to the largest shard so one all_gather_into_tensor suffices."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    M = local.shape[-1]
    per = (n_scenes + world - 1) // world
    buf = torch.full((per, max_agents, M), float("nan"), dtype=local.dtype, device=local.device)
    ids = torch.full((per,), -1, dtype=torch.int64, device=local.device)
    n = len(scene_ids)
    if n:
        buf[:n] = local
        ids[:n] = torch.as_tensor(list(scene_ids), dtype=torch.int64, device=local.device)
    if world == 1:
        all_buf, all_ids = buf, ids
    else:
        all_buf = torch.empty((world * per, max_agents, M), dtype=local.dtype, device=local.device)
        all_ids = torch.empty((world * per,), dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(all_buf, buf)
        dist.all_gather_into_tensor(all_ids, ids)
    out = torch.full((n_scenes, max_agents, M), float("nan"), dtype=local.dtype, device=local.device)
    keep = all_ids >= 0
    out[all_ids[keep]] = all_buf[keep]
    return out


def reduce_metrics(gathered: torch.Tensor) -> Dict[str, float]:
    """Scene-averaged rollout ADE / FDE over valid agents (metrics/motion_pred.py:125-143: mean over
    agents, then over scenes)."""
    valid = ~torch.isnan(gathered[..., 0])
    cnt = valid.sum(dim=1).clamp(min=1)
    per_scene = torch.nan_to_num(gathered, nan=0.0).sum(dim=1) / cnt[:, None]
    has = valid.any(dim=1)
    m = per_scene[has].mean(dim=0)
    return {"rollout_ade": float(m[0]), "rollout_fde": float(m[1]), "scenes": int(has.sum())}
