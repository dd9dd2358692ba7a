"""Scene sharding + the one collective of the rollout path.

Scenes are independent (every graph op of the model is batch-segmented, e.g. act_decoder.py:250,
attn_fusion.py:107), so the path shards by scene with NO data-path collective: scene i runs on rank
``i % world`` (mirrors rollout/callbacks.py:76,247).  The only exchange is the metric reduction
after a rollout: the per-agent metric rows each rank computed on its device (``ps_pair_metric``: the sums behind the
reference's PairMotionPred; ``ps_rollout_metric``: closed-loop displacement) are all-gathered -- RCCL over xGMI on the GPU box (backend "nccl"), gloo in the
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


class SceneMetricGather:
    """The all-gather of per-scene metric rows, with every buffer and index tensor built ONCE: a call enqueues two
    collectives and an index_copy on the current stream and never reads device data on the host, so the caller can
    keep launching rollouts while the gather of the previous one is in flight.  Uneven shards are padded to the
    largest shard (padding rows carry scene id -1 and land in a scratch row)."""

    def __init__(self, scene_ids: Sequence[int], n_scenes: int, max_agents: int, n_metrics: int, device,
                 dtype=torch.float32):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.n_scenes, self.n_local = n_scenes, len(scene_ids)
        per = (n_scenes + self.world - 1) // self.world
        if self.n_local > per:
            raise ValueError("shard larger than ceil(n_scenes / world)")
        self.buf = torch.full((per, max_agents, n_metrics), float("nan"), dtype=dtype, device=device)
        ids = torch.full((per,), -1, dtype=torch.int64)
        ids[:self.n_local] = torch.as_tensor(list(scene_ids), dtype=torch.int64)
        if self.world == 1:
            all_ids = ids
        else:
            ids = ids.to(device)
            all_ids = torch.empty((self.world * per,), dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(all_ids, ids)
            all_ids = all_ids.cpu()
        if ((all_ids < -1) | (all_ids >= n_scenes)).any():
            raise ValueError("scene id outside [0, n_scenes)")
        # scene ids never change between calls: gather them once, keep the scatter index on the device
        self.index = torch.where(all_ids >= 0, all_ids, torch.full_like(all_ids, n_scenes)).to(device)
        self.all_buf = torch.empty((self.world * per, max_agents, n_metrics), dtype=dtype, device=device)
        # (round 6: the fills and copies above ran on torch's default stream; the gather is later enqueued on an ENGINE's stream, which is
        # non-blocking -- nothing orders the two.  Built once, so drain once.)
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize()

    def __call__(self, local: torch.Tensor) -> torch.Tensor:
        """``local`` [n_local, max_agents, M] -> [n_scenes, max_agents, M] on every rank, in scene order."""
        if self.n_local:
            self.buf[:self.n_local].copy_(local, non_blocking=True)
        if self.world == 1:
            self.all_buf.copy_(self.buf)
        else:
            dist.all_gather_into_tensor(self.all_buf, self.buf)
        out = torch.full((self.n_scenes + 1,) + tuple(self.all_buf.shape[1:]), float("nan"), dtype=self.all_buf.dtype,
                         device=self.all_buf.device)
        out.index_copy_(0, self.index, self.all_buf)
        return out[:self.n_scenes]


def gather_scene_metrics(local: torch.Tensor, scene_ids: Sequence[int], n_scenes: int, max_agents: int) -> torch.Tensor:
    """One-shot form of :class:`SceneMetricGather`.  ``local`` [n_local, max_agents, M] (NaN-padded rows for missing
    agents) for the scenes in ``scene_ids``; returns [n_scenes, max_agents, M] on every rank, in scene order."""
    return SceneMetricGather(scene_ids, n_scenes, max_agents, local.shape[-1], local.device, local.dtype)(local)


def rows_to_slots(rows: torch.Tensor, slots: torch.Tensor, n_scenes: int, max_agents: int) -> torch.Tensor:
    """Per-agent-row results [A, M] (the engine's compact row order) -> the padded [n_scenes, max_agents, M] slot layout
    the gather works on; slots without an agent row stay NaN.  ``slots`` = Engine.row_slots as a device int64 tensor."""
    out = torch.full((n_scenes * max_agents, rows.shape[-1]), float("nan"), dtype=rows.dtype, device=rows.device)
    out.index_copy_(0, slots, rows)
    return out.view(n_scenes, max_agents, rows.shape[-1])


def reduce_pair_metrics(gathered: torch.Tensor, batches: Sequence[Sequence[int]] = None) -> Dict[str, float]:
    """The scalars PairMotionPred logs (metrics/motion_pred.py:20-25, :184-199) from gathered ``ps_pair_metric`` rows
    [n_scenes, max_agents, 10]: ade / fde / min_ade / min_fde = sum of the finite pair values / their count (what
    MeanMetric keeps); rollout_ade = per update (``batches``: the scene ids of each update, default one update with every
    scene) the mean over the agents with a valid rollout step, then the mean over the updates."""
    g = torch.nan_to_num(gathered.double(), nan=0.0)
    tot = g.sum(dim=(0, 1))
    out = {k: float(tot[i] / tot[4 + i]) if float(tot[4 + i]) > 0 else float("nan")
           for i, k in enumerate(("ade", "fde", "min_ade", "min_fde"))}
    batches = [list(range(gathered.shape[0]))] if batches is None else batches
    per = []
    for ids in batches:
        gb = g[list(ids)]
        n = gb[..., 9].sum()
        if float(n) > 0:
            per.append((gb[..., 8] * gb[..., 9]).sum() / n)
    out["rollout_ade"] = float(torch.stack(per).mean()) if per else float("nan")
    out["agents"] = int(g[..., 9].sum())
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
