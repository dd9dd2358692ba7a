"""Pipelined rollouts: several engines on one GPU taking scene batches in turn, each engine's stream kept non-empty.

A rollout is ~330 dependent launches whose tail is latency-bound, so several rollouts in flight (one per engine, each on its own
NON-BLOCKING stream) share the chip.  Round 5 removed what a NEW batch used to cost on top of a resident one:
* ``ps_set_scene`` stages its uploads through pinned host memory (two arenas in turn) and ends without a stream synchronisation;
* ``ps_rollout`` keeps the captured hipGraph when the batch has the shape of the previous one (it compares the signature of its
  launch sequence -- row counts, flags, device pointers -- instead of re-capturing ~330 launches and instantiating them);
* results leave through ``ps_get_async`` into pinned memory behind the rollout, with an event per batch;
so an engine can hold ``queue`` (2) batches at once: batch n + depth is uploaded and launched BEHIND batch n on the same stream
while the host is still waiting for n's results, and the stream never runs dry between two rollouts.  Measured on one MI355X,
8 x 128-agent scenes per batch (tools/gpu_pipeline_depth.py, bench.py `streaming`): DESIGN.md section 5.

The reference runs its batches strictly one after the other (rollout/callbacks.py); this is the serving-side
counterpart of its M-replica fan-out (rollout/gpu_utils.py:59-123): independent batches, no data exchanged.
"""
from __future__ import annotations

from collections import deque
from typing import Deque, Dict, Iterable, Iterator, List, Sequence, Tuple

import numpy as np
import torch

from .engine import Engine
from .spec import ModelSpec

DEFAULT_OUTPUTS = ("traj", "vel")
# per-agent rows that are zeroed for log-replay agents and scattered to the padded [B, N, ...] slot layout (Engine.padded)
_PADDED = ("traj", "vel", "policy_emd", "reconst_pred", "fused")


class _Ticket:
    __slots__ = ("slot", "bufs", "event", "policy_rows", "slots", "shape", "shapes")


class RolloutPipeline:
    """``depth`` engines on one device, up to ``queue`` batches per engine; ``run(scenes)`` yields ``(index, outputs)`` in
    submission order.  ``outputs``: names ``Engine.ASYNC_RESULTS`` knows ('traj', 'vel', 'motion_pred', ...)."""

    def __init__(self, spec: ModelSpec, weights: Dict[str, np.ndarray], device: int = 0, depth: int = 3,
                 outputs: Sequence[str] = DEFAULT_OUTPUTS, queue: int = 2):
        if depth < 1 or queue < 1:
            raise ValueError("depth and queue must be >= 1")
        bad = [n for n in outputs if n not in Engine.ASYNC_RESULTS]
        if bad:
            raise ValueError(f"outputs {bad}: the pipeline reads results back asynchronously, which covers {Engine.ASYNC_RESULTS}")
        from . import hw_queues_configured
        if depth > 1 and hw_queues_configured() < depth + 2:   # (engine streams + the upload stream + the default stream)
            import warnings
            warnings.warn(f"RolloutPipeline(depth={depth}): GPU_MAX_HW_QUEUES is {hw_queues_configured()} -- engines that share a hardware queue "
                          "serialise their rollouts; call prosim_amd.configure_runtime() before the first GPU call of the process", RuntimeWarning, stacklevel=2)
        self.engines: List[Engine] = [Engine(spec, weights, device=device) for _ in range(depth)]
        if depth > 1:   # throughput mode (k_chain16, 16 rows per workgroup): the launches of the rollouts in flight share the chip
            for e in self.engines:
                e.set_chain_rows(16)
        self.outputs = tuple(outputs)
        self.queue = int(queue)
        dev = torch.device("cuda", device)
        self._streams = [torch.cuda.ExternalStream(e.stream_handle, device=dev) for e in self.engines]
        self._pending: List[Deque[int]] = [deque() for _ in range(depth)]    # tickets each engine holds, oldest first
        self._tickets: Dict[int, _Ticket] = {}
        self._pool: List[List[Dict[str, torch.Tensor]]] = [[] for _ in range(depth)]   # free pinned buffer sets per engine
        self._next = 0

    @property
    def capacity(self) -> int:
        """Batches the pipeline holds before ``submit`` asks for a ``collect``."""
        return len(self.engines) * self.queue

    # ---- low level: submit / collect -------------------------------------------------------------------------------
    def _buffers(self, slot: int, shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, torch.Tensor]:
        bufs = self._pool[slot].pop() if self._pool[slot] else {}
        for name, shp in shapes.items():
            n = int(np.prod(shp))
            if name not in bufs or bufs[name].numel() < n:
                bufs[name] = torch.empty(max(n, 16), dtype=torch.float32, pin_memory=True)
        return bufs

    def submit(self, scene: Dict[str, np.ndarray]) -> int:
        """Upload ``scene`` to the next engine, launch its rollout and the copies of its results (all asynchronous: the call
        returns while the GPU may still be working on the engine's previous batch).  The engine must hold fewer than ``queue``
        batches: collect its oldest ticket first."""
        slot = self._next % len(self.engines)
        if len(self._pending[slot]) >= self.queue:
            raise RuntimeError(f"engine {slot} still holds tickets {list(self._pending[slot])}: collect it before submitting more "
                               f"than {self.capacity} batches")
        eng = self.engines[slot]
        eng.set_scene(scene)
        eng.rollout()
        t = _Ticket()
        t.slot = slot
        t.shapes = {n: eng.result_shape(n) for n in self.outputs}
        t.bufs = self._buffers(slot, t.shapes)
        for n, shp in t.shapes.items():
            eng.get_async(n, t.bufs[n].data_ptr(), t.bufs[n].numel())
        t.event = torch.cuda.Event()
        t.event.record(self._streams[slot])
        # (the engine's host-side view of the batch is gone with its next set_scene: keep what the scatter below needs)
        t.policy_rows, t.slots, t.shape = eng.policy_rows.copy(), eng.row_slots, eng._shape
        ticket = self._next
        self._tickets[ticket] = t
        self._pending[slot].append(ticket)
        self._next += 1
        return ticket

    def collect(self, ticket: int) -> Dict[str, np.ndarray]:
        """Wait for the results of ``ticket`` (padded ``[B, N, ...]`` layout; 'motion_pred' and other per-row arrays come back as
        the engine stores them).  Tickets of one engine are collected oldest first."""
        t = self._tickets.get(ticket)
        if t is None:
            raise RuntimeError(f"ticket {ticket} is not in flight")
        if self._pending[t.slot][0] != ticket:
            raise RuntimeError(f"ticket {ticket} is not the oldest of its engine: collect {self._pending[t.slot][0]} first")
        t.event.synchronize()
        out = {}
        B, N = t.shape
        for name, shp in t.shapes.items():
            a = t.bufs[name][:int(np.prod(shp))].numpy().reshape(shp)
            if name in _PADDED:
                a = np.where(t.policy_rows.reshape((-1,) + (1,) * (a.ndim - 1)), a, 0.0).astype(np.float32)   # log-replay rows
                full = np.zeros((B * N,) + a.shape[1:], np.float32)
                full[t.slots] = a
                out[name] = full.reshape((B, N) + a.shape[1:])
            else:
                out[name] = a.copy()
        out["policy_rows"] = t.policy_rows
        self._pool[t.slot].append(t.bufs)
        self._pending[t.slot].popleft()
        del self._tickets[ticket]
        return out

    # ---- the usual loop ----------------------------------------------------------------------------------------------
    def run(self, scenes: Iterable[Dict[str, np.ndarray]]) -> Iterator[Tuple[int, Dict[str, np.ndarray]]]:
        """Keep up to ``capacity`` batches queued over ``scenes``; yields ``(index, outputs)`` in order."""
        oldest = self._next
        for scene in scenes:
            if self._next - oldest >= self.capacity:
                yield oldest, self.collect(oldest)
                oldest += 1
            self.submit(scene)
        while oldest < self._next:
            yield oldest, self.collect(oldest)
            oldest += 1

    def close(self) -> None:
        for e in self.engines:
            e.sync()
            e.close()
        self.engines = []
        self._tickets.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
