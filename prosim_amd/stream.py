"""Pipelined rollouts: several engines on one GPU taking scene batches in turn.

A rollout is ~330 dependent launches whose tail is latency-bound, and ``ps_set_scene`` of a new batch is ~1 ms of host
work plus its upload.  Engines own their device buffers and a NON-BLOCKING stream each, so while one engine's
rollout drains on the GPU the host prepares, captures and launches the next batch on another engine and the two
overlap on the device.  Measured on one MI355X, 8 x 128-agent scenes per batch (tools/gpu_stream_scenes.py,
tools/gpu_pipeline_depth.py, bench.py): over a stream of NEW batches, results read back, depth 1: 7.6 M agent-steps/s,
2: 9.7 M, 3: 10.2 M, 4: 9.4 M; 10.7 M for a resident batch with three rollouts in flight.

The reference runs its batches strictly one after the other (rollout/callbacks.py); this is the serving-side
counterpart of its M-replica fan-out (rollout/gpu_utils.py:59-123): independent batches, no data exchanged.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .engine import Engine
from .spec import ModelSpec

DEFAULT_OUTPUTS = ("traj", "vel")


class RolloutPipeline:
    """``depth`` engines on one device; ``run(scenes)`` yields ``(index, outputs)`` in submission order."""

    def __init__(self, spec: ModelSpec, weights: Dict[str, np.ndarray], device: int = 0, depth: int = 3,
                 outputs: Sequence[str] = DEFAULT_OUTPUTS):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.engines: List[Engine] = [Engine(spec, weights, device=device) for _ in range(depth)]
        if depth > 1:   # throughput mode (k_chain16, 16 rows per workgroup): the launches of the rollouts in flight share the chip
            for e in self.engines:
                e.set_chain_rows(16)
        self.outputs = tuple(outputs)
        self._pending: List[Optional[int]] = [None] * depth      # ticket each engine is working on
        self._next = 0

    # ---- low level: submit / collect -------------------------------------------------------------------------------
    def submit(self, scene: Dict[str, np.ndarray]) -> int:
        """Upload ``scene`` to the next engine and launch its rollout (asynchronous).  The engine must be free:
        collect its previous ticket first."""
        slot = self._next % len(self.engines)
        if self._pending[slot] is not None:
            raise RuntimeError(f"engine {slot} still holds ticket {self._pending[slot]}: collect it before submitting more "
                               f"than {len(self.engines)} batches")
        eng = self.engines[slot]
        eng.set_scene(scene)
        eng.rollout()
        ticket = self._next
        self._pending[slot] = ticket
        self._next += 1
        return ticket

    def collect(self, ticket: int) -> Dict[str, np.ndarray]:
        """Wait for the rollout of ``ticket`` and read its outputs back (padded ``[B, N, ...]`` layout; 'motion_pred'
        and other per-row arrays come back as the engine stores them)."""
        slot = ticket % len(self.engines)
        if self._pending[slot] != ticket:
            raise RuntimeError(f"ticket {ticket} is not in flight")
        eng = self.engines[slot]
        eng.sync()
        out = {}
        for name in self.outputs:
            out[name] = eng.padded(name) if name in ("traj", "vel", "policy_emd", "reconst_pred", "fused") else eng.get(name)
        out["policy_rows"] = eng.policy_rows.copy()
        self._pending[slot] = None
        return out

    # ---- the usual loop ----------------------------------------------------------------------------------------------
    def run(self, scenes: Iterable[Dict[str, np.ndarray]]) -> Iterator[Tuple[int, Dict[str, np.ndarray]]]:
        """Keep ``depth`` rollouts in flight over ``scenes``; yields ``(index, outputs)`` in order."""
        oldest = self._next
        for scene in scenes:
            if self._next - oldest >= len(self.engines):
                yield oldest, self.collect(oldest)
                oldest += 1
            self.submit(scene)
        while oldest < self._next:
            yield oldest, self.collect(oldest)
            oldest += 1

    def close(self) -> None:
        for e in self.engines:
            e.close()
        self.engines = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
