"""prosim_amd -- MI355X-native closed-loop rollout engine for ProSim-style traffic simulation.

Only the rollout hot path lives here: the HIP kernels + C-ABI in ``csrc/``,
the ctypes binding in ``engine.py`` and the host-side mirror of the reference's
scene-encoder / decoder / policy plugin interface in ``modules.py``.
"""
from .spec import ModelSpec, DEMO_SPEC, SMALL_SPEC  # noqa: F401

import os as _os
import warnings as _warnings


def configure_runtime(hw_queues: int = 24) -> bool:
    """One hardware queue per stream: call this BEFORE the process makes its first HIP call (before torch touches the GPU, before the
    first Engine).  The HIP runtime folds streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; a serving process runs several
    engines (non-blocking streams) beside the upload stream, torch's own streams and RCCL's, and engines that share a queue serialise
    their rollouts: 15 - 16 M instead of 27 M agent-steps/s with four rollouts in flight as soon as the process holds more streams than
    queues (tools/gpu_stream_variants.py).  The variable is read once, at the runtime's initialisation; a value the caller exported
    wins.  Returns False (and warns) when HIP is already initialised in this process -- the setting then has no effect.
    (Round 6, ADVICE round 5: importing the package no longer changes the process environment; bench.py, RolloutPipeline's users and
    the tools call this explicitly.)"""
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))
    try:
        import sys as _sys
        torch = _sys.modules.get("torch")
        if torch is not None and torch.cuda.is_initialized():
            _warnings.warn("prosim_amd.configure_runtime: HIP is already initialised in this process; GPU_MAX_HW_QUEUES=%s will not take "
                           "effect (call it before the first GPU call)" % _os.environ["GPU_MAX_HW_QUEUES"], RuntimeWarning, stacklevel=2)
            return False
    except Exception:
        pass
    return True


def hw_queues_configured() -> int:
    """GPU_MAX_HW_QUEUES as the runtime will read it (its own default, 4, when the variable is unset)."""
    try:
        return int(_os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4
