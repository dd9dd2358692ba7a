"""prosim_amd -- MI355X-native closed-loop rollout engine for ProSim-style traffic simulation.

Only the rollout hot path lives here: the HIP kernels + C-ABI in ``csrc/``,
the ctypes binding in ``engine.py`` and the host-side mirror of the reference's
scene-encoder / decoder / policy plugin interface in ``modules.py``.
"""
from .spec import ModelSpec, DEMO_SPEC, SMALL_SPEC  # noqa: F401

import os as _os

# One hardware queue per stream.  The HIP runtime folds streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; a serving
# process runs several engines (non-blocking streams) beside the upload stream, torch's own streams and RCCL's, and engines that
# share a queue serialise their rollouts: 15 - 16 M instead of 27 M agent-steps/s with four rollouts in flight as soon as the
# process holds more streams than queues (tools/gpu_stream_variants.py; four idle engines beside a depth-4 pipeline at 8 queues).  Read by the
# runtime when the process makes its first HIP call, so it has to be in the environment before that -- importing this package first
# is enough; a value the caller exported wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
