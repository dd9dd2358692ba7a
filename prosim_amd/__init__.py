"""prosim_amd -- MI355X-native closed-loop rollout engine for ProSim-style traffic simulation.

Only the rollout hot path lives here: the HIP kernels + C-ABI in ``csrc/``,
the ctypes binding in ``engine.py`` and the host-side mirror of the reference's
scene-encoder / decoder / policy plugin interface in ``modules.py``.
"""
from .spec import ModelSpec, DEMO_SPEC, SMALL_SPEC  # noqa: F401
