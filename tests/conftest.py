import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # one hardware queue per stream for the multi-engine tests: before anything touches the GPU (prosim_amd.configure_runtime;
    # importing the package no longer sets it)
    import prosim_amd
    prosim_amd.configure_runtime()


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
