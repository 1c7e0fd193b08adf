import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")

# torch brings its own copy of the HIP runtime: it must be imported before librtc_amd.so initialises /opt/rocm's, or
# torch.cuda later reports "No HIP GPUs are available" (the GPU tests allocate device buffers through torch)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - the CPU suite does not need it
    torch = None
