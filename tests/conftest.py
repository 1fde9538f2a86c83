import os
import sys

import pytest

# torch ships its own copy of the HIP runtime; whichever copy is loaded first serves the whole process.  The binding takes
# care of the order by itself (capi._share_torchs_hip_runtime, tests/test_gpu_import_order.py); importing torch here first
# merely keeps the suite on the path most of it was written on.
try:
    import torch  # noqa: F401
except Exception:       # noqa: BLE001
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C part of the oracle is compiled once per session (gcc, < 1 s)."""
    from oracle import mexops
    mexops.build()
