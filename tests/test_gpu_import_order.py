"""libdotsocp and PyTorch in one process, in either import order: the binding loads torch's copy of the HIP runtime before
libdotsocp when torch is installed but not yet imported (capi._share_torchs_hip_runtime), so a later `import torch` still
finds the GPU.  Each order runs in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SOLVE = """
import numpy as np
rho0, rho1 = D.get_example_2d("example1", 16, 16)
var, model = D.initialize(rho0, rho1, 8)
D.InitialScaling(var, model, True, None, dim=2)
hist, sigma = D.solver_socp_inPALM(var, dict(tau=1.9, sigma=1.0, tol=0.0, maxit=5, scaling=True, ifCheckStepByStep=False), model)
assert np.all(np.isfinite(var.phi)) and hist["len"] >= 1
"""


@pytest.mark.parametrize("order", ["dotsocp_first", "torch_first"])
def test_both_import_orders(order):
    if order == "dotsocp_first":
        code = "import sys\nassert 'torch' not in sys.modules\nimport dotsocp_amd as D\n" + SOLVE + \
               "import torch\nassert torch.cuda.is_available(), 'torch lost the GPU'\n" \
               "x = torch.ones(8, device='cuda', dtype=torch.float64)\nassert float(x.sum()) == 8.0\n" + SOLVE + "print('ok')\n"
    else:
        code = "import torch\nassert torch.cuda.is_available()\nimport dotsocp_amd as D\n" + SOLVE + \
               "x = torch.ones(8, device='cuda', dtype=torch.float64)\nassert float(x.sum()) == 8.0\nprint('ok')\n"
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-1000:] + out.stderr[-3000:]
