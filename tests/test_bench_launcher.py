"""bench.py's own rank launcher (`python bench.py --gpus N` without torchrun) on a box without a GPU: the rank
children fail ("needs an MI355X"), and the launcher must notice, stop the others, print no JSON line and return
non-zero quickly instead of hanging in a rendezvous."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_reports_failed_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("error path of the launcher: needs a box without a GPU")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--grid", "16", "12", "8", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "2-rank run failed" in out.stderr
    assert time.time() - t0 < 120
