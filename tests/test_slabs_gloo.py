"""The N > 1 path on the CPU: the time-slab decomposition (owned layers, halo exchanges E1-E5,
slab<->pencil transposes of the Poisson solve, all-reduced KKT sums) restated in
oracle/slab_oracle.py must reproduce the single-process oracle -- world_size 1 in process,
world_size 2 and 3 as separate processes over torch.distributed's gloo backend."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import driver as OD                      # noqa: E402
from oracle.examples import get_example_2d           # noqa: E402
from oracle.inpalm import InPALMState                # noqa: E402
from oracle.slab_oracle import GlooComm, LocalComm, SlabInPALM, slab_range   # noqa: E402

NY, NX, NT, K = 12, 10, 9, 14


def _level():
    rho0, rho1 = get_example_2d("example1", NY, NX)
    return OD.make_level(rho0, rho1, NT, dict(tol=0.0, maxit=K))


def _reference():
    var, model, o = _level()
    st = InPALMState(var, o, model)
    st.run()
    F = lambda a, shp: np.asarray(a).reshape(shp, order="F")
    qi = var.qInd
    return dict(phi=F(st.phi, (NY, NX, NT)), q0=F(st.q[:qi.bx], (NY, NX, NT - 1)),
                qbx=F(st.q[qi.bx:qi.by], (NY, NX - 1, NT)), qby=F(st.q[qi.by:], (NY - 1, NX, NT)),
                a0=F(st.alpha[:qi.bx], (NY, NX, NT - 1)), beta=F(st.beta, (NY, NX, NT - 1, 10)),
                z=F(st.z, (NY, NX, NT - 1, 10)), sigma=st.sigma, kkt=np.array(st.kkt_hist), iters=st.iter_hist)


def _check(slab, ref):
    t0, ntl, ncl = slab.t0, slab.ntl, slab.ncl
    tol = 1e-10
    def close(a, b):
        assert np.max(np.abs(a - b)) <= tol * max(np.max(np.abs(b)), 1.0)
    close(slab.phi, ref["phi"][:, :, t0:t0 + ntl])
    close(slab.q0, ref["q0"][:, :, t0:t0 + ncl])
    close(slab.qbx, ref["qbx"][:, :, t0:t0 + ntl])
    close(slab.qby, ref["qby"][:, :, t0:t0 + ntl])
    close(slab.a0, ref["a0"][:, :, t0:t0 + ncl])
    close(slab.beta, ref["beta"][:, :, t0:t0 + ncl])
    close(slab.z, ref["z"][:, :, t0:t0 + ncl])
    assert abs(slab.sigma - ref["sigma"]) <= 1e-12 * ref["sigma"]
    assert [h[0] for h in slab.hist] == list(ref["iters"])
    np.testing.assert_allclose(np.array([h[1] for h in slab.hist]), ref["kkt"], rtol=1e-8, atol=1e-13)


def test_slab_oracle_world1_equals_oracle():
    var, model, o = _level()
    slab = SlabInPALM(var, o, model, LocalComm()).run()
    _check(slab, _reference())


def test_slab_ranges_cover_grid():
    for nt, w in [(9, 2), (9, 3), (128, 8), (33, 4)]:
        edges = [slab_range(nt, w, r) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == nt
        assert all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))


def _worker(rank, world, port, q, tsolve):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        var, model, o = _level()
        slab = SlabInPALM(var, o, model, GlooComm(), tsolve=tsolve).run()
        _check(slab, _reference())
        q.put((rank, "ok"))
    except Exception as e:            # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tsolve", ["tridiag", "dct"])
@pytest.mark.parametrize("world", [2, 3])
def test_slab_oracle_gloo(world, tsolve):
    """The time-slab algorithm restated on the CPU, one process per slab over gloo, with both ways of crossing the
    slabs in the Poisson solve (partitioned tridiagonal systems = the device default; transposes)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, tsolve)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"
