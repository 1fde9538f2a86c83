"""Host-side model set-up of the drivers (dot-socp_amd/model.py) on the CPU: the lazy start used by the device-resident level
loop must give exactly the scalings of the eager one against the oracle's restatement of solver_dotsocp2d.m:304-365, and the
layer-by-layer mass check must agree with the reference formulation (check_massConservation.m:16-34) on any array."""
import numpy as np
import pytest

import dotsocp_amd as D
from oracle import driver as OD
from oracle import model as OM
from oracle.examples import get_example_1d, get_example_2d


@pytest.mark.parametrize("dim", [1, 2])
def test_lazy_start_gives_the_scalings_of_the_eager_one(dim):
    if dim == 2:
        rho0, rho1 = get_example_2d("example1", 40, 24)
        nt = 12
    else:
        rho0, rho1 = get_example_1d("gaussian", 64)
        nt = 10
    ov, om = OM.initialize(rho0, rho1, nt)
    OD.InitialScaling(ov, om, True, None, dim=dim)
    ev, em = D.initialize(rho0, rho1, nt)
    lv, lm = D.initialize(rho0, rho1, nt, lazy_zeros=True, phi=False)
    for v, m in ((ev, em), (lv, lm)):
        D.InitialScaling(v, m, True, None, dim=dim)
    assert lv.phi is None and lv.q is None and lv.z is None and lm._c_ends == rho0.size
    for name in ("cScale", "dScale", "D", "E", "E2"):
        assert getattr(ev, name) == getattr(ov, name)
        assert abs(getattr(lv, name) - getattr(ov, name)) <= 1e-15 * abs(getattr(ov, name))
    assert em.normc == om.normc and abs(lm.normc - om.normc) <= 1e-15 * om.normc and lm.normd == om.normd
    np.testing.assert_array_equal(em.c, om.c)
    np.testing.assert_allclose(lm.c, om.c, rtol=1e-15, atol=0)
    assert not lm.c[rho0.size:lm.c.size - rho0.size].any()          # still zero between its end layers
    # a later level (warm start) takes E2 from the previous KKT vector: same branch for both
    kkt = np.array([3e-3, 1e-3, 2e-3, 1e-3, 1e-9, 1e-3, 1e-3])
    for v, m in ((ev, em), (lv, lm)):
        D.InitialScaling(v, m, True, kkt, dim=dim)
    assert abs(lv.E - ev.E) <= 1e-15 * ev.E and abs(lv.cScale - ev.cScale) <= 1e-15 * ev.cScale


def _reference_mass_check(rho, tol):
    return bool(OD.check_massConservation(rho, tol)[0])         # the oracle's restatement of check_massConservation.m:16-34


@pytest.mark.parametrize("shape", [(9, 7, 5), (300, 300, 50), (4200000, 2)])
def test_mass_check_layer_by_layer_equals_the_reference_formulation(shape):
    rng = np.random.default_rng(3)
    rho = np.asfortranarray(np.abs(rng.standard_normal(shape)) + 0.1)
    axes = tuple(range(rho.ndim - 1))
    rho /= rho.mean(axis=axes, keepdims=True)
    assert D.check_massConservation(rho, 1e-2) and _reference_mass_check(rho, 1e-2)
    bad = rho.copy(order="F")
    bad[..., 1] *= 1.02                                            # one layer carries 2 % too much mass
    assert not D.check_massConservation(bad, 1e-2) and not _reference_mass_check(bad, 1e-2)
    assert D.check_massConservation(bad, 3e-2)
    neg = rho.copy(order="F")
    flat = neg.reshape((-1, shape[-1]), order="F")
    flat[0, 0] -= 0.05 * flat.shape[0]                             # mean of the negative part of layer 0: -0.05 + ...
    flat[1, 0] += 0.05 * flat.shape[0]
    assert D.check_massConservation(neg, 1e-2) == _reference_mass_check(neg, 1e-2) == False   # noqa: E712
    c_order = np.ascontiguousarray(rho)                            # not Fortran-ordered: the plain path
    assert D.check_massConservation(c_order, 1e-2)
