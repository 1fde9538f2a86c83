"""The closed-form problem generators of the reference's examples/ directory (dot-socp_amd/examples.py): shapes,
normalisation (get_example.m:45-46: mean 1 after the lower bound), a few values worked out by hand from the
formulas, and -- on the GPU -- that the drivers solve them (KKT below the tolerance, mass conserved)."""
import numpy as np
import pytest

import dotsocp_amd as D
from dotsocp_amd import examples as E


@pytest.mark.parametrize("name", sorted(E._EXAMPLES_2D))
def test_2d_generators_are_normalised(name):
    n = 33
    rho0, rho1 = D.get_example_2d(name, n, n)
    assert rho0.shape == rho1.shape == (n, n)
    assert rho0.min() >= 0 and rho1.min() >= 0
    np.testing.assert_allclose([rho0.mean(), rho1.mean()], 1.0, rtol=1e-13)
    lb0, lb1 = D.get_example_2d(name, n, n, 0.1)
    np.testing.assert_allclose(lb0, (rho0 + 0.1) / 1.1, rtol=1e-13)
    np.testing.assert_allclose([lb0.mean(), lb1.mean()], 1.0, rtol=1e-13)


def test_values_from_the_formulas():
    r0, r1 = E.gene_example2(5, 5)               # grid 0, .25, .5, .75, 1 in both directions
    assert r0[1, 1] == 1.0 and r0[0, 0] == pytest.approx(np.exp(-(2 * 0.25 ** 2) / (2 * 0.1 ** 2)))
    assert r1[1, 1] == pytest.approx(1 + 2 * np.exp(-0.25 / (2 * 0.05 ** 2)) + np.exp(-0.5 / (2 * 0.05 ** 2)))
    r0, _ = E.gene_example3(5, 5)
    assert r0[1, 1] == pytest.approx(np.e) and r0[1, 3] == pytest.approx(np.exp(np.exp(-5 * 0.5)))   # 5 |y - b| along a row
    r0, _ = E.gene_example4(5, 5)
    assert r0[2, 2] == 0.0 and r0[0, 4] == pytest.approx(2 * 0.5 ** 4)
    r0, r1 = E.gene_exampleCircle(41, 41)
    assert r0[30, 10] == 1 and r0[10, 30] == 0 and r1[10, 30] == 1     # centres (x, y) = (.25, .75) and (.75, .25)
    b0, b1 = E.gene_example_box(101)
    assert 40 <= b0.sum() <= 41 and 10 <= b1.sum() <= 11 and b0[30] == 1 and b1[90] == 1 and b0[60] == 0
    _, r1 = E.gene_example7(101, 101)
    assert r1.sum() == 30 and r1[82, 44] == 1               # first listed point (0.8323, 0.4477): rho1(83, 45) in MATLAB
    w = E.gene_weight_circle(4, 9, 9)
    assert w.size == 9 * 9 * 3 + 2 * 9 * 8 * 4 and np.all(w[:243] == 1)
    np.testing.assert_allclose(w[243:243 + 72].sum(), 72.0)             # one time layer of x edges, normalised
    heart = E.gene_barrier_of_love_heart()
    assert heart(np.array(0.02), np.array(0.02)) and not heart(np.array(0.5), np.array(0.3))


def test_unknown_names_are_rejected():
    with pytest.raises(ValueError):
        D.get_example_2d("example5", 9, 9)        # reads image files in the reference: not restated
    with pytest.raises(ValueError):
        D.get_example_1d("triangle", 9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["example2", "example4", "circle"])
def test_gpu_solves_the_2d_examples(name):
    rho0, rho1 = D.get_example_2d(name, 65, 65, 1e-3)
    out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, 17, 2, dict(tol=1e-3, maxit=5000), "inPALM")
    assert np.max(hist["kkt"][-1][[0, 2, 5, 6]]) < 1e-3
    assert D.check_massConservation(out["rho"], 1e-2)
    np.testing.assert_allclose(out["rho"][:, :, 0], rho0, atol=1e-12)
    np.testing.assert_allclose(out["rho"][:, :, -1], rho1, atol=1e-12)


@pytest.mark.gpu
def test_gpu_solves_the_1d_box_and_the_heart_barrier():
    rho0, rho1 = D.get_example_1d("box", 129, 1e-3)
    out, _, _, hist = D.solver_dotsocp1d(rho0, rho1, 33, 1, dict(tol=1e-3, maxit=20000), "inPALM")
    assert np.max(hist["kkt"][-1][[0, 2, 5, 6]]) < 1e-3 and D.check_massConservation(out["rho"], 1e-2)
    n, nt = 65, 17
    barrier = D.gene_barrier_of_love_heart()
    rho0, rho1 = D.get_example_2d("love-heart", n, n)
    rho0, rho1, mask = D.ensure_barrier_validity(rho0, rho1, barrier)
    weight = D.get_weight_by_barrier(n, n, nt, barrier)
    out, _, _, hist = D.solver_wdotsocp2d(rho0, rho1, nt, 1, dict(tol=1e-3, maxit=8000, weight=weight), "inPALM", barrier)
    assert np.max(hist["kkt"][-1][[0, 2, 5]]) < 1e-3
    assert D.check_massConservation(out["rho"], 2e-2)
    assert mask.any() and not mask.all()
