"""Pins the oracle's B-spline / penalty code against the reference's own golden files
(inst/test-data/Tensor_BSpline.txt, P_mat.txt; src/test-BSplines.cpp:9-52,66,81) and against
scipy's clamped design matrix."""
import os

import numpy as np
from scipy.interpolate import BSpline

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_arma_ascii(path):
    with open(path) as f:
        assert f.readline().startswith("ARMA_MAT_TXT")
        r, c = map(int, f.readline().split())
        data = np.array(f.read().split(), dtype=np.float64).reshape(r, c)
    return data


def test_tensor_bspline_golden():
    # src/test-BSplines.cpp:9-28
    t = np.arange(0, 1000, 10.0)
    tt = np.stack([t, t], axis=1)
    B = O.tensor_bspline(tt, [3, 3], [[0, 990], [0, 990]], [[250, 500, 750], [250, 500, 750]])
    gold = load_arma_ascii(os.path.join(GOLD, "Tensor_BSpline.txt"))
    assert B.shape == gold.shape == (100, 49)
    assert np.abs(B - gold).max() <= 1e-7      # reference tolerance (absdiff 1e-7), test-BSplines.cpp:66
    assert np.abs(B - gold).max() <= 1e-15     # and in fact exact to the printed precision


def test_pmat_golden():
    # src/test-BSplines.cpp:34-52
    Pm = O.get_P([3, 3], [3, 3])
    gold = load_arma_ascii(os.path.join(GOLD, "P_mat.txt"))
    assert Pm.shape == gold.shape == (49, 49)
    assert np.abs(Pm - gold).max() <= 1e-7     # test-BSplines.cpp:81


def test_univariate_basis_properties_and_scipy():
    rng = np.random.default_rng(0)
    for degree, n_int in [(3, 26), (3, 4), (2, 5), (1, 3)]:
        b0, b1 = 0.0, 990.0
        ik = np.linspace(b0, b1, n_int + 2)[1:-1]
        x = np.concatenate([[b0, b1], rng.uniform(b0, b1, 200), ik])
        B = O.bspline_basis(x, ik, degree, [b0, b1])
        P = n_int + degree + 1
        assert B.shape == (len(x), P)
        np.testing.assert_allclose(B.sum(axis=1), 1.0, atol=1e-14)   # partition of unity
        assert (B >= 0).all()
        assert B[1, -1] == 1.0 and B[0, 0] == 1.0                     # clamped ends, right end inclusive
        knots = np.concatenate([[b0] * (degree + 1), ik, [b1] * (degree + 1)])
        ref = BSpline.design_matrix(x, knots, degree).toarray()
        np.testing.assert_allclose(B, ref, atol=1e-14)
        # band structure used by the HIP path: B'B has half-bandwidth = degree
        G = B.T @ B
        i, j = np.nonzero(np.abs(G) > 0)
        assert np.abs(i - j).max() <= degree


def test_rw1_penalty():
    Pm = O.pmat_rw1(6)
    D1 = np.diff(np.eye(6), axis=0)
    np.testing.assert_array_equal(Pm, D1.T @ D1)
