"""The product's tensor-product B-spline basis and penalty (bayesfmmm_amd/csrc/tensor_basis.cpp: TensorBSpline / GetP of
inst/include/BayesFMMM/BSplines.h:18-120) against the reference's own golden files (src/test-BSplines.cpp:9-52, tolerance
1e-7 at :66, :81) and against scipy's clamped design matrix.  Host code only."""
import os

import numpy as np
import pytest
from scipy.interpolate import BSpline

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def api():
    import __graft_entry__ as g
    g.build()
    from bayesfmmm_amd import api
    return api


def load_arma_ascii(path):
    with open(path) as f:
        assert f.readline().startswith("ARMA_MAT_TXT")
        r, c = map(int, f.readline().split())
        return np.array(f.read().split(), dtype=np.float64).reshape(r, c)


def test_tensor_bspline_golden(api):
    t = np.arange(0, 1000, 10.0)                   # src/test-BSplines.cpp:9-28
    B = api.TensorBSpline(np.stack([t, t], axis=1), [3, 3], [[0, 990], [0, 990]], [[250, 500, 750], [250, 500, 750]])
    gold = load_arma_ascii(os.path.join(GOLD, "Tensor_BSpline.txt"))
    assert B.shape == gold.shape == (100, 49)
    assert np.abs(B - gold).max() <= 1e-7          # the reference's tolerance
    assert np.abs(B - gold).max() <= 1e-15


def test_penalty_golden(api):
    Pm = api.GetP([3, 3], [3, 3])                  # src/test-BSplines.cpp:34-52
    gold = load_arma_ascii(os.path.join(GOLD, "P_mat.txt"))
    assert Pm.shape == gold.shape == (49, 49)
    assert np.abs(Pm - gold).max() <= 1e-7


def test_against_scipy_and_structure(api):
    rng = np.random.default_rng(1)
    degs, iks, bks = [3, 2, 1], [np.array([2.0, 5.0]), np.array([0.3]), np.array([10.0, 20.0, 30.0])], [[0, 9], [0, 1], [0, 40]]
    t = np.stack([rng.uniform(lo, hi, 57) for lo, hi in bks], axis=1)
    t[0] = [9.0, 1.0, 40.0]                        # every right boundary (inclusive)
    t[1] = [0.0, 0.0, 0.0]
    B = api.TensorBSpline(t, degs, bks, iks)
    uni = []
    for l in range(3):
        kn = np.concatenate([[bks[l][0]] * (degs[l] + 1), iks[l], [bks[l][1]] * (degs[l] + 1)])
        uni.append(BSpline.design_matrix(t[:, l], kn, degs[l]).toarray())
    ref = np.einsum("ka,kb,kc->kabc", *uni).reshape(57, -1)      # last dimension fastest (BSplines.h:29-31, 56-60)
    np.testing.assert_allclose(B, ref, atol=1e-14)
    np.testing.assert_allclose(B.sum(axis=1), 1.0, atol=1e-13)
    assert B[0, -1] == 1.0 and B[1, 0] == 1.0
    Pm = api.GetP(degs, [2, 1, 3])
    assert np.allclose(Pm, Pm.T) and np.allclose(Pm.sum(axis=1), 0.0) and np.linalg.eigvalsh(Pm).min() > -1e-12
    # one dimension: the RW1 penalty of BFMMM.h:1027-1037
    D1 = np.diff(np.eye(6), axis=0)
    np.testing.assert_array_equal(api.GetP([3], [2]), D1.T @ D1)
