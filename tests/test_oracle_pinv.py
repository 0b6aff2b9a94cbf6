"""The oracle's route for rank-deficient conditional precisions (oracle/linalg.c; the reference: arma::pinv + the
eigen-decomposition fallback of arma::mvnrnd, UpdateNu.h:67-69, UpdateEta.h:85-87) on its own -- CPU only."""
import numpy as np

import oracle_lib as O


def test_oracle_pinv_route_is_the_pseudo_inverse_law():
    """the shared specification on its own (CPU): mean = pinv(Prec) rhs, covariance = pinv(Prec)"""
    import ctypes as C
    L = O.lib()
    L.orc_prec_is_singular.restype = C.c_int
    P = 9
    Pm = O.pmat_rw1(P) * 1.7
    assert L.orc_prec_is_singular(P, O.dp(np.asfortranarray(Pm))) == 1
    assert L.orc_prec_is_singular(P, O.dp(np.asfortranarray(Pm + 1e-3 * np.eye(P)))) == 0
    rhs = np.random.default_rng(0).standard_normal(P)
    out = np.zeros(P)
    L.orc_pinv_draw(P, O.dp(np.asfortranarray(Pm)), O.dp(rhs), O.dp(np.zeros(P)), O.dp(out))
    np.testing.assert_allclose(out, np.linalg.pinv(Pm) @ rhs, rtol=1e-10, atol=1e-12)
    draws = []
    rng = np.random.default_rng(1)
    for _ in range(4000):
        z = rng.standard_normal(P)
        L.orc_pinv_draw(P, O.dp(np.asfortranarray(Pm)), O.dp(np.zeros(P)), O.dp(z), O.dp(out))
        draws.append(out.copy())
    cov = np.cov(np.array(draws).T)
    assert np.abs(cov - np.linalg.pinv(Pm)).max() < 0.15 * np.abs(np.linalg.pinv(Pm)).max()


