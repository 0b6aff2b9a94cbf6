"""CPU checks of the oracle's tempered-transition restatement (BFMMM.h:1452-1460, 1556-1672;
CalculateTTAcceptance.h:22-97)."""
import numpy as np

import oracle_lib as O
from gpu_parity import random_state
from simdata import simulate_functional, truth_chain


def _chain(seed, T):
    sim = simulate_functional(n=23, M=2, sigma_sq=0.01, seed=seed, ragged=True)
    model, ch = truth_chain(sim, T)
    random_state(sim, ch, seed + 100)
    return sim, model, ch


def test_ladder_is_the_references_geometric_one():
    # beta_ladder(N_t - 1) = beta_N_t is overwritten by the loop: rung i is geom_mult^i, geom_mult = beta_N_t^(1/N_t)
    lad = O.beta_ladder(5, 0.2)
    g = 0.2 ** (1 / 5)
    np.testing.assert_allclose(lad, g ** np.arange(5), rtol=1e-15)
    np.testing.assert_allclose(O.beta_ladder(1, 0.3), [0.3])


def test_unit_ladder_always_accepts_with_zero_log_ratio():
    # beta_N_t = 1: every rung is 1, so the acceptance ratio is exactly 1 (log 0) and the proposal is always accepted
    sim, model, ch = _chain(3, 7)
    h = O.make_hyper(sim["K"])
    logA, acc = O.run_warm_tt(model, h, ch, N_t=3, n_temp_trans=2, beta_N_t=1.0, seed=2)
    blocks = [2, 4, 6]
    assert (logA[blocks] == 0.0).all() and (acc[blocks] == 1).all()
    assert np.isnan(logA[[0, 1, 3, 5]]).all()
    assert np.isfinite(ch.loglik).all() and np.allclose(ch.Z.sum(axis=1), 1.0)


def test_blocks_leave_earlier_iterations_untouched_and_are_reproducible():
    sim, model, ch0 = _chain(4, 8)
    h = O.make_hyper(sim["K"])
    _, _, ch1 = _chain(4, 8)
    _, _, ch2 = _chain(4, 8)
    O.run_sweeps(model, h, ch0, O.SWEEP_WARM, seed=6)
    la1, ac1 = O.run_warm_tt(model, h, ch1, N_t=2, n_temp_trans=3, beta_N_t=0.5, seed=6)
    la2, ac2 = O.run_warm_tt(model, h, ch2, N_t=2, n_temp_trans=3, beta_N_t=0.5, seed=6)
    np.testing.assert_array_equal(ch1.nu, ch2.nu)
    np.testing.assert_array_equal(la1[[3, 6]], la2[[3, 6]])
    # iterations 0..2 are the plain warm-start sweep; the first block acts on slot 3
    np.testing.assert_array_equal(ch1.nu[:, :, :3], ch0.nu[:, :, :3])
    np.testing.assert_array_equal(ch1.chi[:, :, :3], ch0.chi[:, :, :3])
    assert np.isfinite(la1[[3, 6]]).all() and set(ac1[[3, 6]]) <= {0, 1}
