"""The sufficient-statistics ("Gram") form of the warm-start sweep on the CPU (oracle/gram.c) against the
reference-structure restatement: the same draws up to floating-point reassociation."""
import numpy as np

import oracle_lib as O
from simdata import simulate_functional, truth_chain


def test_gram_form_reproduces_the_reference_structure_sweep():
    sim = simulate_functional(n=23, M=3, sigma_sq=0.01, seed=5, ragged=True)
    T = 4
    model, ch_ref = truth_chain(sim, T)
    _, ch_gram = truth_chain(sim, T)
    rng = np.random.default_rng(9)
    for ch in (ch_ref, ch_gram):
        r2 = np.random.default_rng(9)
        ch.nu[:, :, 0] = sim["nu"] + 0.3 * r2.standard_normal(ch.nu.shape[:2])
        ch.chi[:, :, 0] = sim["chi"] + 0.2 * r2.standard_normal(ch.chi.shape[:2])
    h = O.make_hyper(sim["K"])
    O.run_sweeps(model, h, ch_ref, O.SWEEP_WARM, seed=3)
    model2 = O.Model(sim["y"], sim["B"], sim["K"], sim["M"])
    O.run_warm_gram(model2, h, ch_gram, seed=3)
    for nm in ["nu", "Phi", "chi", "Z", "pi", "alpha3", "delta", "A", "gamma", "tau", "sigma", "loglik"]:
        a, b = getattr(ch_gram, nm), getattr(ch_ref, nm)
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
        assert err < 1e-7, (nm, err)


def test_gram_form_reproduces_the_reference_structure_nu_z_sweep():
    """the Nu_Z sweep (BFMMM.h:1073-1107) in Gram form, with Phi = 0 and chi = 0 as the multi-try entry point runs it and with a
    generic Phi / chi carried through"""
    for zero in (True, False):
        sim = simulate_functional(n=19, M=2, sigma_sq=0.01, seed=6, ragged=True)
        T = 4
        model, ch_ref = truth_chain(sim, T)
        _, ch_gram = truth_chain(sim, T)
        for ch in (ch_ref, ch_gram):
            r2 = np.random.default_rng(10)
            ch.nu[:, :, 0] = sim["nu"] + 0.3 * r2.standard_normal(ch.nu.shape[:2])
            if zero:
                ch.chi[:] = 0.0
                ch.Phi[:] = 0.0
        h = O.make_hyper(sim["K"])
        O.run_sweeps(model, h, ch_ref, O.SWEEP_NU_Z, seed=4, chain_id=3)
        O.run_warm_gram(model, h, ch_gram, seed=4, chain_id=3, sweep=O.SWEEP_NU_Z)
        for nm in ["nu", "Phi", "chi", "Z", "pi", "alpha3", "delta", "A", "gamma", "tau", "sigma", "loglik"]:
            a, b = getattr(ch_gram, nm), getattr(ch_ref, nm)
            err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
            assert err < 1e-7, (zero, nm, err)
