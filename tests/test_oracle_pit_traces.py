"""Conditional-PIT check of the oracle's update formulas against REAL output of the R package (CPU only).

The reference ships three saved chains (inst/test-data/{Functional,Multivariate,HDFunctional}_trace, copied to tests/golden/):
150 saved draws of the covariate-adjusted warm-start sampler each (BFMMM.h:4809-4894 and its MV / HD counterparts), written
with thinning_num = 1, so rows r >= 1 of every file are the consecutive iterations r - 1 (row 0 and row 1 both hold slot 0,
BFMMM.h:1686-1711).  The covariates the chains were fitted with are not shipped (the Rd examples draw `X <- rnorm(...)`), so
the data-dependent blocks cannot be replayed; the hyper-parameter blocks can, because everything they condition on sits in
the saved rows:

    delta | Phi, gamma, A      UpdateDelta.h:17-64        gamma    | Phi, delta      UpdateGamma.h:17-37
    tau   | nu                 UpdateTau.h:18-36 / :47     tau_eta  | eta             UpdateTau.h:75-95 / :106
    delta_xi | xi, gamma_xi, A_xi   UpdateDelta.h:76-124   gamma_xi | xi, delta_xi    UpdateGamma.h:48-72

For every saved iteration the oracle's update function (oracle/updates.c) is run on the state the reference had at that
point of its sweep, with its gamma sampler hooked (orc_rgamma_hook): each draw is replaced by the reference's own saved draw
and the (shape, scale) the oracle asked for is recorded.  The oracle therefore walks the reference's sequence of
conditionals, and the reference's draw pushed through the recorded Gamma CDF must be U(0, 1) if -- and only if -- the
oracle's conditional is the law the R package sampled from (R::rgamma itself is exact).  This pins the shape / rate
formulas (integer divisions, the cumulative products, the MV model's inverted tau, the tensor-product penalty) to outputs
of the real reference; the device is then tied to the oracle by the -m gpu parity tests.
"""
import ctypes as C
import os

import numpy as np
import pytest
from scipy import stats

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# sweep order of the covariate-adjusted warm-start drivers (BFMMM.h:4809-4894; MV :6230-6420; HD :7860-8050)
ORDER = ["Phi", "delta", "A", "gamma", "nu", "tau", "eta", "tau_eta", "xi", "delta_xi", "A_xi", "gamma_xi"]
UPD = dict(delta=8, gamma=11, tau=13, tau_eta=17, delta_xi=19, gamma_xi=22)       # oracle.h update ids


@pytest.fixture(scope="module")
def api():
    import __graft_entry__ as g
    g.build()
    from bayesfmmm_amd import api
    return api


def _load(api, dirn):
    d = os.path.join(GOLD, dirn) + "/"
    t = dict(nu=api.ReadCube(d + "Nu0.txt"), delta=api.ReadCube(d + "Delta0.txt"), A=api.ReadCube(d + "A0.txt"),
             tau=api.ReadMat(d + "Tau0.txt"), tau_eta=api.ReadCube(d + "Tau_Eta0.txt"))
    for nm, f in [("Phi", "Phi0"), ("gamma", "Gamma0"), ("eta", "Eta0"), ("xi", "Xi0"), ("gamma_xi", "Gamma_Xi0"),
                  ("delta_xi", "Delta_Xi0"), ("A_xi", "A_Xi0")]:
        t[nm] = api.ReadFieldCube(d + f + ".txt")
    return t


def _row(t, nm, r):
    v = t[nm]
    if v.dtype == object:
        if v.shape[1] == 1:
            return v[r, 0]
        return np.stack([v[r, k] for k in range(v.shape[1])], axis=-1)      # K cubes P x D x M -> (P, D, M, K)
    return v[r, :] if nm == "tau" else v[..., r]


def _hook():
    L = O.lib()
    L.orc_rgamma_hook.restype = None
    L.orc_rgamma_hook.argtypes = [C.c_int, C.c_uint32, O.c_double_p, O.c_double_p, C.c_int]
    return L


def conditional_pits(api, dirn, U, mv=False, Pmat=None, **hyper):
    """PIT values of the reference's saved draws of block U under the oracle's conditionals (all saved iterations)."""
    t = _load(api, dirn)
    K, P, n_rows = t["nu"].shape
    M, D = t["delta"].shape[1], t["tau_eta"].shape[1]
    # the hyper-parameter updates touch no data: two dummy curves carry the dimensions
    model = O.Model([np.zeros(P)] * 2, [np.eye(P)] * 2, K, M, X=np.zeros((2, D)), mv=mv, Pmat=Pmat)
    h = dict(O.HYPER_DEFAULTS)       # the Rd examples run with the entry points' defaults
    h.update(hyper)
    L = _hook()
    out = []
    for r in range(2, n_rows):
        ch = O.Chain(model, 2)
        # state at the point of the sweep where U runs: blocks updated before U hold this iteration's draw (row r),
        # the others still hold the previous iteration's (row r - 1)
        ch.set_slot0(**{nm: _row(t, nm, r if ORDER.index(nm) < ORDER.index(U) else r - 1) for nm in ORDER})
        x = _row(t, U, r)
        if U == "delta":
            inj = [x[k, i] for k in range(K) for i in range(M)]
        elif U == "gamma":
            inj = [x[i, l, j] for i in range(K) for l in range(P) for j in range(M)]
        elif U == "tau":          # the MV model stores 1 / rgamma (UpdateTau.h:58)
            inj = [(1 / x[i] if mv else x[i]) for i in range(K)]
        elif U == "tau_eta":
            inj = [(1 / x[j, i] if mv else x[j, i]) for j in range(K) for i in range(D)]
        elif U == "delta_xi":
            inj = [x[k, i, dd] for dd in range(D) for k in range(K) for i in range(M)]
        else:
            inj = [x[l, i, j, k] for k in range(K) for i in range(D) for l in range(P) for j in range(M)]
        inj = np.ascontiguousarray(inj, dtype=np.float64)
        rec = np.full(2 * len(inj), np.nan)
        L.orc_rgamma_hook(1, UPD[U], O.dp(inj), O.dp(rec), len(inj))
        try:
            if U == "delta":
                O.updateDelta(model, ch, 0)
            elif U == "gamma":
                O.updateGamma(model, ch, 0, h["nu_1"])
            elif U == "tau":
                O.updateTau(model, ch, 0, h["alpha_nu"], h["beta_nu"])
            elif U == "tau_eta":
                O.updateTauEta(model, ch, 0, h["alpha_eta"], h["beta_eta"])
            elif U == "delta_xi":
                O.updateDeltaXi(model, ch, 0)
            else:
                O.updateGammaXi(model, ch, 0, h["nu_1"])
        finally:
            L.orc_rgamma_hook(0, 0, None, None, 0)
        assert np.isfinite(rec).all()          # every draw of the block went through the hook
        out.append(stats.gamma.cdf(inj, rec[0::2], scale=rec[1::2]))
    return np.concatenate(out)


def _trace_kwargs(dirn):
    if dirn == "Multivariate_trace":
        return dict(mv=True)
    if dirn == "HDFunctional_trace":       # quadratic splines, knots 250/500/750 in both dimensions (man/BHDFMMM_warm_start.Rd)
        return dict(Pmat=O.get_P([2, 2], [3, 3]))
    return {}


@pytest.mark.parametrize("dirn", ["Functional_trace", "Multivariate_trace", "HDFunctional_trace"])
@pytest.mark.parametrize("U", list(UPD))
def test_reference_draws_are_uniform_under_the_oracle_conditionals(api, dirn, U):
    u = conditional_pits(api, dirn, U, **_trace_kwargs(dirn))
    assert len(u) >= 2 * 148
    assert (u > 0).all() and (u < 1).all()
    p = stats.kstest(u, "uniform").pvalue
    assert p > 0.01, (dirn, U, p, u.mean())
    assert abs(u.mean() - 0.5) < 4.0 / np.sqrt(12 * len(u)), (dirn, U, u.mean())


@pytest.mark.parametrize("U,wrong", [("gamma", dict(nu_1=2.0)), ("gamma_xi", dict(nu_1=4.0)), ("tau", dict(alpha_nu=5.0)),
                                     ("tau_eta", dict(alpha_eta=5.0))])
def test_the_check_has_power(api, U, wrong):
    """Negative control: the same statistic rejects a conditional with a wrong hyper-parameter."""
    u = conditional_pits(api, "Functional_trace", U, **wrong)
    assert stats.kstest(u, "uniform").pvalue < 1e-4


def test_multivariate_tau_must_be_inverted(api):
    """UpdateTau.h:58 stores 1 / rgamma for the multivariate model: reading the saved tau as the gamma draw itself
    (the functional convention) is rejected."""
    u = conditional_pits(api, "Multivariate_trace", "tau", mv=False, Pmat=np.eye(10))
    assert stats.kstest(u, "uniform").pvalue < 1e-4
