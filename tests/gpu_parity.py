"""Shared helpers of the GPU parity tests: build the HIP sampler and the CPU oracle on the same
simulated data / state / RNG key and compare them.  (Test infrastructure.)"""
import numpy as np

import oracle_lib as O
from simdata import simulate_functional, truth_chain

STATE_NAMES = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq"]
ORC_FIELD = {"alpha_3": "alpha3", "sigma_sq": "sigma"}


def make_sampler(sim, T, **cfg_kw):
    import bayesfmmm_amd as bf
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=sim["K"], n_eigen=sim["M"], basis_degree=3,
                            tot_mcmc_iters=T, **cfg_kw)
    return bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])


def oracle_slot(ch, name, slot):
    arr = getattr(ch, ORC_FIELD.get(name, name))
    if name == "tau":
        return arr[slot, :].copy()
    if arr.ndim == 1:
        return np.array([arr[slot]])
    return arr[..., slot].copy()


def push_state(sampler, ch, slot=0):
    """copy slot `slot` of the oracle chain into the sampler's current state"""
    sampler.set_state(**{nm: oracle_slot(ch, nm, slot) for nm in STATE_NAMES})


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


def random_state(sim, ch, seed):
    """a generic (not truth) state in slot 0 so that every term of every update is exercised"""
    rng = np.random.default_rng(seed)
    n, K, P, M = sim["n"], sim["K"], sim["P"], sim["M"]
    ch.nu[:, :, 0] = sim["nu"] + 0.3 * rng.standard_normal((K, P))
    ch.Phi[..., 0] = sim["Phi"] + 0.1 * rng.standard_normal((K, P, M))
    ch.chi[:, :, 0] = sim["chi"] + 0.2 * rng.standard_normal((n, M))
    Z = rng.dirichlet(np.full(K, 2.0), size=n)
    ch.Z[:, :, 0] = Z
    ch.pi[:, 0] = rng.dirichlet(np.full(K, 5.0))
    ch.alpha3[0] = 3.0 + rng.uniform()
    ch.delta[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, M))
    ch.A[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, 2))
    ch.gamma[..., 0] = rng.gamma(2.0, 0.7, size=(K, P, M))
    ch.tau[0, :] = rng.gamma(3.0, 0.5, size=K)
    ch.sigma[0] = sim["sigma_sq"] * (1.0 + rng.uniform())


def gram_reference(sim, Z, chi, MD):
    """numpy Gram-form quantities the device computes: per-curve records and pair-weighted sums"""
    n, K, P, M = sim["n"], sim["K"], sim["P"], sim["M"]
    BW = 3
    G = np.stack([B.T @ B for B in sim["B"]])
    s = np.stack([B.T @ y for B, y in zip(sim["B"], sim["y"])])
    yy = np.array([y @ y for y in sim["y"]])
    band = np.zeros((n, BW + 1, P))
    for d in range(BW + 1):
        for p in range(P - d):
            band[:, d, p] = G[:, p, p + d]
    chit = np.concatenate([np.ones((n, 1)), chi], axis=1)[:, :MD]
    W = np.einsum("ij,im->ijm", Z, chit).reshape(n, K * MD)       # a = j*MD + mt
    H = {}
    for a in range(K * MD):
        for b in range(K * MD):
            H[(a, b)] = np.einsum("i,ipq->pq", W[:, a] * W[:, b], G)
    tvec = W.T @ s
    return dict(G=G, s=s, yy=yy, band=band, W=W, H=H, tvec=tvec)


def smoke():
    """one small invocation of the hot path on cuda:0, checked against the oracle"""
    import bayesfmmm_amd as bf
    sim = simulate_functional(n=24, M=2, sigma_sq=0.01, seed=123)
    T = 4
    model, ch = truth_chain(sim, T)
    random_state(sim, ch, 5)
    h = O.make_hyper(sim["K"])
    smp = make_sampler(sim, T)
    push_state(smp, ch)
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, n_iter=T, seed=11, chain_id=0)
    smp.run(bf.SWEEP_WARM, T, seed=11, chain=0)
    worst = 0.0
    for nm in ["nu", "Phi", "chi", "Z", "sigma_sq", "loglik"]:
        got = smp.get_chain(nm)
        ref = getattr(ch, ORC_FIELD.get(nm, nm))
        worst = max(worst, rel_err(got, ref))
    assert worst < 1e-6, f"smoke parity failed: rel err {worst}"
    print(f"smoke ok: {T} warm-start sweeps on the GPU match the CPU oracle, worst rel err {worst:.3e}")
