"""Synthetic data in the style of the reference's unit tests (src/test-Nu.cpp:11-72 and
siblings): t = 0:10:990, cubic B-splines with 8 degrees of freedom, the fixed 3 x 8 nu table."""
import numpy as np

import oracle_lib as O

NU_TABLE = np.array([[2, 0, 1, 0, 0, 0, 1, 3],
                     [1, 3, 0, 2, 0, 0, 3, 0],
                     [5, 2, 5, 0, 3, 4, 1, 0]], dtype=np.float64)


def simulate_functional(n, M, sigma_sq, seed, K=3, D=0, phi_scale=0.1, alpha_dir=10.0, n_pts=100,
                        ragged=False, min_pi=0.15):
    rng = np.random.default_rng(seed)
    t_full = np.arange(0, 10.0 * n_pts, 10.0)
    P = 8
    nu = NU_TABLE[:K].copy()
    Phi = np.zeros((K, P, M))
    for m in range(M):
        Phi[:, :, m] = (M - m) * phi_scale * rng.uniform(size=(K, P))
    chi = rng.standard_normal((n, M))
    pi = rng.dirichlet(np.ones(K))
    while pi.min() < min_pi:          # keep every cluster identifiable at these tiny sample sizes
        pi = rng.dirichlet(np.ones(K))
    Z = rng.dirichlet(pi * alpha_dir, size=n)
    Z = np.clip(Z, 1e-12, None)
    Z /= Z.sum(axis=1, keepdims=True)
    X = eta = xi = None
    if D > 0:
        X = rng.standard_normal((n, D))
        eta = rng.standard_normal((P, D, K))
        xi = np.zeros((P, D, M, K))
        for m in range(M):
            xi[:, :, m, :] = (M - m) * phi_scale * rng.uniform(size=(P, D, K))
    ts, Bs, ys = [], [], []
    B_full = O.bspline_df(t_full, 8)
    ik = np.quantile(t_full, np.arange(1, 5) / 5)
    for i in range(n):
        if ragged:
            keep = np.sort(rng.choice(n_pts, size=rng.integers(n_pts // 2, n_pts + 1), replace=False))
        else:
            keep = np.arange(n_pts)
        t = t_full[keep]
        B = B_full[keep]
        c = np.zeros(P)
        for k in range(K):
            u = nu[k].copy()
            if D > 0:
                u += eta[:, :, k] @ X[i]
            for m in range(M):
                v = Phi[k, :, m].copy()
                if D > 0:
                    v += xi[:, :, m, k] @ X[i]
                u += chi[i, m] * v
            c += Z[i, k] * u
        y = B @ c + np.sqrt(sigma_sq) * rng.standard_normal(len(t))
        ts.append(t); Bs.append(B); ys.append(y)
    return dict(t=ts, B=Bs, y=ys, nu=nu, Phi=Phi, chi=chi, pi=pi, Z=Z, X=X, eta=eta, xi=xi,
                sigma_sq=sigma_sq, internal_knots=ik, boundary_knots=np.array([t_full[0], t_full[-1]]),
                K=K, P=P, M=M, D=D, n=n)


def truth_chain(sim, T):
    """Chain object with every slot pre-filled with the simulation truth (the reference's
    single-update tests hold all other parameters at truth)."""
    model = O.Model(sim["y"], sim["B"], sim["K"], sim["M"], X=sim["X"])
    ch = O.Chain(model, T)
    ch.nu[:] = sim["nu"][:, :, None]
    ch.Phi[:] = sim["Phi"][..., None]
    ch.chi[:] = sim["chi"][:, :, None]
    ch.Z[:] = sim["Z"][:, :, None]
    ch.pi[:] = sim["pi"][:, None]
    ch.sigma[:] = sim["sigma_sq"]
    ch.alpha3[:] = 10.0
    ch.tau[:] = 1.0
    ch.delta[:] = 1.0
    ch.A[:] = 1.0
    ch.gamma[:] = 1.0
    if sim["D"] > 0:
        ch.eta[:] = sim["eta"][..., None]
        ch.xi[:] = sim["xi"][..., None]
        ch.tau_eta[:] = 1.0
        ch.gamma_xi[:] = 1.0
        ch.delta_xi[:] = 1.0
        ch.A_xi[:] = 1.0
    return model, ch
