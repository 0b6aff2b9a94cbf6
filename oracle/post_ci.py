"""TEST INFRASTRUCTURE ONLY: numpy restatement of the reference's credible-interval functions over saved draws
(src/PostProcessing.cpp): SigmaCI :3435-3503, ZCI :3505-3572, FMeanCI :99-700 (without and with covariates, pointwise and
simultaneous, rescale / trans_mats).  arma::quantile is Armadillo's (RcppArmadillo, a dependency that is not vendored in the
reference): Hyndman & Fan definition 5, restated from its published algorithm (op_quantile).  Parity unpinned: the reference
ships no expected values for these functions."""
import numpy as np


def arma_quantile(v, probs):
    y = np.sort(np.asarray(v, dtype=np.float64))
    N = float(len(y))
    out = []
    for p in probs:
        if p < 0.5 / N:
            out.append(-np.inf if p < 0 else y[0])
        elif p > (N - 0.5) / N:
            out.append(np.inf if p > 1 else y[-1])
        else:
            k = int(np.floor(N * p + 0.5))
            pk = (k - 0.5) / N
            w = (p - pk) * N
            out.append((1 - w) * y[k - 1] + w * y[min(k, len(y) - 1)])
    return np.array(out)


def kept_count(T, burnin_prop):
    return int(np.floor((T * (1 - burnin_prop)) + 0.5))        # std::round of a positive number


def sigma_ci(sigma, alpha, burnin_prop):
    kept = kept_count(len(sigma), burnin_prop)
    q = arma_quantile(sigma[len(sigma) - kept:], [alpha / 2, 0.5, 1 - alpha / 2])
    return dict(CI_Upper=q[2], CI_50=q[1], CI_Lower=q[1])     # :3496: CI_Lower = q(1) in the reference


def transform_mat(Zj):
    K = Zj.shape[1]
    T = np.zeros((K, K))
    for i in range(K):
        T[i, :] = Zj[int(np.argmax(Zj[:, i])), :]              # arma::index_max: first maximum
    return T


def z_ci(Z, alpha, rescale, burnin_prop):
    Z = np.array(Z, dtype=np.float64, copy=True)
    n, K, T = Z.shape
    if rescale and K > 2:
        rescale = False
    if rescale:
        for j in range(T):
            Tm = transform_mat(Z[:, :, j])
            Z[:, :, j] = np.linalg.solve(Tm.T, Z[:, :, j].T).T
    kept = kept_count(T, burnin_prop)
    up, md, lo = np.zeros((n, K)), np.zeros((n, K)), np.zeros((n, K))
    for i in range(n):
        for j in range(K):
            q = arma_quantile(Z[i, j, T - kept:], [alpha / 2, 0.5, 1 - alpha / 2])
            up[i, j], md[i, j], lo[i, j] = q[2], q[1], q[0]
    first = int(np.floor(T * burnin_prop + 0.5))
    return dict(CI_Upper=up, CI_50=md, CI_Lower=lo, Z_trace=Z[:, :, first:])


def bands(f, alpha, simultaneous):
    """f: draws x time points"""
    nt = f.shape[1]
    if not simultaneous:
        q = np.array([arma_quantile(f[:, i], [alpha / 2, 0.5, 1 - alpha / 2]) for i in range(nt)])
        return q[:, 2], q[:, 1], q[:, 0]
    mean, sd = f.mean(axis=0), f.std(axis=0, ddof=1)
    C = np.max(np.abs((f - mean) / sd), axis=1)
    q = arma_quantile(C, [1 - alpha])[0]
    return mean + q * sd, mean, mean - q * sd


def f_mean_ci(nu, B, k, alpha, rescale, simultaneous, burnin_prop, Z=None, X=None, eta=None, trans_mats=None):
    """nu (K, P, T); B (n_time, P); eta (P, D, K, T); Z (n, K, T); X (n_x, D)"""
    K, P, T = nu.shape
    kept = kept_count(T, burnin_prop)
    nu = np.array(nu[:, :, T - kept:], copy=True)
    eta = None if eta is None else np.array(eta[..., T - kept:], copy=True)
    if rescale and K > 2:
        rescale = False
    for j in range(kept):
        Tm = None
        if rescale:
            Tm = transform_mat(Z[:, :, T - kept + j])
        elif trans_mats is not None:
            Tm = trans_mats[j * K:(j + 1) * K, :K]
        if Tm is not None:
            nu[:, :, j] = Tm @ nu[:, :, j]
            if eta is not None:
                for d in range(eta.shape[1]):
                    eta[:, d, :, j] = (Tm @ eta[:, d, :, j].T).T
    if X is None:
        f = np.stack([B @ nu[k - 1, :, j] for j in range(kept)])
        up, md, lo = bands(f, alpha, simultaneous)
        return dict(CI_Upper=up, CI_50=md, CI_Lower=lo, mean_trace=f)
    nx = X.shape[0]
    nt = B.shape[0]
    fs = np.zeros((nx, nt, kept))
    for n_ in range(nx):
        for j in range(kept):
            fs[n_, :, j] = B @ (nu[k - 1, :, j] + eta[:, :, k - 1, j] @ X[n_])
    up, md, lo = np.zeros((nx, nt)), np.zeros((nx, nt)), np.zeros((nx, nt))
    for n_ in range(nx):
        up[n_], md[n_], lo[n_] = bands(fs[n_].T, alpha, simultaneous)
    return dict(CI_Upper=up, CI_50=md, CI_Lower=lo, mean_trace=fs)


def f_cov_ci(Phi, B1, B2, l, m, alpha, rescale, simultaneous, burnin_prop, Z=None, trans_mats=None):
    """FCovCI :1781-2050 without covariates.  Phi (K, P, M, T); both the rescale transform and trans_mats apply when given."""
    K, P, M, T = Phi.shape
    kept = kept_count(T, burnin_prop)
    Phi = np.array(Phi[..., T - kept:], copy=True)
    if rescale and K > 2:
        rescale = False
    n1, n2 = B1.shape[0], B2.shape[0]
    cov = np.zeros((n1, n2, kept))
    for j in range(kept):
        if rescale:
            Tm = transform_mat(Z[:, :, T - kept + j])
            for b in range(M):
                Phi[:, :, b, j] = Tm @ Phi[:, :, b, j]
        if trans_mats is not None:
            Tm = trans_mats[j * K:(j + 1) * K, :K]
            for b in range(M):
                Phi[:, :, b, j] = Tm @ Phi[:, :, b, j]
        for b in range(M):
            cov[:, :, j] += np.outer(B1 @ Phi[l - 1, :, b, j], B2 @ Phi[m - 1, :, b, j])
    flat = cov.reshape(n1 * n2, kept, order="F").T             # draws x cells, cell = s1 + n1 s2
    up, md, lo = bands(flat, alpha, simultaneous)
    sh = lambda v: v.reshape(n1, n2, order="F")
    return dict(CI_Upper=sh(up), CI_50=sh(md), CI_Lower=sh(lo), cov_trace=cov)


def f_cov_ci_x(Phi, xi, X, B1, B2, l, m, alpha, rescale, simultaneous, burnin_prop, Z=None):
    """The covariate branch of FCovCI (:2051-2330), HDFCovCI (:2740-3010) and MVCovCI (:3240-3405): the covariance between
    clusters l and m at every covariate setting X[b], sum_j (B1 (phi_lj + xi_lj x_b)) (B2 (phi_mj + xi_mj x_b))'.
    Phi (K, P, M, T); xi (P, D, M, K, T); X (n_x, D).  No trans_mats in this branch; the rescale transform acts on Phi and,
    as xi_k <- sum_b T(k, b) xi_b (:2175-2185), on xi.  Returns bands (n1, n2, n_x) and cov_trace (n1, n2, kept, n_x)."""
    K, P, M, T = Phi.shape
    kept = kept_count(T, burnin_prop)
    Phi = np.array(Phi[..., T - kept:], copy=True)
    xi = np.array(xi[..., T - kept:], copy=True)
    if rescale and K > 2:
        rescale = False
    n1, n2, nx = B1.shape[0], B2.shape[0], X.shape[0]
    cov = np.zeros((n1, n2, kept, nx))
    for j in range(kept):
        if rescale:
            Tm = transform_mat(Z[:, :, T - kept + j])
            for b in range(M):
                Phi[:, :, b, j] = Tm @ Phi[:, :, b, j]
            old = xi[..., j].copy()
            for k in range(K):
                acc = old[..., 0] * Tm[k, 0]
                for b in range(1, K):
                    acc = acc + old[..., b] * Tm[k, b]
                xi[..., k, j] = acc
        for b in range(M):
            for w in range(nx):
                cl = Phi[l - 1, :, b, j] + xi[:, :, b, l - 1, j] @ X[w]
                cm = Phi[m - 1, :, b, j] + xi[:, :, b, m - 1, j] @ X[w]
                cov[:, :, j, w] += np.outer(B1 @ cl, B2 @ cm)
    up, md, lo = (np.zeros((n1, n2, nx)) for _ in range(3))
    for w in range(nx):
        flat = cov[..., w].reshape(n1 * n2, kept, order="F").T
        u, c, d = bands(flat, alpha, simultaneous)
        up[..., w], md[..., w], lo[..., w] = (v.reshape(n1, n2, order="F") for v in (u, c, d))
    return dict(CI_Upper=up, CI_50=md, CI_Lower=lo, cov_trace=cov)


def mv_mean_ci(nu, alpha, rescale, burnin_prop, Z=None, X=None, eta=None):
    """MVMeanCI :1410-1660.  nu (K, P, T); eta (P, D, K, T); X (n_x, D)."""
    K, P, T = nu.shape
    kept = kept_count(T, burnin_prop)
    nu = np.array(nu[:, :, T - kept:], copy=True)
    eta = None if eta is None else np.array(eta[..., T - kept:], copy=True)
    if rescale and K > 2:
        rescale = False
    if rescale:
        for j in range(kept):
            Tm = transform_mat(Z[:, :, T - kept + j])
            nu[:, :, j] = Tm @ nu[:, :, j]
            if eta is not None:
                for d in range(eta.shape[1]):
                    eta[:, d, :, j] = (Tm @ eta[:, d, :, j].T).T
    probs = [alpha / 2, 0.5, 1 - alpha / 2]
    if X is None:
        q = np.array([[arma_quantile(nu[k, i, :], probs) for i in range(P)] for k in range(K)])
        return dict(CI_Upper=q[..., 2], CI_50=q[..., 1], CI_Lower=q[..., 0], mean_trace=nu)
    nx = X.shape[0]
    ms = np.zeros((K, P, kept, nx))
    for j in range(nx):
        for i in range(kept):
            for k in range(K):
                ms[k, :, i, j] = nu[k, :, i] + eta[:, :, k, i] @ X[j]
    q = np.array([[[arma_quantile(ms[k, i, :, j], probs) for j in range(nx)] for i in range(P)] for k in range(K)])
    return dict(CI_Upper=q[..., 2], CI_50=q[..., 1], CI_Lower=q[..., 0], mean_trace=ms)
