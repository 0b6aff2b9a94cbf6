"""Pins the oracle's random-variate layer: Philox4x32-10 against the Random123 known-answer
vectors, the AS241 normal quantile against scipy, and the gamma / truncated-normal samplers
against their distributions (Kolmogorov-Smirnov)."""
import numpy as np
from scipy import special, stats

import oracle_lib as O


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_qnorm_matches_scipy():
    L = O.lib()
    p = np.concatenate([np.linspace(1e-12, 1 - 1e-12, 20001), 10.0 ** -np.arange(3, 300, 7.0),
                        1 - 10.0 ** -np.arange(3, 16.0)])
    got = np.array([L.orc_qnorm(float(x)) for x in p])
    ref = special.ndtri(p)
    np.testing.assert_allclose(got, ref, rtol=2e-15, atol=2e-15)


def test_pnorm_dtruncnorm():
    L = O.lib()
    for x in [-8.0, -1.3, 0.0, 0.7, 5.0]:
        assert abs(L.orc_pnorm(x) - stats.norm.cdf(x)) < 1e-15
    for (x, mu, sd) in [(0.3, 0.5, 0.2), (2.0, 1.0, 1.0), (0.01, 3.0, 0.05)]:
        ref = stats.truncnorm.logpdf(x, (0 - mu) / sd, np.inf, loc=mu, scale=sd)
        assert abs(L.orc_dtruncnorm_log(x, mu, sd, 0.0, np.inf) - ref) < 1e-12


def test_uniform_and_normal_distribution():
    u = O.fill(0, 200000)
    assert 0 < u.min() and u.max() < 1
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    z = O.fill(1, 200000, upd=7)
    assert stats.kstest(z, "norm").pvalue > 1e-3
    # different update ids / iterations give different streams
    assert not np.allclose(O.fill(1, 16, upd=7), O.fill(1, 16, upd=8))
    assert not np.allclose(O.fill(1, 16, it=0), O.fill(1, 16, it=1))
    np.testing.assert_array_equal(O.fill(1, 16, it=3, chain=2), O.fill(1, 16, it=3, chain=2))


def test_gamma_distribution():
    for shape, scale in [(0.3, 2.0), (1.0, 1.0), (2.5, 0.5), (60.0, 0.01), (20481.0, 1e-4)]:
        g = O.fill(2, 100000, p1=shape, p2=scale, upd=11)
        assert (g > 0).all()
        assert stats.kstest(g, "gamma", args=(shape, 0, scale)).pvalue > 1e-3, (shape, scale)


def test_truncnorm_distribution():
    for mu, sd in [(1.0, 0.05), (0.2, 1.0), (0.01, 0.5), (-2.0, 1.0)]:
        x = O.fill(3, 100000, p1=mu, p2=sd, upd=5)
        assert (x >= 0).all()
        a = (0 - mu) / sd
        assert stats.kstest(x, "truncnorm", args=(a, np.inf, mu, sd)).pvalue > 1e-3, (mu, sd)
