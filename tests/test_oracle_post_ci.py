"""CPU checks of the numpy restatement of arma::quantile (oracle/post_ci.py) against known answers of Hyndman & Fan's
definition 5 (R: quantile(x, p, type = 5)), and of SigmaCI on the trace the package ships."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import post_ci as R      # noqa: E402


def test_quantile_definition_5_known_answers():
    x = [4.0, 1.0, 3.0, 2.0]
    # quantile(1:4, c(0, .1, .25, .5, .75, .9, 1), type = 5) = 1, 1, 1.5, 2.5, 3.5, 4, 4
    np.testing.assert_allclose(R.arma_quantile(x, [0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0]), [1, 1, 1.5, 2.5, 3.5, 4, 4], rtol=1e-15)
    y = [10.0, 20.0, 30.0, 40.0, 50.0]
    # quantile(c(10,20,30,40,50), c(.3, .5, .62), type = 5) = 20, 30, 36
    np.testing.assert_allclose(R.arma_quantile(y, [0.3, 0.5, 0.62]), [20.0, 30.0, 36.0], rtol=1e-14)
    assert R.arma_quantile(y, [-0.1])[0] == -np.inf and R.arma_quantile(y, [1.1])[0] == np.inf


def test_kept_range_and_sigma_ci_quirk():
    assert R.kept_count(150, 0.1) == 135 and R.kept_count(150, 0.73) == 41 and R.kept_count(7, 0.5) == 4      # std::round half away from zero
    sig = np.linspace(1.0, 2.0, 101)
    ci = R.sigma_ci(sig, 0.05, 0.0)
    assert ci["CI_Lower"] == ci["CI_50"] == 1.5              # PostProcessing.cpp:3496
    assert abs(ci["CI_Upper"] - (1.0 + 0.01 * (101 * 0.975 - 0.5))) < 1e-12
