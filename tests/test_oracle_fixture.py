"""The committed oracle fixture (tests/golden/oracle_warm_sweep.npz, made by tests/golden/make_oracle_fixtures.py) pins the
oracle: rebuilding it here must give the committed numbers (a change in oracle/ that alters any draw shows up as a diff)."""
import os
import sys

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)


def test_oracle_reproduces_its_committed_fixture():
    import make_oracle_fixtures as mk
    ref = np.load(os.path.join(GOLD, "oracle_warm_sweep.npz"))
    _, _, _, out = mk.build()
    assert set(out) == set(ref.files)
    for k in ref.files:
        np.testing.assert_allclose(out[k], ref[k], rtol=1e-12, atol=1e-14, err_msg=k)
    assert np.isfinite(ref["chain_loglik"]).all() and ref["chain_sigma"].min() > 0
