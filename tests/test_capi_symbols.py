"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/bfmmm.h
declares (no compute calls -- there is no GPU in the build container)."""
import os
import re

import pytest


def test_library_loads_and_exports_header_symbols():
    import __graft_entry__ as g
    g.build()
    from bayesfmmm_amd import _lib
    lib = _lib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "bfmmm.h")).read()
    declared = set(re.findall(r"\b(bfmmm_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bfmmm.h but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    from bayesfmmm_amd import api
    hdr2 = open(os.path.join(root, "include", "bfmmm_entry.h")).read()
    declared2 = set(re.findall(r"\b(bfmmm_[A-Za-z_0-9]+)\s*\(", hdr2))
    for name in declared2:
        assert hasattr(lib, name), f"{name} declared in include/bfmmm_entry.h but not exported"
    assert declared2 == set(api.ENTRY_SYMBOLS), declared2 ^ set(api.ENTRY_SYMBOLS)
    hdr3 = open(os.path.join(root, "include", "bfmmm_post.h")).read()
    declared3 = set(re.findall(r"\b(bfmmm_[A-Za-z_0-9]+)\s*\(", hdr3)) - {"bfmmm_entry_last_error"}
    for name in declared3:
        assert hasattr(lib, name), f"{name} declared in include/bfmmm_post.h but not exported"
    assert declared3 == set(api.POST_SYMBOLS), declared3 ^ set(api.POST_SYMBOLS)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import bayesfmmm_amd as bf
    cfg = bf.default_config(model=0, K=2, n_eigen=2, basis_degree=3, tot_mcmc_iters=10)
    t = [np.arange(0, 100, 10.0)] * 3
    y = [np.zeros(10)] * 3
    with pytest.raises(bf._lib.BfmmmError, match="no HIP device"):
        bf.Sampler(cfg, y, t, [50.0], [0.0, 90.0])


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "bayesfmmm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() or f == "rng.hpp" and "oracle" not in src, (dirpath, f)
