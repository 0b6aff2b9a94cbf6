"""The Armadillo file formats of the reference's on-disk chain batches (BFMMM.h:1680-1746), pinned by the files the
reference ships: every trace file under tests/golden/ (copies of inst/test-data/Functional_trace/*, fieldmat.txt,
fieldvec.txt) is read with the ReadVec / ReadMat / ReadCube / ReadField* counterparts and must be reproduced byte for
byte by the writers.  Host code only (no GPU)."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TRACE = os.path.join(GOLD, "Functional_trace")


@pytest.fixture(scope="module")
def api():
    import __graft_entry__ as g
    g.build()
    from bayesfmmm_amd import api
    return api


def _is_field(path):
    return open(path, "rb").read(12) == b"ARMA_FLD_BIN"


def test_every_shipped_trace_file_round_trips_byte_for_byte(api, tmp_path):
    files = sorted(glob.glob(os.path.join(TRACE, "*.txt"))) + [os.path.join(GOLD, "fieldmat.txt"), os.path.join(GOLD, "fieldvec.txt")]
    assert len(files) >= 15
    for f in files:
        out = str(tmp_path / os.path.basename(f))
        if _is_field(f):
            api.write_arma_field(out, api.ReadFieldCube(f))
        else:
            x = api.ReadCube(f)
            api.write_arma_ascii(out, x)
        assert open(out, "rb").read() == open(f, "rb").read(), os.path.basename(f)


def test_shapes_and_quirks_of_the_shipped_batch(api):
    # the shipped batch: K = 2, P = 7 (cubic, 3 internal knots), M = 3, D = 1, 150 saved draws
    nu = api.ReadCube(os.path.join(TRACE, "Nu0.txt"))
    assert nu.shape == (2, 7, 150)
    assert api.ReadMat(os.path.join(TRACE, "Pi0.txt")).shape == (2, 150)
    assert api.ReadMat(os.path.join(TRACE, "Tau0.txt")).shape == (150, 2)
    sig = api.ReadVec(os.path.join(TRACE, "Sigma0.txt"))
    assert sig.shape == (150,) and np.all(sig > 0)
    a3 = api.ReadVec(os.path.join(TRACE, "alpha_30.txt"))
    assert a3[0] == 0.0 and np.all(a3[1:] > 0)        # alpha_31(0) is never assigned (BFMMM.h:1686, 1700-1711)
    phi = api.ReadFieldCube(os.path.join(TRACE, "Phi0.txt"))
    assert phi.shape == (150, 1) and phi[0, 0].shape == (2, 7, 3)
    xi = api.ReadFieldCube(os.path.join(TRACE, "Xi0.txt"))
    assert xi.shape == (150, 2) and xi[3, 1].shape == (7, 1, 3)
    fv = api.ReadFieldVec(os.path.join(GOLD, "fieldvec.txt"))
    assert fv.shape == (20, 1) and fv[0, 0].shape == (5,)
    fm = api.ReadFieldMat(os.path.join(GOLD, "fieldmat.txt"))
    assert fm.shape[1] == 1 and fm[0, 0].ndim == 2


def test_reader_errors(api, tmp_path):
    from bayesfmmm_amd import _lib
    with pytest.raises(_lib.BfmmmError, match="cannot open"):
        api.ReadMat(str(tmp_path / "missing.txt"))
    bad = tmp_path / "bad.txt"
    bad.write_bytes(b"ARMA_MAT_TXT_FN008\n2 2\n 1.0 2.0\n")
    with pytest.raises(_lib.BfmmmError, match="truncated"):
        api.ReadMat(str(bad))
    with pytest.raises(_lib.BfmmmError, match="ARMA_FLD_BIN"):
        api.ReadFieldCube(str(bad))


def test_special_values_and_empty_objects(api, tmp_path):
    f = str(tmp_path / "m.txt")
    x = np.array([[1.5, -np.inf], [np.nan, 2.0 ** -1060]])
    api.write_arma_ascii(f, x)
    y = api.ReadMat(f)
    assert y[0, 0] == 1.5 and y[0, 1] == -np.inf and np.isnan(y[1, 0]) and y[1, 1] == x[1, 1]
    g = str(tmp_path / "f.txt")
    fld = np.empty((3, 1), dtype=object)
    fld[0, 0] = np.arange(6.0).reshape(1, 2, 3)
    fld[1, 0] = np.zeros((0, 0, 0))
    fld[2, 0] = np.ones((2, 1, 1))
    api.write_arma_field(g, fld)
    back = api.ReadFieldCube(g)
    assert back.shape == (3, 1) and back[1, 0].size == 0 and np.array_equal(back[0, 0], fld[0, 0])
