"""Matrix file readers of the drivers around the path: Rutherford-Boeing with
the reference drivers' values = 3 conditioning (drivers/spllt_omp.F90:78-85) and MatrixMarket."""
import numpy as np
import pytest
import scipy.sparse as sp

from spllt_amd import matgen


def _sym(n, seed):
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=0.08, random_state=np.random.RandomState(seed), format="csr")
    S = (M + M.T).tolil()
    S.setdiag(rng.uniform(1, 2, n))
    return S.tocsc()


def test_rb_round_trip_values_as_file(tmp_path):
    A = _sym(40, 3)
    path = tmp_path / "a.rb"
    matgen.write_rb(str(path), A)
    B = matgen.read_rb(str(path), values=0)
    assert (abs(A - B) > 1e-15 * abs(A).max()).nnz == 0


def test_rb_values3_forces_diagonal_dominance(tmp_path):
    A = _sym(60, 5)
    path = tmp_path / "a.rb"
    matgen.write_rb(str(path), A)
    B = matgen.read_rb(str(path), values=3)
    off = B - sp.diags(B.diagonal())
    ref_off = A - sp.diags(A.diagonal())
    assert (abs(off - ref_off) > 1e-15).nnz == 0             # off-diagonal values are the file's
    d = B.diagonal()
    np.testing.assert_allclose(d, 1.0 + np.asarray(abs(off).sum(axis=1)).ravel(), rtol=1e-15)
    assert np.linalg.eigvalsh(B.toarray()).min() > 0           # positive definite


def test_rb_pattern_only_gets_reproducible_values(tmp_path):
    A = _sym(30, 7)
    path = tmp_path / "p.rb"
    matgen.write_rb(str(path), A, pattern_only=True)
    B1, B2 = matgen.read_rb(str(path)), matgen.read_rb(str(path))
    assert (B1 != B2).nnz == 0
    assert (B1 != 0).astype(int).sum() == (A != 0).astype(int).sum()
    assert np.linalg.eigvalsh(B1.toarray()).min() > 0
    with pytest.raises(ValueError):
        matgen.read_rb(str(path), values=0)


def test_rb_abutting_fixed_width_fields(tmp_path):
    """fields that fill their width touch each other: the reader must slice by the format"""
    path = tmp_path / "t.rb"
    n = 3
    with open(path, "w") as fh:
        fh.write(f"{'tiny':<72}{'K':<8}\n")
        fh.write(f"{3:14d}{1:14d}{1:14d}{1:14d}\n")
        fh.write(f"{'rsa':<14}{n:14d}{n:14d}{5:14d}{0:14d}\n")
        fh.write(f"{'(4I2)':<16}{'(5I2)':<16}{'(5E10.3)':<20}\n")
        fh.write(" 1 3 5 6\n")
        fh.write(" 1 2 2 3 3\n")
        fh.write(" 2.000E+00-1.000E+00 2.000E+00-1.000E+00 2.000E+00\n")
    B = matgen.read_rb(str(path), values=0).toarray()
    np.testing.assert_array_equal(B, [[2, -1, 0], [-1, 2, -1], [0, -1, 2]])
