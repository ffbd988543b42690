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


# ---- the library's readers (C boundary) against the Python ones ---------------------------
def _lower_csc(A):
    L = sp.tril(sp.csc_matrix(A), format="csc")
    L.sort_indices()
    return L.shape[0], L.indptr + 1, L.indices + 1, L.data


@pytest.mark.parametrize("values", [0, 3])
def test_c_reader_rb_matches_python(tmp_path, values):
    """spllt_hip_read_rb (include/spllt_hip.h; what the reference takes from SPRAL's rb_read,
    drivers/spllt_omp.F90:78-85) on the files this module writes: same pattern, same values."""
    A = _sym(60, 11)
    path = tmp_path / "a.rb"
    matgen.write_rb(str(path), A)
    n, ptr, row, val = matgen.read_file_c(str(path), "rb", values=values)
    en, eptr, erow, eval_ = _lower_csc(matgen.read_rb(str(path), values=values))
    assert n == en and np.array_equal(ptr, eptr) and np.array_equal(row, erow)
    np.testing.assert_allclose(val, eval_, rtol=1e-15, atol=0)


def test_c_reader_rb_pattern_only_and_abutting_fields(tmp_path):
    A = _sym(30, 7)
    path = tmp_path / "p.rb"
    matgen.write_rb(str(path), A, pattern_only=True)
    n, ptr, row, val = matgen.read_file_c(str(path), "rb", values=3, seed=5)
    en, eptr, erow, eval_ = _lower_csc(matgen.read_rb(str(path), values=3, seed=5))
    assert n == en and np.array_equal(ptr, eptr) and np.array_equal(row, erow)
    np.testing.assert_allclose(val, eval_, rtol=1e-15, atol=0)      # the same made-up values
    with pytest.raises(ValueError):
        matgen.read_file_c(str(path), "rb", values=0)
    path = tmp_path / "t.rb"
    with open(path, "w") as fh:
        fh.write(f"{'tiny':<72}{'K':<8}\n")
        fh.write(f"{3:14d}{1:14d}{1:14d}{1:14d}\n")
        fh.write(f"{'rsa':<14}{3:14d}{3:14d}{5:14d}{0:14d}\n")
        fh.write(f"{'(4I2)':<16}{'(5I2)':<16}{'(5E10.3)':<20}\n")
        fh.write(" 1 3 5 6\n")
        fh.write(" 1 2 2 3 3\n")
        fh.write(" 2.000E+00-1.000E+00 2.000E+00-1.000E+00 2.000E+00\n")
    n, ptr, row, val = matgen.read_file_c(str(path), "rb", values=0)
    assert n == 3 and list(ptr) == [1, 3, 5, 6] and list(row) == [1, 2, 2, 3, 3]
    np.testing.assert_array_equal(val, [2, -1, 2, -1, 2])


@pytest.mark.parametrize("symmetric", [True, False])
def test_c_reader_mm_matches_python(tmp_path, symmetric):
    """spllt_hip_read_mm (reference src/spllt_mod.F90:426-491 mm_double_read + :543-620
    coo_to_csc_double) against scipy's reader: a symmetric file, and a general one read as
    (A + A^T) / 2; with and without the values = 3 conditioning."""
    import scipy.io
    A = _sym(50, 21)
    if not symmetric:
        A = (A + sp.random(50, 50, density=0.05, random_state=np.random.RandomState(4), format="csc")).tocsc()
    path = tmp_path / "a.mtx"
    scipy.io.mmwrite(str(path), sp.coo_matrix(A), symmetry="symmetric" if symmetric else "general", precision=17)
    E = matgen.read_mtx(str(path))
    n, ptr, row, val = matgen.read_file_c(str(path), "mm", values=0)
    en, eptr, erow, eval_ = _lower_csc(E)
    assert n == en and np.array_equal(ptr, eptr) and np.array_equal(row, erow)
    np.testing.assert_allclose(val, eval_, rtol=1e-15, atol=1e-300)
    n, ptr, row, val = matgen.read_file_c(str(path), "mm", values=3)
    off = E - sp.diags(E.diagonal())
    D = (off + sp.diags(1.0 + np.asarray(abs(off).sum(axis=1)).ravel())).tocsc()
    en, eptr, erow, eval_ = _lower_csc(D)
    assert np.array_equal(ptr, eptr) and np.array_equal(row, erow)
    np.testing.assert_allclose(val, eval_, rtol=1e-14, atol=0)


def test_c_reader_feeds_the_c_api(tmp_path):
    """file -> spllt_hip_read_rb -> spllt_analyse, all at the C boundary (no GPU needed up to here)"""
    from spllt_amd import api
    A = _sym(80, 31)
    path = tmp_path / "a.rb"
    matgen.write_rb(str(path), A)
    n, ptr, row, val = matgen.read_file_c(str(path), "rb", values=3)
    f = api.Factorization(n, ptr.astype(np.int32), row.astype(np.int32), nb=16, nemin=4)
    assert f.sym_info()["n"] == 80 and f.sym_info()["nnz_a"] == val.size
    f.close()


@pytest.mark.parametrize("body,why", [
    ("%%MatrixMarket matrix coordinate real symmetric\n3 3 4000000000000\n1 1 2.0\n", "count beyond the 32-bit interface"),
    ("%%MatrixMarket matrix coordinate real symmetric\n3 3 2000000000\n1 1 2.0\n", "count the file is too short for"),
    ("%%MatrixMarket matrix coordinate real symmetric\n4000000000 4000000000 1\n1 1 2.0\n", "n beyond int"),
    ("%%MatrixMarket matrix coordinate real symmetric\n3 3 3\n1 1 2.0\n2 1 1.0\n1 2 1.0\n", "both triangles of a symmetric file"),
    ("%%MatrixMarket matrix coordinate real symmetric\n3 3 2\n1 1 2.0\n4 1 1.0\n", "row index out of range"),
    ("%%MatrixMarket matrix coordinate real symmetric\n3 3 2\n1 1 2.0\n", "fewer entries than promised"),
])
def test_c_reader_mm_malformed_files_set_a_flag(tmp_path, body, why):
    """Header counts are claims, not facts: a size line that promises more than the file (or the
    32-bit interface) holds, an index out of range, or a symmetric file that lists an entry in
    both triangles comes back as an error flag -- never as std::terminate from a reserve(), and
    never as a silently doubled entry."""
    path = tmp_path / "bad.mtx"
    path.write_text(body)
    with pytest.raises(ValueError, match="flag -1"):      # -10 (parameter) or -1 (allocation)
        matgen.read_file_c(str(path), "mm", values=0)


def test_c_reader_mm_upper_triangle_only_is_mirrored(tmp_path):
    """a symmetric file that stores the UPPER triangle (some writers do) is read as its mirror;
    repeated entries of the same triangle are summed (assembled finite-element files)"""
    path = tmp_path / "upper.mtx"
    path.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 5\n1 1 2.0\n1 2 1.0\n1 2 0.5\n2 2 3.0\n3 3 4.0\n")
    n, ptr, row, val = matgen.read_file_c(str(path), "mm", values=0)
    assert n == 3 and list(ptr) == [1, 3, 4, 5] and list(row) == [1, 2, 2, 3]
    np.testing.assert_array_equal(val, [2.0, 1.5, 3.0, 4.0])


def test_c_reader_rb_header_counts_are_checked(tmp_path):
    """a Rutherford-Boeing header whose card / entry counts exceed the file is an error flag"""
    path = tmp_path / "bad.rb"
    with open(path, "w") as fh:
        fh.write(f"{'bad header':<72}{'KEY':<8}\n")
        fh.write(f"{3:14d}{1:14d}{1:14d}{1:14d}\n")
        fh.write(f"{'rsa':<14}{3:14d}{3:14d}{2000000000:14d}{0:14d}\n")
        fh.write(f"{'(4I2)':<16}{'(5I2)':<16}{'(5E10.3)':<20}\n")
        fh.write(" 1 3 5 6\n 1 2 2 3 3\n 2.000E+00-1.000E+00 2.000E+00-1.000E+00 2.000E+00\n")
    with pytest.raises(ValueError, match="flag -10"):
        matgen.read_file_c(str(path), "rb", values=0)
    with open(path, "w") as fh:
        fh.write(f"{'bad header':<72}{'KEY':<8}\n")
        fh.write(f"{3000000:14d}{1000000:14d}{1000000:14d}{1000000:14d}\n")
        fh.write(f"{'rsa':<14}{3:14d}{3:14d}{5:14d}{0:14d}\n")
        fh.write(f"{'(4I2)':<16}{'(5I2)':<16}{'(5E10.3)':<20}\n")
        fh.write(" 1 3 5 6\n 1 2 2 3 3\n 2.000E+00-1.000E+00 2.000E+00-1.000E+00 2.000E+00\n")
    with pytest.raises(ValueError, match="flag -10"):
        matgen.read_file_c(str(path), "rb", values=0)
