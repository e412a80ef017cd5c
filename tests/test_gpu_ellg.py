"""GPU parity of the explicit-column sliced-ELL exact SpMV (csrc/its_ellg.hip, round 3) through the C ABI.

Any matrix whose longest row has at most 8 entries gets the coalesced two-rows-per-lane walk of the ELL kernels, whatever
its columns: a base column per (slice of 128 rows, slot) plus deltas of 8, 16 or 32 bits, the width chosen per slice, the
rows of a slice aligned to its longest row.  The kernel still runs the reference's in-order fused chain per row
(kernels.hpp:380-409), so every epilogue must be BIT-identical to the oracle and to the CRS stream kernels.
Covers: matrices with more than 255 distinct col - row (no dictionary exists), unsorted columns, all three delta widths in
one matrix, rows of 0..8 entries, empty slices, sizes around the slice / row-block sizes, the stencil matrices with the
dictionary forms switched off ("spmv_ell_dict" 0), the period work map on and off, every epilogue of the solver trips."""
import numpy as np
import pytest

import oracle_py as orc
from iterative_solvers_amd import capi, matgen
from test_gpu_ell import CASES as ELL_CASES, _with_diagonal

pytestmark = pytest.mark.gpu


def scattered(n, w, seed, spread, sort=True, empty_every=0, full=0.9):
    """rows of <= w entries, columns row + uniform(-spread, spread) (clipped, distinct); spread >= n: anywhere"""
    rng = np.random.default_rng(seed)
    rp = [0]; col = []; val = []
    for i in range(n):
        if not (empty_every and i % empty_every == 3):
            k = w if rng.random() < full else int(rng.integers(0, w + 1))
            lo, hi = max(0, i - spread), min(n, i + spread + 1)
            c = rng.choice(np.arange(lo, hi), size=min(k, hi - lo), replace=False)
            if sort:
                c = np.sort(c)
            col.extend(int(v) for v in c); val.extend(rng.uniform(-1, 1, len(c)))
        rp.append(len(col))
    return np.array(rp, np.int32), np.array(col, np.int32), np.array(val)


def three_widths(seed):
    """first third near the diagonal (8-bit deltas), second third within +-20 000 (16-bit), last third anywhere (32-bit)"""
    n = 3 * 70000
    parts = [scattered(70000, 7, seed, 40), scattered(70000, 7, seed + 1, 20000), scattered(70000, 7, seed + 2, 10**9)]
    rp = [0]; col = []; val = []
    for p, (r, c, v) in enumerate(parts):
        # part p's columns live in its own third for parts 0 and 1, anywhere in the matrix for part 2
        if p < 2:
            c = c + p * 70000
        else:
            c = np.random.default_rng(seed + 9).integers(0, n, len(c)).astype(np.int32)
            for i in range(70000):      # distinct columns inside a row (CRS as the reference builds it has no duplicates)
                seg = c[r[i]:r[i + 1]]
                while len(np.unique(seg)) < len(seg):
                    seg[:] = np.random.default_rng(seed + i).integers(0, n, len(seg))
        col.append(c); val.append(v)
        rp.extend((r[1:] + rp[-1]).tolist())
    return np.array(rp, np.int32), np.concatenate(col).astype(np.int32), np.concatenate(val)


def permuted_rows(rp, col, val, seed):
    """P A: row i of the result is row perm[i] of A (columns unchanged)"""
    n = len(rp) - 1
    perm = np.random.default_rng(seed).permutation(n)
    ln = np.diff(rp)[perm]
    rq = np.concatenate([[0], np.cumsum(ln)]).astype(np.int32)
    idx = np.concatenate([np.arange(rp[p], rp[p + 1]) for p in perm]) if n else np.zeros(0, int)
    return rq, col[idx], val[idx]


GEN = {
    "rand_w8": lambda: scattered(5001, 8, 31, 10**9),
    "rand_w8_unsorted": lambda: scattered(5001, 8, 32, 10**9, sort=False),
    "rand_w7": lambda: scattered(5001, 7, 30, 10**9),
    "rand_w3": lambda: scattered(1500, 3, 33, 10**9),
    "near_w7": lambda: scattered(20000, 7, 34, 60),
    "near_w7_unsorted": lambda: scattered(20000, 7, 35, 60, sort=False),
    "mid_w5": lambda: scattered(60000, 5, 36, 9000),
    "holes": lambda: scattered(9000, 6, 37, 300, empty_every=5),
    "n127": lambda: scattered(127, 4, 38, 10**9),
    "n129": lambda: scattered(129, 4, 39, 10**9),
    "n513": lambda: scattered(513, 8, 40, 10**9),
    "three_widths": lambda: three_widths(41),
    "lap3d_24_rows_permuted": lambda: permuted_rows(*matgen.laplace3d_7pt(24), 42),
}
# the stencil-class matrices of test_gpu_ell.py with rows of <= 8 entries, dictionary forms off
DICT_OFF = ["n1", "n2", "n127", "n128", "n511", "n513", "n5001_w8", "far_band", "lap2d_64", "lap3d_20", "lap3d_64", "wide_w8_20off"]


def _check_store(h, rp, col, val, tag):
    n = len(rp) - 1
    x = np.random.default_rng(11).uniform(-1, 1, n)
    ref = orc.spmv(rp, col, val, x)
    for ell, mp in ((1, 1), (1, 0), (0, 1)):
        h.set_option("spmv_ell", ell); h.set_option("spmv_ell_map", mp)
        name = h.spmv_kernel_info(capi.MAT_A, False)[0]
        assert (name == "spmv_ellg_kernel") == bool(ell), (tag, ell, name)
        assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref), (tag, ell, mp)


@pytest.fixture
def dict_off(gpu_handle):
    """small matrices have few distinct col - row whatever their columns: keep the dictionary forms out of the way"""
    gpu_handle.set_option("spmv_ell_dict", 0)
    yield gpu_handle
    gpu_handle.set_option("spmv_ell_dict", 1)


def test_ellg_is_what_a_matrix_without_a_dictionary_gets(gpu_handle):
    h = gpu_handle
    rp, col, val = GEN["rand_w8"]()                          # 5001 rows, ~40 000 distinct col - row
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert h.ell_info()["kind"] == 3 and h.spmv_kernel_info(capi.MAT_A, True)[0] == "spmv_ellg_kernel"
    rp, col, val = scattered(5001, 9, 44, 10**9)             # one entry too long: CRS kernels
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert h.ell_info()["kind"] == 0 and not h.spmv_kernel_info(capi.MAT_A, True)[0].startswith("spmv_ell")
    rp, col, val = scattered(5001, 8, 45, 10**9, full=0.3)   # too much padding: CRS kernels
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert h.ell_info()["kind"] == 0


@pytest.mark.parametrize("name", sorted(GEN))
def test_ellg_store_bit_exact(dict_off, name):
    h = dict_off
    rp, col, val = GEN[name]()
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    info = h.ell_info()
    assert info["kind"] == 3 and info["width"] == int(np.diff(rp).max()), info
    hist = info["slices_by_delta_width"]
    assert sum(hist) == -(-(len(rp) - 1) // 512) * 4
    if name == "three_widths":
        assert min(hist) > 100, hist                       # all three widths in one matrix
    if name.startswith("near"):
        assert hist[1] == 0 and hist[2] == 0, hist         # a slot spans at most 127 rows + 2 x 60 columns: 8-bit deltas
    if name == "mid_w5":
        assert hist[1] >= 465 and hist[2] == 0, hist       # +-9000 columns: 16-bit deltas (the empty padding slices: 8-bit)
    if name.startswith("rand_w8"):
        assert hist[0] == 0 and hist[2] == 0, hist         # 5001 columns: any two are less than 65 535 apart
    try:
        _check_store(h, rp, col, val, name)
    finally:
        h.set_option("spmv_ell", 1); h.set_option("spmv_ell_map", 1)


@pytest.mark.parametrize("name", DICT_OFF)
def test_ellg_on_stencil_class_matrices_with_the_dictionary_forms_off(gpu_handle, name):
    """"spmv_ell_dict" 0 at upload: the matrices that would get the nibble / byte dictionary image get the explicit-column
    one.  On grids the alignment to the slice's longest row keeps boundary rows in their neighbours' slots: 8-bit deltas."""
    h = gpu_handle
    rp, col, val = ELL_CASES[name]()
    n = len(rp) - 1
    h.set_option("spmv_ell_dict", 0)
    try:
        h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
        info = h.ell_info()
        assert info["kind"] == 3, info
        hist = info["slices_by_delta_width"]
        if name in ("lap2d_64", "lap3d_20", "lap3d_64"):
            assert hist[1] == 0 and hist[2] == 0, (name, hist)
            w = info["width"]
            assert info["index_words"] == -(-n // 512) * 512 * ((w + 3) // 4)
        _check_store(h, rp, col, val, name)
    finally:
        h.set_option("spmv_ell_dict", 1); h.set_option("spmv_ell", 1); h.set_option("spmv_ell_map", 1)


@pytest.mark.parametrize("name", ["rand_w7", "rand_w8_unsorted", "near_w7_unsorted", "mid_w5", "holes", "three_widths", "lap3d_24_rows_permuted"])
def test_ellg_every_epilogue_matches_the_crs_kernels(dict_off, name):
    """its_spmv_timed modes 0..7 (plain, CG form, residual stored / norm only, Jacobi sweep, GS right-hand side and the two
    fused forms, b := x): the stored vectors agree bit for bit between spmv_ellg_kernel and the CRS stream kernels."""
    h = dict_off
    rp, col, val = GEN[name]()
    n = len(rp) - 1
    if "permuted" not in name and "unsorted" not in name:
        rp, col, val = matgen.crs_from_coo(n, *_with_diagonal(rp, col, val))      # the Jacobi forms divide by the diagonal
    x = np.random.default_rng(12).uniform(0.5, 1.5, n)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    px = h.to_dev(x); py = h.dev_alloc(8 * n + 64)
    has_diag = bool(np.all(np.bincount(np.repeat(np.arange(n), np.diff(rp))[col == np.repeat(np.arange(n), np.diff(rp))], minlength=n) == 1))
    modes = (0, 1, 2, 4, 5, 6, 7) if has_diag else (0, 1, 2, 5, 7)
    try:
        outs = {}
        for ell in (0, 1):
            h.set_option("spmv_ell", ell)
            assert (h.spmv_kernel_info(capi.MAT_A, True)[0] == "spmv_ellg_kernel") == bool(ell)
            for mode in modes:
                h.check(h.L.its_memcpy_h2d(h.h, py, np.zeros(n).ctypes.data, 8 * n), "clear")
                h.spmv_timed(px, py, capi.FMT_CSR_EXACT, reps=1, with_dot=mode)
                outs[(ell, mode)] = h.from_dev(py, n)
        for mode in modes:
            assert np.array_equal(outs[(0, mode)], outs[(1, mode)], equal_nan=True), (name, mode)
        assert np.array_equal(outs[(1, 0)], orc.spmv(rp, col, val, x))
    finally:
        h.set_option("spmv_ell", 1)
        h.dev_free(px); h.dev_free(py)


def test_ellg_solvers_on_a_matrix_without_a_dictionary(gpu_handle):
    """CG, Jacobi and Gauss-Seidel trips on a symmetric diagonally dominant matrix with scattered columns (no offset
    dictionary): the products run on spmv_ellg_kernel and the iterates equal the oracle's to the bar of test_gpu_solvers."""
    from iterative_solvers_amd import solver as S
    rng = np.random.default_rng(50)
    n = 6000
    I = np.repeat(np.arange(n), 3); J = rng.integers(0, n, 3 * n)
    keep = I != J
    I, J = I[keep], J[keep]
    V = rng.uniform(-1, 0, len(I))
    # symmetric pattern, at most 7 off-diagonal entries per row, strictly diagonally dominant
    import scipy.sparse as sp
    M = sp.coo_matrix((V, (I, J)), shape=(n, n)).tocsr(); M = M + M.T
    M.sum_duplicates()
    rows = []
    for i in range(n):
        c = M.indices[M.indptr[i]:M.indptr[i + 1]][:7]; v = M.data[M.indptr[i]:M.indptr[i + 1]][:7]
        rows.append((c, v))
    # (truncation breaks symmetry; CG is not run to convergence here, only compared trip by trip)
    II = []; JJ = []; VV = []
    for i, (c, v) in enumerate(rows):
        II.extend([i] * len(c)); JJ.extend(c.tolist()); VV.extend(v.tolist())
        II.append(i); JJ.append(i); VV.append(1.0 + float(np.abs(v).sum()))
    rp, col, val = matgen.crs_from_coo(n, np.array(II), np.array(JJ), np.array(VV))
    assert int(np.diff(rp).max()) <= 8
    b = np.ones(n)
    orc.set_threads(1)
    for name, kind in (("jacobi", orc.JACOBI), ("gauss-seidel", orc.GAUSS_SEIDEL), ("conjugate-gradient", orc.CG)):
        kw = dict(jacobi_mode=1) if name == "jacobi" else {}
        ref = orc.solve(rp, col, val, b, b.copy(), kind, jacobi_mode=1 if name == "jacobi" else 0, fixed_iters=25, max_iters=25)
        args = S.Args(name, S.LoopParams(max_iters=25, tol=1e-30))
        sol = S.Solver(name, device=0, **kw)
        try:
            S.preprocessing_crs(args, sol, rp, col, val, b, b.copy())
            assert sol.handle.spmv_kernel_info(capi.MAT_A, name == "conjugate-gradient")[0] == "spmv_ellg_kernel"
            S.solve(args, sol)
            got = args.normed_residuals[:args.loop_params.residual_count]
            want = ref["history"][:len(got)]
            assert len(got) >= 20
            assert np.all(np.abs(got - want) <= 1e-10 * np.maximum(want, 1e-2 * want[0])), name
        finally:
            sol.handle.close()
