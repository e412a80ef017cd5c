"""GPU parity of the column-panel blocked exact SpMV (csrc/its_panel.hip) through the C ABI.

The panel image stores the entries of the rows of <= 128 entries sorted by (row tile, column panel, row, position in the
row); a row's chain still runs in CRS order because its columns ascend, so every value must be BIT-identical to the oracle
and to the irregular-row kernel it replaces ("spmv_panel" 0).  The image is meant for x vectors far beyond an XCD's L2;
the tests force it onto small matrices ("spmv_panel_min_cols" 1) with narrow panels so that rows cross many panels,
chunks fill up and overflow, tiles are ragged, and the long rows (left to the irregular-row kernel) add their partial sums
to the panel kernel's."""
import numpy as np
import pytest

import oracle_py as orc
from iterative_solvers_amd import capi, matgen, solver as S

pytestmark = pytest.mark.gpu


@pytest.fixture()
def forced(gpu_handle):
    h = gpu_handle
    h.set_option("spmv_panel_min_cols", 1)
    yield h
    h.set_option("spmv_panel_min_cols", 1 << 21)
    h.set_option("spmv_panel_log2pw", 16)
    h.set_option("spmv_panel", 1)


def heavy_tail(n, seed, max_len=3000, n_cols=None, diag=False):
    """rows of 1 .. max_len entries (Pareto tail), ascending distinct columns, some rows empty (none with diag=True:
    every row then holds its diagonal)"""
    rng = np.random.default_rng(seed)
    n_cols = n_cols or n
    rows = []
    for i in range(n):
        ln = 0 if (i % 23 == 7 and not diag) else int(min(max_len, n_cols, 3.0 * rng.random() ** (-0.75)))
        c = rng.choice(n_cols, ln, replace=False)
        if diag:
            c = np.union1d(c, [i])
        rows.append(np.sort(c))
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows])
    return rp, np.concatenate(rows).astype(np.int32), rng.uniform(-1, 1, int(rp[-1])), n_cols


@pytest.mark.parametrize("n,log2pw,seed", [(1, 4, 1), (300, 4, 2), (4096, 6, 3), (4097, 8, 4), (9000, 5, 5), (20000, 10, 6), (20000, 19, 7)])
def test_panel_store_dot_residual_bit_exact(forced, n, log2pw, seed):
    h = forced
    rp, col, val, n_cols = heavy_tail(n, seed, n_cols=n if seed % 2 else n + 700)
    if np.diff(rp).max() <= 32:                                  # the image is for the irregular-row class only
        col = np.concatenate([col, np.arange(40, dtype=np.int32) % n_cols]); col[rp[-1]:] = np.sort(col[rp[-1]:])
        val = np.concatenate([val, np.ones(40)]); rp = np.concatenate([rp, [rp[-1] + 40]]).astype(np.int32)
    n = len(rp) - 1
    h.set_option("spmv_panel_log2pw", log2pw)
    x = np.random.default_rng(seed).uniform(-1, 1, n_cols)
    ref = orc.spmv(rp, col, val, x)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT, n_cols=n_cols)
    assert h.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel"
    assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref)
    if n_cols != n:
        return
    # the forms with partial sums: CG form (y and y . x), residual stored, residual norm only
    px = h.to_dev(x); py = h.dev_alloc(8 * n + 64)
    outs = {}
    try:
        for pan in (1, 0):
            h.set_option("spmv_panel", pan)
            assert (h.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel") == bool(pan)
            for mode in (0, 1, 2):
                h.check(h.L.its_memcpy_h2d(h.h, py, np.zeros(n).ctypes.data, 8 * n), "clear")
                h.spmv_timed(px, py, capi.FMT_CSR_EXACT, reps=1, with_dot=mode)
                outs[(pan, mode)] = h.from_dev(py, n)
        for mode in (0, 1, 2):
            assert np.array_equal(outs[(1, mode)], outs[(0, mode)]), mode
        assert np.array_equal(outs[(1, 2)], x - ref)
    finally:
        h.set_option("spmv_panel", 1)
        h.dev_free(px); h.dev_free(py)


def test_panel_needs_ascending_columns_and_the_irregular_class(forced):
    h = forced
    rp, col, val, _ = heavy_tail(5000, 11)
    n = len(rp) - 1
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert h.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel"
    # one row with two columns swapped: the panel order would no longer be the CRS order
    i = int(np.argmax(np.diff(rp) >= 2))
    c2 = col.copy(); c2[rp[i]], c2[rp[i] + 1] = col[rp[i] + 1], col[rp[i]]
    h.upload_crs(rp, c2, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert h.spmv_kernel_info(capi.MAT_A)[0] == "spmv_irr_kernel"
    x = np.random.default_rng(1).uniform(-1, 1, n)
    assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), orc.spmv(rp, c2, val, x))
    # short regular rows never take it
    rp3, col3, val3 = matgen.laplace3d_7pt(12)
    h.upload_crs(rp3, col3, val3, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert h.spmv_kernel_info(capi.MAT_A)[0] == "spmv_ell_kernel"


def test_panel_off_by_default_for_small_x(gpu_handle):
    rp, col, val, _ = heavy_tail(5000, 12)
    gpu_handle.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert gpu_handle.spmv_kernel_info(capi.MAT_A)[0] == "spmv_irr_kernel"


def test_cg_history_identical_with_and_without_the_panel_image(forced):
    """CG on an irregular SPD matrix (A = B + B^T + dominant diagonal): SpMV + fused p.Ap through the panel kernel and the
    long-row launch that adds its partial sums.  The partials are grouped differently (other grid), so the histories agree
    to rounding, not bitwise; the iterate after the same number of trips to 1e-12."""
    h = forced
    n = 6000
    rng = np.random.default_rng(21)
    rp0, col0, val0, _ = heavy_tail(n, 21, max_len=800)
    rows0 = np.repeat(np.arange(n), np.diff(rp0))
    I = np.concatenate([rows0, col0]); J = np.concatenate([col0, rows0]); V = np.concatenate([val0, val0]) * 0.01
    keep = I != J
    I, J, V = I[keep], J[keep], V[keep]
    lin = I.astype(np.int64) * n + J
    order = np.argsort(lin, kind="stable"); lin = lin[order]; V = V[order]
    first = np.ones(len(lin), bool); first[1:] = lin[1:] != lin[:-1]
    lin, V = lin[first], V[first]
    I = (lin // n).astype(np.int64); J = (lin % n).astype(np.int64)
    dg = 1.0 + np.bincount(I, weights=np.abs(V), minlength=n)
    I = np.concatenate([I, np.arange(n)]); J = np.concatenate([J, np.arange(n)]); V = np.concatenate([V, dg])
    order = np.lexsort((J, I))
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum(np.bincount(I, minlength=n))
    col = J[order].astype(np.int32); val = V[order]
    assert np.diff(rp).max() > 128
    b = np.ones(n)
    out = []
    for pan in (1, 0):
        args = S.Args("conjugate-gradient", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("conjugate-gradient", device=0)
        try:
            sol.handle.set_option("spmv_panel_min_cols", 1); sol.handle.set_option("spmv_panel_log2pw", 7)
            sol.handle.set_option("spmv_panel", pan)
            S.preprocessing_crs(args, sol, rp, col, val, b, np.zeros(n))
            assert (sol.handle.spmv_kernel_info(capi.MAT_A, True)[0] == "spmv_panel_kernel") == bool(pan)
            hist = np.array([sol.handle.iterate(it) for it in range(12)])
            out.append((hist, sol.save_x_star(12).copy()))
        finally:
            sol.handle.set_option("spmv_panel", 1)
            sol.handle.set_option("spmv_panel_min_cols", 1 << 21); sol.handle.set_option("spmv_panel_log2pw", 16)
            sol.handle.close()
    assert np.all(np.abs(out[0][0] - out[1][0]) <= 1e-12 * out[1][0][0])
    assert np.max(np.abs(out[0][1] - out[1][1])) <= 1e-12 * np.max(np.abs(out[1][1]))
    orc.set_threads(1)
    ref = orc.solve(rp, col, val, b, np.zeros(n), orc.CG, fixed_iters=12, max_iters=12)["all_values"][:12]
    assert np.all(np.abs(out[0][0] - ref) <= 1e-10 * ref)


@pytest.mark.parametrize("n,log2pw,seed", [(4097, 6, 31), (20000, 9, 32)])
def test_panel_jacobi_forms_bit_exact(forced, n, log2pw, seed):
    """EPI_JACOBI_REF (one chain without the diagonal) and EPI_JACOBI_FUSED (that chain and the full one: two running sums
    per row) store the same bits as the irregular-row kernel: its_spmv_timed modes 4 and 6 with b := x."""
    h = forced
    rp, col, val, _ = heavy_tail(n, seed, diag=True)
    rows = np.repeat(np.arange(n), np.diff(rp))
    val = np.where(col == rows, 5.0 + np.abs(val), val)
    h.set_option("spmv_panel_log2pw", log2pw)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    x = np.random.default_rng(seed).uniform(0.5, 1.5, n)
    px = h.to_dev(x); py = h.dev_alloc(8 * n + 64)
    outs = {}
    try:
        for pan in (1, 0):
            h.set_option("spmv_panel", pan)
            for mode in (4, 6):
                h.check(h.L.its_memcpy_h2d(h.h, py, np.zeros(n).ctypes.data, 8 * n), "clear")
                h.spmv_timed(px, py, capi.FMT_CSR_EXACT, reps=1, with_dot=mode)
                outs[(pan, mode)] = h.from_dev(py, n)
        for mode in (4, 6):
            assert np.array_equal(outs[(1, mode)], outs[(0, mode)]), mode
        # the sweep itself: x_i' = (b_i - sum_{j != i} a_ij x_j) / a_ii, chain in CRS order without the diagonal
        off = np.where(col == rows, 0.0, val)
        assert np.array_equal(outs[(1, 4)], (x - orc.spmv(rp, col, off, x)) / val[col == rows])
    finally:
        h.set_option("spmv_panel", 1)
        h.dev_free(px); h.dev_free(py)


def test_jacobi_solve_identical_iterate_with_the_panel_image(forced):
    """Jacobi (correct sweep, one fused pass over A per trip) with and without the panel image: the iterate is a per-row
    chain, so x after 15 trips is bit-identical; the recorded norms differ only by the grouping of the partial sums."""
    n = 12000
    rp, col, val, _ = heavy_tail(n, 41, max_len=1500, diag=True)
    rows = np.repeat(np.arange(n), np.diff(rp))
    val = 0.05 * val
    dg = 1.0 + np.bincount(rows, weights=np.abs(val), minlength=n)
    val = np.where(col == rows, dg[rows], val)
    b = np.ones(n)
    out = []
    for pan in (1, 0):
        args = S.Args("jacobi", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("jacobi", device=0, jacobi_mode=1)
        try:
            sol.handle.set_option("spmv_panel_min_cols", 1); sol.handle.set_option("spmv_panel_log2pw", 8)
            sol.handle.set_option("spmv_panel", pan)
            S.preprocessing_crs(args, sol, rp, col, val, b, np.zeros(n))
            hist = np.array([sol.handle.iterate(it) for it in range(15)])
            out.append((hist, sol.save_x_star(15).copy()))
        finally:
            sol.handle.set_option("spmv_panel", 1)
            sol.handle.set_option("spmv_panel_min_cols", 1 << 21); sol.handle.set_option("spmv_panel_log2pw", 16)
            sol.handle.close()
    assert np.array_equal(out[0][1], out[1][1])
    assert np.all(np.abs(out[0][0] - out[1][0]) <= 1e-13 * out[1][0][0])
    orc.set_threads(1)
    ref = orc.solve(rp, col, val, b, np.zeros(n), orc.JACOBI, jacobi_mode=1, fixed_iters=15, max_iters=15)
    assert np.array_equal(out[0][1], ref["x_star"])
    assert np.all(np.abs(out[0][0] - ref["all_values"][:15]) <= 1e-10 * ref["all_values"][:15])


def test_gauss_seidel_multicolor_identical_iterate_with_the_panel_image(forced):
    """Multicolour GS on an irregular matrix: the b - U x pass (image of crs_U) and the residual pass (image of A) go through
    the panel kernels, the colour sweeps do not; the iterate after 10 trips is bit-identical to the run without the images
    and to the oracle's natural-order sweep of the permuted matrix."""
    n = 12000
    rp, col, val, _ = heavy_tail(n, 51, max_len=1500, diag=True)
    rows = np.repeat(np.arange(n), np.diff(rp))
    val = 0.05 * val
    dg = 1.0 + np.bincount(rows, weights=np.abs(val), minlength=n)
    val = np.where(col == rows, dg[rows], val)
    b = np.ones(n)
    out = []
    for pan in (1, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_MULTICOLOR)
        try:
            sol.handle.set_option("spmv_panel_min_cols", 1); sol.handle.set_option("spmv_panel_log2pw", 8)
            sol.handle.set_option("spmv_panel", pan)
            S.preprocessing_crs(args, sol, rp, col, val, b, np.zeros(n))
            if pan:
                assert sol.handle.spmv_kernel_info(capi.MAT_U)[0] in ("spmv_panel_kernel", "spmv_irr_kernel")
            hist = np.array([sol.handle.iterate(it) for it in range(10)])
            out.append((hist, sol.save_x_star(10).copy(), sol.perm.copy()))
        finally:
            sol.handle.set_option("spmv_panel", 1)
            sol.handle.set_option("spmv_panel_min_cols", 1 << 21); sol.handle.set_option("spmv_panel_log2pw", 16)
            sol.handle.close()
    assert np.array_equal(out[0][1], out[1][1])
    assert np.all(np.abs(out[0][0] - out[1][0]) <= 1e-13 * out[1][0][0])
    perm = out[0][2]
    rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
    orc.set_threads(1)
    ref = orc.solve(rpp, cp, vp, b, np.zeros(n), orc.GAUSS_SEIDEL, fixed_iters=10, max_iters=10)
    x_ref = np.empty(n); x_ref[perm] = ref["x_star"]
    assert np.array_equal(out[0][1], x_ref)


@pytest.mark.parametrize("seed", range(24))
def test_panel_fuzz(forced, seed):
    """random shapes, panel widths and row-length cuts; columns ascend but may repeat (the sort is stable: equal columns keep
    their CRS order); empty rows, rows longer than the cut, fewer rows than a tile, more columns than rows"""
    h = forced
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([3, 70, 600, 4095, 4096, 4100, 9000, 15000]))
    n_cols = n + int(rng.choice([0, 0, 17, 900]))
    lcut = int(rng.choice([1, 8, 40, 128, 300]))
    log2pw = int(rng.integers(4, 13))
    rows = []
    for i in range(n):
        u = rng.random()
        ln = 0 if u < 0.1 else int(min(2000, 2.0 * rng.random() ** (-0.9)))
        c = np.sort(rng.integers(0, n_cols, ln))                      # repeats allowed
        rows.append(c)
    if max(len(r) for r in rows) <= 32:
        rows[n // 2] = np.sort(rng.integers(0, n_cols, 40))
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    val = rng.uniform(-1, 1, int(rp[-1]))
    x = rng.uniform(-1, 1, n_cols)
    h.set_option("spmv_panel_log2pw", log2pw); h.set_option("spmv_panel_lcut", lcut)
    try:
        h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT, n_cols=n_cols)
        ln = np.diff(rp)
        if ln[ln > lcut].sum() * 2 <= rp[-1]:
            assert h.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel", (n, lcut, log2pw)
        got = h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT)
        assert np.array_equal(got, orc.spmv(rp, col, val, x)), (n, n_cols, lcut, log2pw)
    finally:
        h.set_option("spmv_panel_lcut", 128)
