"""GPU parity of the sliced-ELL exact SpMV (csrc/its_ell.hip) through the C ABI.

The ELL image stores the CRS entries of every row in CRS order and the kernel runs the reference's in-order fused
chain per row (kernels.hpp:380-409), so every epilogue must be BIT-identical to the oracle and to the CRS stream
kernels it replaces ("spmv_ell" 0).  Covers: row counts around the slice / row-block sizes (128 / 512 rows), rows
of 0..8 entries (nibble indices) and of 9..32 entries or up to 255 offsets (byte indices), the period work map (far band >= 8 row blocks) on and off, all epilogues the solver trips launch,
unaligned caller vectors (fall back to the CRS kernels) and matrices that are not eligible."""
import numpy as np
import pytest

import oracle_py as orc
from iterative_solvers_amd import capi, matgen

pytestmark = pytest.mark.gpu


def banded(n, offsets, seed, drop=0.2):
    """Rows of <= len(offsets) entries with col - row drawn from `offsets` (ascending), some entries dropped,
    some rows empty: the class of matrices the ELL image is built for."""
    rng = np.random.default_rng(seed)
    rp = [0]; col = []; val = []
    for i in range(n):
        if i % 17 != 5:
            for o in offsets:
                c = i + o
                if 0 <= c < n and rng.random() >= drop:
                    col.append(c); val.append(rng.uniform(-1, 1))
        rp.append(len(col))
    return np.array(rp, np.int32), np.array(col, np.int32), np.array(val)


CASES = {
    "n1": lambda: banded(1, [0], 1, drop=0.0),
    "n2": lambda: banded(2, [-1, 0, 1], 2),
    "n127": lambda: banded(127, [-3, -1, 0, 1, 3], 3),
    "n128": lambda: banded(128, [-3, -1, 0, 1, 3], 4),
    "n511": lambda: banded(511, [-9, -1, 0, 1, 9], 5),
    "n513": lambda: banded(513, [-9, -1, 0, 1, 9], 6),
    "n5001_w8": lambda: banded(5001, [-70, -7, -2, -1, 0, 1, 7, 70], 7, drop=0.05),
    "n5001_sparse": lambda: banded(5001, [-70, -7, -2, -1, 0, 1, 7, 70], 9, drop=0.5),      # too much padding: CRS kernels
    "far_band": lambda: banded(50000, [-5000, -70, -1, 0, 1, 70, 5000], 8, drop=0.03),      # period map: 5000 // 512 = 9 blocks
    "lap2d_64": lambda: matgen.laplace2d_5pt(64),
    "lap3d_20": lambda: matgen.laplace3d_7pt(20),
    "lap3d_64": lambda: matgen.laplace3d_7pt(64),                               # period map: 4096 // 512 = 8 blocks
    # the byte-index form (spmv_ellw_kernel): rows of 9..32 entries, <= 255 distinct offsets
    "wide_w9": lambda: banded(3000, [-40, -9, -3, -1, 0, 1, 3, 9, 40], 21, drop=0.02),
    "wide_w17": lambda: banded(2049, list(range(-8, 9)), 22, drop=0.05),
    "wide_w32": lambda: banded(4097, list(range(-16, 16)), 23, drop=0.03),
    "wide_w33": lambda: banded(4097, list(range(-16, 17)), 24, drop=0.0),       # one entry too long: CRS kernels
    "wide_w8_20off": lambda: _shifting(6000, 25),                               # rows of 8, 20 offsets: too many for nibbles
    "cd27_20": lambda: matgen.convdiff3d_27pt(20),
    "cd27_48": lambda: matgen.convdiff3d_27pt(48),                              # period map: 2353 // 512 ... below 8: chunk map
    "cd27_72": lambda: matgen.convdiff3d_27pt(72),                              # far band 5257 -> 10 row blocks: period map
}


def _shifting(n, seed):
    """8 entries per row whose offsets depend on row % 5: more than 15 distinct offsets over the matrix"""
    rng = np.random.default_rng(seed)
    rp = [0]; col = []; val = []
    for i in range(n):
        for o in sorted({-60 - i % 5, -11 + i % 5, -2, 0, 1, 3 + i % 5, 20 - i % 5, 64 + i % 5}):
            if 0 <= i + o < n:
                col.append(i + o); val.append(rng.uniform(-1, 1))
        rp.append(len(col))
    return np.array(rp, np.int32), np.array(col, np.int32), np.array(val)


def _eligible(rp, col):
    """the upload's rule: "spmv_ell_kernel" for rows <= 8 with <= 15 distinct offsets, "spmv_ellw_kernel" for rows <= 32
    with <= 255, both only when the stored slots (row blocks of 512 rows x longest row) are within 25 % of nnz"""
    n = len(rp) - 1
    w = int(np.diff(rp).max()) if n else 0
    nnz = int(rp[-1])
    if not (1 <= w <= 32 and -(-n // 512) * 512 * w <= nnz + nnz // 4 + 8 * 512):
        return None
    n_off = len(np.unique(col - np.repeat(np.arange(n), np.diff(rp))))
    if w <= 8 and n_off <= 15:
        return "spmv_ell_kernel"
    return "spmv_ellw_kernel" if n_off <= 255 else None


def _kernel(h, with_dot=False):
    return h.spmv_kernel_info(capi.MAT_A, with_dot)[0]


@pytest.mark.parametrize("name", sorted(CASES))
def test_ell_store_bit_exact(gpu_handle, name):
    h = gpu_handle
    rp, col, val = CASES[name]()
    n = len(rp) - 1
    x = np.random.default_rng(11).uniform(-1, 1, n)
    ref = orc.spmv(rp, col, val, x)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    try:
        for ell, mp in ((1, 1), (1, 0), (0, 1)):
            h.set_option("spmv_ell", ell); h.set_option("spmv_ell_map", mp)
            want = _eligible(rp, col) if ell else None
            assert (_kernel(h) if want else want) == want and (want or not _kernel(h).startswith("spmv_ell")), (name, _kernel(h))
            assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref), (name, ell, mp)
    finally:
        h.set_option("spmv_ell", 1); h.set_option("spmv_ell_map", 1)


@pytest.mark.parametrize("name", ["n513", "n5001_w8", "far_band", "lap3d_20", "lap3d_64", "wide_w9", "wide_w17", "wide_w32",
                                  "wide_w8_20off", "cd27_20", "cd27_72"])
def test_ell_every_epilogue_matches_the_crs_kernels(gpu_handle, name):
    """its_spmv_timed modes 0..7 launch the epilogues of the solver trips (plain, CG form, residual stored / norm
    only, Jacobi sweep, GS right-hand side and the two fused forms) with b := x; the stored vectors must agree bit
    for bit between the ELL kernel and the CRS stream kernel."""
    h = gpu_handle
    rp, col, val = CASES[name]()
    n = len(rp) - 1
    # the Jacobi forms divide by the diagonal: make sure every row has one
    rp, col, val = matgen.crs_from_coo(n, *_with_diagonal(rp, col, val))
    x = np.random.default_rng(12).uniform(0.5, 1.5, n)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    px = h.to_dev(x); py = h.dev_alloc(8 * n + 64)
    try:
        outs = {}
        for ell in (0, 1):
            h.set_option("spmv_ell", ell)
            for mode in (0, 1, 2, 4, 5, 6, 7):
                h.check(h.L.its_memcpy_h2d(h.h, py, np.zeros(n).ctypes.data, 8 * n), "clear")
                h.spmv_timed(px, py, capi.FMT_CSR_EXACT, reps=1, with_dot=mode)
                outs[(ell, mode)] = h.from_dev(py, n)
        for mode in (0, 1, 2, 4, 5, 6, 7):
            assert np.array_equal(outs[(0, mode)], outs[(1, mode)]), (name, mode)
        assert np.array_equal(outs[(1, 0)], orc.spmv(rp, col, val, x))
    finally:
        h.set_option("spmv_ell", 1)
        h.dev_free(px); h.dev_free(py)


def _with_diagonal(rp, col, val):
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    keep = rows != col
    I = np.concatenate([rows[keep], np.arange(n)]); J = np.concatenate([col[keep], np.arange(n)])
    V = np.concatenate([val[keep], np.full(n, 4.0)])
    order = np.lexsort((J, I))
    return I[order], J[order], V[order]


def test_ell_unaligned_output_falls_back(gpu_handle):
    """Caller vectors that are only 8-byte aligned cannot take the 16-byte stores of the ELL kernel: the dispatch
    must fall back to the CRS kernels and still be exact."""
    h = gpu_handle
    rp, col, val = matgen.laplace3d_7pt(12)
    n = len(rp) - 1
    x = np.random.default_rng(13).uniform(-1, 1, n)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    px = h.to_dev(x); py = h.dev_alloc(8 * n + 64)
    try:
        h.spmv_dev(px, py + 8)
        assert np.array_equal(h.from_dev(py + 8, n), orc.spmv(rp, col, val, x))
    finally:
        h.dev_free(px); h.dev_free(py)


def test_ell_not_built_for_ineligible_matrices(gpu_handle):
    h = gpu_handle
    for mk in (lambda: matgen.convdiff3d_27pt(6), lambda: matgen.powerlaw(300)):
        rp, col, val = mk()
        h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
        assert not _kernel(h).startswith("spmv_ell")
    rp, col, val = matgen.laplace3d_7pt(8)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    name, streamed = h.spmv_kernel_info(capi.MAT_A, True)
    assert name == "spmv_ell_kernel" and streamed == 512 * (8 * 7 + 4) + 16 * 512
    rp, col, val = matgen.convdiff3d_27pt(16)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    name, streamed = h.spmv_kernel_info(capi.MAT_A, True)
    assert name == "spmv_ellw_kernel" and streamed == 4096 * (8 * 27 + 4 * 7) + 16 * 4096
    h.set_option("spmv_ell_wide", 0)
    try:
        assert not _kernel(h).startswith("spmv_ell")
    finally:
        h.set_option("spmv_ell_wide", 1)
