"""GPU parity: L1 kernels through the C ABI vs the oracle (CPU restatement of kernels.hpp).

Bar (BASELINE.json north_star): SpMV bit-exact for the row-ordered scalar-CSR variant, 1e-12
relative for the reordered variants (vector-CSR, SELL-C-sigma); elementwise kernels bit-exact;
reductions 1e-13 relative (summation order differs from the OpenMP reduction by design).
"""
import numpy as np
import pytest

import oracle_py as orc
from helpers import MATRICES, random_crs, rel_err
from iterative_solvers_amd import capi, matgen

pytestmark = pytest.mark.gpu


def _spmv_all(h, rp, col, val, x, n_cols=None):
    out = {}
    for fmt in (capi.FMT_CSR_EXACT, capi.FMT_CSR_VECTOR, capi.FMT_SELL):
        h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_SELL, n_cols=n_cols)
        out[fmt] = h.spmv_host(x, len(rp) - 1, capi.MAT_A, fmt)
    return out


@pytest.mark.parametrize("name", sorted(MATRICES))
def test_spmv_generated(gpu_handle, name):
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    x = np.random.default_rng(1).uniform(-1, 1, n)
    ref = orc.spmv(rp, col, val, x)
    got = _spmv_all(gpu_handle, rp, col, val, x)
    assert np.array_equal(got[capi.FMT_CSR_EXACT], ref), "scalar-CSR must be bit-exact"
    assert rel_err(got[capi.FMT_CSR_VECTOR], ref) <= 1e-12
    assert rel_err(got[capi.FMT_SELL], ref) <= 1e-12


@pytest.mark.parametrize("n,max_len,empty_every,long_rows", [
    (1, 3, 0, ()), (7, 0, 0, ()), (63, 5, 2, ()), (257, 9, 3, ()), (1000, 40, 7, ()),
    (3000, 12, 0, ((5, 1792), (6, 1793), (100, 9000), (2999, 4000))),   # rows at / past the LDS chunk size
    (70000, 6, 5, ((69999, 300),)),
    # either side of SPMV_SOLO_LEN (256) and of the row block's capacity (1536), neighbours, first and last row
    (5000, 10, 4, ((0, 257), (1, 256), (2, 1000), (3, 1536), (4, 1537), (2500, 600), (2501, 700), (4999, 258))),
    # rows at / around the register walk's threshold (2048) and its 1024-entry chunks
    (4100, 8, 0, ((7, 2048), (8, 2049), (9, 3072), (10, 4095), (11, 2047), (12, 3073), (4099, 4096))),
])
def test_spmv_ragged(gpu_handle, n, max_len, empty_every, long_rows):
    rp, col, val = random_crs(n, max_len, seed=n, empty_every=empty_every, long_rows=long_rows)
    x = np.random.default_rng(2).uniform(-1, 1, n)
    ref = orc.spmv(rp, col, val, x)
    got = _spmv_all(gpu_handle, rp, col, val, x)
    assert np.array_equal(got[capi.FMT_CSR_EXACT], ref)
    scale = np.abs(val).max() * np.abs(x).max() if len(val) else 1.0
    for fmt in (capi.FMT_CSR_VECTOR, capi.FMT_SELL):
        # per-row bound: |err| <= 1e-12 * sum|a_ij x_j|  (rows with cancellation have tiny results)
        absrow = orc.spmv(rp, col, np.abs(val), np.abs(x))
        assert np.all(np.abs(got[fmt] - ref) <= 1e-12 * np.maximum(absrow, scale * 1e-3))


def test_spmv_both_exact_kernels(gpu_handle):
    """Regular (v4 stream) and irregular (two-phase) exact kernels on the same ragged matrix: both bit-exact."""
    rp, col, val = random_crs(3000, 12, seed=11, long_rows=((5, 1792), (100, 9000), (2999, 40)))
    n = len(rp) - 1
    x = np.random.default_rng(3).uniform(-1, 1, n)
    ref = orc.spmv(rp, col, val, x)
    try:
        for irr in (0, 1):
            gpu_handle.set_option("spmv_irr", irr)
            gpu_handle.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
            assert np.array_equal(gpu_handle.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref), irr
    finally:
        gpu_handle.set_option("spmv_irr", 1)


@pytest.mark.parametrize("name", ["lap2d_64", "lap3d_16", "cd27_12"])
def test_spmv_offset_dictionary_stream(gpu_handle, name):
    """Stencil-class matrices (<= 256 distinct col - row) stream a 1-byte offset index instead of the
    4-byte column: same arithmetic, so bit-exact with and without it."""
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    x = np.random.default_rng(4).uniform(-1, 1, n)
    ref = orc.spmv(rp, col, val, x)
    gpu_handle.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    try:
        for idx, vd in ((0, 0), (1, 0), (1, 1)):
            gpu_handle.set_option("spmv_idx", idx)
            gpu_handle.set_option("spmv_valdict", vd)       # opt-in 1-byte value dictionary (<= 256 distinct values)
            assert np.array_equal(gpu_handle.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref), (idx, vd)
    finally:
        gpu_handle.set_option("spmv_idx", 1)
        gpu_handle.set_option("spmv_valdict", 0)


def test_spmv_known_answers(gpu_handle):
    """M1*V1 = {74,134,194}, Mn*V1 = {-58,-22,-76.2} (reference test_suite/test_data.cpp:40-49,4-13,467-494)."""
    import json, os
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "test_data_tables.json")))
    for case in d["spmv"]:
        n = case["n"]
        rp, col, val = matgen.crs_from_coo(n, case["I"], case["J"], case["V"])
        for fmt in (capi.FMT_CSR_EXACT, capi.FMT_CSR_VECTOR, capi.FMT_SELL):
            gpu_handle.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_SELL)
            y = gpu_handle.spmv_host(np.array(case["x"], dtype=np.float64), n, capi.MAT_A, fmt)
            assert np.allclose(y, case["y"], rtol=1e-13, atol=1e-13), (case["name"], fmt)


def test_spmv_rejects_bad_input(gpu_handle):
    rp = np.array([0, 1, 2], np.int32)
    with pytest.raises(capi.ItsError):
        gpu_handle.upload_crs(rp, np.array([0, 5], np.int32), np.ones(2))      # column out of range
    with pytest.raises(capi.ItsError):
        gpu_handle.upload_crs(np.array([0, 2, 1], np.int32), np.array([0, 1], np.int32), np.ones(2))


def test_sweep_rejects_a_lower_triangle_with_columns_beyond_its_rows():
    """crs_L with n_cols > n_rows and an entry in [n_rows, n_cols) is only legal in a row-partitioned run (halo columns of
    earlier colours): on one GPU the sweep would read past the iterate, so the schedule refuses it."""
    h = capi.Handle(0)
    try:
        n = 6
        rp = np.array([0, 0, 1, 2, 3, 4, 5], np.int32)
        h.upload_crs(np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.full(n, 2.0), capi.MAT_A)
        h.upload_vectors(np.full(n, 2.0), np.ones(n), np.ones(n))
        xd = h.to_dev(np.zeros(n)); rd = h.to_dev(np.ones(n))
        h.upload_crs(rp, np.array([0, 1, 2, 3, 4], np.int32), np.ones(5), capi.MAT_L, n_cols=n + 2)     # fine: strictly lower
        h.spltsv(xd, rd)
        h.upload_crs(rp, np.array([0, 1, n + 1, 3, 4], np.int32), np.ones(5), capi.MAT_L, n_cols=n + 2)  # row 3 reads column 7
        with pytest.raises(capi.ItsError, match="strictly lower"):
            h.spltsv(xd, rd)
    finally:
        h.close()


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 4097, 100001, 1 << 20])
def test_blas1(gpu_handle, n):
    h = gpu_handle
    rng = np.random.default_rng(n)
    a = rng.uniform(-1, 1, n); b = rng.uniform(-1, 1, n)
    pa, pb = h.to_dev(a), h.to_dev(b)
    po = h.dev_alloc(8 * n)
    try:
        s = 0.37251
        h.axpby(po, pa, pb, n, s)
        assert np.array_equal(h.from_dev(po, n), orc.sum_vectors(a, b, s))          # kernels.hpp:54-71
        h.axpby(po, pa, pb, n, -s)
        assert np.array_equal(h.from_dev(po, n), orc.subtract_vectors(a, b, s))     # kernels.hpp:73-87
        h.scale(po, pa, s, n)
        assert np.array_equal(h.from_dev(po, n), orc.scale(a, s))                   # kernels.hpp:136-168
        # reductions: compare against the oracle and an exact (math.fsum) value
        import math
        exact = math.fsum(a * b)
        bound = 1e-13 * math.fsum(np.abs(a * b)) + 1e-300
        assert abs(h.dot(pa, pb, n) - exact) <= bound
        assert abs(orc.dot(a, b) - exact) <= bound * 10
        assert abs(h.nrm2(pa, n) - math.sqrt(math.fsum(a * a))) <= 1e-13 * math.sqrt(math.fsum(a * a))
        assert h.norm_inf(pa, n) == orc.norm_inf(a)                                  # kernels.hpp:558-575
        # run-to-run reproducibility of the fixed-order reduction
        assert h.dot(pa, pb, n) == h.dot(pa, pb, n)
    finally:
        for p in (pa, pb, po):
            h.dev_free(p)


def test_blas1_unaligned_views(gpu_handle):
    """GMRES addresses rows V[k*n] of odd length: operands 8-byte but not 16-byte aligned."""
    h = gpu_handle
    n = 1001
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, n + 1); b = rng.uniform(-1, 1, n + 1)
    pa, pb = h.to_dev(a), h.to_dev(b)
    po = h.dev_alloc(8 * (n + 1))
    try:
        h.axpby(po + 8, pa + 8, pb + 8, n, 0.5)
        assert np.array_equal(h.from_dev(po + 8, n), orc.sum_vectors(a[1:], b[1:], 0.5))
        import math
        assert abs(h.dot(pa + 8, pb + 8, n) - math.fsum(a[1:] * b[1:])) <= 1e-13 * math.fsum(np.abs(a[1:] * b[1:]))
    finally:
        for p in (pa, pb, po):
            h.dev_free(p)


@pytest.mark.parametrize("name", ["lap2d_16", "lap3d_8", "cd27_6", "pl_200", "pl_5000"])
def test_spltsv_levels_bit_exact(name):
    """Level-scheduled forward substitution == the reference's serial sweep (kernels.hpp:451-473)."""
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    (Lp, Lc, Lv), _ = orc.split_LU(rp, col, val)
    D = orc.extract_diag(rp, col, val)
    rhs = np.random.default_rng(3).uniform(-1, 1, n)
    ref = orc.spltsv(Lp, Lc, Lv, D, rhs)
    h = capi.Handle(0)
    try:
        h.upload_crs(rp, col, val, capi.MAT_A)
        h.upload_crs(Lp, Lc, Lv, capi.MAT_L)
        h.upload_vectors(D, rhs, np.zeros(n))
        px = h.to_dev(np.zeros(n)); pr = h.to_dev(rhs)
        h.spltsv(px, pr)
        assert np.array_equal(h.from_dev(px, n), ref)
        h.dev_free(px); h.dev_free(pr)
    finally:
        h.close()
