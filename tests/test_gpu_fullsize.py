"""GPU parity at BASELINE.json's full sizes (configs[1]: 3-D 7-pt 256^3; configs[2]: 27-pt 128^3; a 10^6-row
cut of configs[3]'s power-law generator).  The oracle is affordable for one SpMV and a few dozen CG trips at
these sizes, so the checks are direct (bit-exact SpMV, 1e-10 histories) plus size-independent properties:
linearity of the exact kernel in x and agreement of its three lossless stream encodings."""
import numpy as np
import pytest

import oracle_py as orc
from iterative_solvers_amd import capi, matgen, solver as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lap256():
    rp, col, val = matgen.laplace3d_7pt(256)
    return rp, col, val


def test_spmv_256cubed_bit_exact_all_encodings(gpu_handle, lap256):
    rp, col, val = lap256
    n = len(rp) - 1
    x = np.random.default_rng(7).uniform(-1, 1, n)
    orc.set_threads(min(16, orc.num_threads()))      # row sums are independent of the thread count
    ref = orc.spmv(rp, col, val, x)
    h = gpu_handle
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    try:
        for idx, vd in ((0, 0), (1, 0), (1, 1)):
            h.set_option("spmv_idx", idx); h.set_option("spmv_valdict", vd)
            y = h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT)
            assert np.array_equal(y, ref), (idx, vd)
    finally:
        h.set_option("spmv_idx", 1); h.set_option("spmv_valdict", 0)
    # exact scaling by a power of two commutes with every rounding: A(4x) == 4 A(x) bit for bit
    y4 = h.spmv_host(4.0 * x, n, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert np.array_equal(y4, 4.0 * ref)
    # reordered variant within 1e-12 of |A||x|
    yv = h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_VECTOR)
    absrow = orc.spmv(rp, col, np.abs(val), np.abs(x))
    assert np.all(np.abs(yv - ref) <= 1e-12 * absrow)


def test_cg_256cubed_history(lap256):
    """40 CG trips on configs[1] against the oracle (1 thread == the compiled reference).  With 1.7e7 terms
    per dot product the reference's own sequential sums carry ~1e-11 of rounding, which CG amplifies trip by
    trip: the oracle at 8 threads (different chunking of the same sums) leaves the 1e-10 band around the
    1-thread history after about 20 trips.  The bound is therefore the 1e-10 bar, widened where needed to
    10x the oracle's own 1-vs-8-thread deviation (running maximum, as in test_gpu_solvers.check_history)."""
    rp, col, val = lap256
    n = len(rp) - 1
    b = np.ones(n)
    trips = 40
    orc.set_threads(1)
    want = orc.solve(rp, col, val, b, b.copy(), orc.CG, fixed_iters=trips, max_iters=trips)["all_values"]
    orc.set_threads(8)
    alt = orc.solve(rp, col, val, b, b.copy(), orc.CG, fixed_iters=trips, max_iters=trips)["all_values"]
    orc.set_threads(1)
    args = S.Args("conjugate-gradient", S.LoopParams(max_iters=10**9, tol=0.0))
    sol = S.Solver("conjugate-gradient", device=0)
    try:
        S.preprocessing_crs(args, sol, rp, col, val, b, b.copy())
        got = np.array([sol.handle.iterate(it) for it in range(trips)])
    finally:
        sol.handle.close()
    k = min(len(want), len(alt), len(got))
    assert k >= trips - 1
    w, a, g = want[:k], alt[:k], got[:k]
    noise = np.maximum.accumulate(np.abs(a - w))
    noise = np.concatenate([noise[8:], np.full(min(8, k), noise[-1])])           # look 8 trips ahead
    bound = np.maximum(1e-10 * np.maximum(w, 1e-2 * w[0]), 10.0 * noise)
    assert np.all(np.abs(g[:10] - w[:10]) <= 1e-10 * w[:10]), (g[:4], w[:4])      # the bar itself, early trips
    assert np.all(np.abs(g - w) <= bound), (np.abs(g - w) / w, bound / w)


def test_spmv_27pt_128cubed_bit_exact(gpu_handle):
    rp, col, val = matgen.convdiff3d_27pt(128)
    n = len(rp) - 1
    x = np.random.default_rng(8).uniform(-1, 1, n)
    orc.set_threads(min(16, orc.num_threads()))
    ref = orc.spmv(rp, col, val, x)
    gpu_handle.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert np.array_equal(gpu_handle.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref)


def test_spmv_powerlaw_1m_bit_exact_and_sell(gpu_handle):
    rp, col, val = matgen.powerlaw(1_000_000)
    n = len(rp) - 1
    x = np.random.default_rng(9).uniform(-1, 1, n)
    orc.set_threads(min(16, orc.num_threads()))
    ref = orc.spmv(rp, col, val, x)
    absrow = orc.spmv(rp, col, np.abs(val), np.abs(x))
    gpu_handle.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_SELL)
    assert np.array_equal(gpu_handle.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref)
    for fmt in (capi.FMT_CSR_VECTOR, capi.FMT_SELL):
        assert np.all(np.abs(gpu_handle.spmv_host(x, n, capi.MAT_A, fmt) - ref) <= 1e-12 * absrow), fmt
    orc.set_threads(1)
