"""GPU: mixed-precision matrix storage (ITS_FMT_MIXED; SURVEY.md §8f.3, the reference's USE_AP "ap[dp_sp]" idea,
main.cpp:171-184 / config.mk:56-63) against the fp64 oracle.  Opt-in and not bit-exact by design; what is asserted is
the stated bound: per row |y_mixed - y_fp64| <= 2^-24 * sum over the fp32-stored entries of |a_ij x_j| plus the
1e-12 reordering allowance of the vector variants — and that a CG solve on the mixed image converges to the fp64
solution within what that perturbation of A permits."""
import numpy as np
import pytest

import oracle_py as orc
from helpers import MATRICES
from iterative_solvers_amd import capi, solver as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,thr", [("pl_5000", 1.0), ("pl_5000", 0.5), ("cd27_12", 2.0), ("lap3d_16", 2.0), ("pl_200", 0.0),
                                      ("pl_200", float("inf"))])
def test_mixed_spmv_error_bound(gpu_handle, name, thr):
    h = gpu_handle
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    # values that are NOT exactly representable in fp32, so that the bound is exercised
    val = val * (1.0 + 1e-9 * np.sin(np.arange(len(val))))
    x = np.random.default_rng(3).uniform(-1, 1, n)
    h.set_mixed_threshold(thr)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_MIXED)
    n64, n32 = h.mixed_info()
    lo = np.abs(val) < thr
    assert n32 == int(lo.sum()) and n64 == len(val) - n32
    y = h.spmv_host(x, n, capi.MAT_A, capi.FMT_MIXED)
    ref = orc.spmv(rp, col, val, x)
    absrow_lo = orc.spmv(rp, col, np.where(lo, np.abs(val), 0.0), np.abs(x))
    absrow = orc.spmv(rp, col, np.abs(val), np.abs(x))
    bound = 2.0 ** -24 * absrow_lo + 1e-12 * np.maximum(absrow, 1e-3 * np.abs(val).max())
    assert np.all(np.abs(y - ref) <= bound), float(np.max(np.abs(y - ref) / np.maximum(bound, 1e-300)))
    if n32 and thr != float("inf"):
        # and it really is the fp32-rounded matrix that was multiplied: equals the oracle on that matrix to 1e-12
        val32 = np.where(lo, val.astype(np.float32).astype(np.float64), val)
        assert np.all(np.abs(y - orc.spmv(rp, col, val32, x)) <= 1e-12 * np.maximum(absrow, 1e-3 * np.abs(val).max()))
    # the exact path of the same upload is untouched
    assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref)
    h.set_mixed_threshold(1.0)


def test_cg_on_the_mixed_image_converges_to_the_fp64_solution():
    rp, col, val = MATRICES["lap3d_16"]()
    n = len(rp) - 1
    val = val * (1.0 + 1e-9 * np.cos(np.arange(len(val))))           # not fp32-representable; still SPD enough for CG
    val = 0.5 * (val + val)                                           # (kept symmetric: the perturbation depends on k only
    rows = np.repeat(np.arange(n), np.diff(rp))                      #  through the pair below)
    key = np.minimum(rows, col) * n + np.maximum(rows, col)
    val = MATRICES["lap3d_16"]()[2] * (1.0 + 1e-9 * np.cos(key % 1000))
    out = {}
    for fmt in (capi.FMT_CSR_EXACT, capi.FMT_MIXED):
        args = S.Args("conjugate-gradient", S.LoopParams(max_iters=500, tol=1e-10), S.Flags(), "", "none")
        sol = S.Solver("conjugate-gradient", device=0, spmv_format=fmt)
        try:
            sol.handle.set_mixed_threshold(2.0)                       # off-diagonals (|a| ~ 1) in fp32, diagonal (6) in fp64
            S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
            S.solve(args, sol)
            out[fmt] = (args.loop_params.iter_count, args.flags.convergence_flag, sol.x_star.copy())
        finally:
            sol.handle.close()
    (it0, ok0, x0), (it1, ok1, x1) = out[capi.FMT_CSR_EXACT], out[capi.FMT_MIXED]
    assert ok0 and ok1 and abs(it1 - it0) <= 2
    # A is perturbed by <= 2^-24 relative in its off-diagonal entries: the solution moves by about cond(A) * 6e-8
    assert np.max(np.abs(x1 - x0)) <= 1e-5 * np.max(np.abs(x0))
    assert np.max(np.abs(x1 - x0)) > 0.0


# ---- the reference's AP_VALUE_TYPE modes with two thresholds and the fp16 type (config.mk:56-63), its_set_mixed_mode ----
def py_tier(v, mode, t1, t2):
    """its_ap::tier restated (csrc/its_ap.h): 0 fp64, 1 fp32, 2 fp16"""
    a = np.abs(v)
    hi = ~(a < t1)
    if mode == capi.AP_DP_SP:
        t = np.where(hi, 0, 1)
    elif mode == capi.AP_DP_HP:
        t = np.where(hi, 0, 2)
    elif mode == capi.AP_SP_HP:
        t = np.where(hi, 1, 2)
    else:
        t = np.where(hi, 0, np.where(~(a < t2), 1, 2))
    t = np.where((t == 2) & ~(a <= 65504.0), 1, t)
    t = np.where((t == 1) & ~(a <= np.finfo(np.float32).max), 0, t)
    return t


def py_stored(v, t):
    with np.errstate(over="ignore"):
        v32 = v.astype(np.float32)
        return np.where(t == 0, v, np.where(t == 1, v32.astype(np.float64), v32.astype(np.float16).astype(np.float64)))


@pytest.mark.parametrize("mode,t1,t2", [(capi.AP_DP_SP, 2.0, 0.0), (capi.AP_DP_HP, 2.0, 0.0), (capi.AP_SP_HP, 2.0, 0.0),
                                        (capi.AP_DP_SP_HP, 2.0, 0.5), (capi.AP_DP_SP_HP, 1e30, 1e29), (capi.AP_DP_HP, 1e30, 0.0)])
@pytest.mark.parametrize("name", ["lap3d_16", "lap2d_33", "pl_5000", "cd27_12"])
def test_tiers_multiply_the_matrix_of_the_stored_values(gpu_handle, name, mode, t1, t2):
    """Entries are stored as fp64 / fp32 / fp16 by magnitude; the product is that of the matrix of the stored values:
    BIT-identical to the oracle on that matrix where the tiered sliced-ELL image is walked (in-order chains), 1e-12 on the
    CRS split (lanes share a row).  Values spread over eleven decades so that every tier, fp16 subnormals and the
    beyond-65504 promotion all occur."""
    h = gpu_handle
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    rng = np.random.default_rng(17)
    val = val * (1.0 + 1e-3 * np.sin(np.arange(len(val)))) * 10.0 ** rng.integers(-6, 6, len(val))
    x = rng.uniform(-1, 1, n)
    h.set_mixed_mode(mode, t1, t2)
    try:
        h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_MIXED)
        tiers = h.mixed_tiers()
        t = py_tier(val, mode, t1, t2)
        assert tiers["nnz"] == [int((t == k).sum()) for k in range(3)]
        stored = py_stored(val, t)
        ref = orc.spmv(rp, col, stored, x)
        y = h.spmv_host(x, n, capi.MAT_A, capi.FMT_MIXED)
        assert tiers["ell_form"] == (name in ("lap3d_16", "lap2d_33"))
        if tiers["ell_form"]:
            assert np.array_equal(y, ref)
            # every (slice, slot) run in the narrowest type its entries allow: never more bytes than all-fp64
            assert tiers["value_bytes"] <= 8 * 128 * ((n + 127) // 128) * int(np.diff(rp).max())
            h.set_option("spmv_mixed_ell", 0)
            try:
                y2 = h.spmv_host(x, n, capi.MAT_A, capi.FMT_MIXED)
            finally:
                h.set_option("spmv_mixed_ell", 1)
            absrow = orc.spmv(rp, col, np.abs(stored), np.abs(x))
            assert np.all(np.abs(y2 - ref) <= 1e-12 * np.maximum(absrow, 1e-300))
        else:
            absrow = orc.spmv(rp, col, np.abs(stored), np.abs(x))
            assert np.all(np.abs(y - ref) <= 1e-12 * np.maximum(absrow, 1e-300))
            assert tiers["value_bytes"] == 8 * tiers["nnz"][0] + 4 * tiers["nnz"][1] + 2 * tiers["nnz"][2]
        # the stated bound against the fp64 product: 2^-24 per fp32 entry, 2^-11 (+ 2^-25 absolute: subnormals) per fp16 entry
        exact = orc.spmv(rp, col, val, x)
        bound = (orc.spmv(rp, col, np.where(t == 1, np.abs(val), 0.0), np.abs(x)) * 2.0 ** -24 +
                 orc.spmv(rp, col, np.where(t == 2, np.abs(val) * (2.0 ** -11 + 2.0 ** -24) + 2.0 ** -25, 0.0), np.abs(x)) +
                 1e-12 * np.maximum(orc.spmv(rp, col, np.abs(val), np.abs(x)), 1e-300))
        assert np.all(np.abs(y - exact) <= bound)
        # residual form with the fused norm, and the exact path of the same upload untouched
        assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), exact)
    finally:
        h.set_mixed_threshold(1.0)


def test_tiered_image_bytes_on_a_stencil(gpu_handle):
    """7-pt Laplacian, AP_THRESHOLD_1 = 2: interior slices hold the diagonal in one slot (fp64) and six slots of -1
    (fp32 / fp16): 8 + 6 x 4 resp. 8 + 6 x 2 bytes of values per row, against 56 in fp64 — and, -1 being exact in every
    type, the product equals the fp64 one bit for bit."""
    h = gpu_handle
    rp, col, val = matgen_lap(48)
    n = len(rp) - 1
    x = np.random.default_rng(2).uniform(-1, 1, n)
    exact = orc.spmv(rp, col, val, x)
    try:
        per_row = {}
        for mode in (capi.AP_DP_SP, capi.AP_DP_HP):
            h.set_mixed_mode(mode, 2.0, 0.0)
            h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_MIXED)
            tiers = h.mixed_tiers()
            assert tiers["ell_form"]
            per_row[mode] = tiers["value_bytes"] / n
            assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_MIXED), exact)
        # boundary slices keep a few slots wider (the diagonal sits in slots 0..3 there): within 25 % of the interior figure
        assert 32.0 <= per_row[capi.AP_DP_SP] <= 40.0 and 20.0 <= per_row[capi.AP_DP_HP] <= 28.0
    finally:
        h.set_mixed_threshold(1.0)


def matgen_lap(N):
    from iterative_solvers_amd import matgen
    return matgen.laplace3d_7pt(N)


def test_set_mixed_mode_rejects_bad_arguments(gpu_handle):
    h = gpu_handle
    for bad in ((7, 1.0, 0.0), (capi.AP_DP_SP_HP, 1.0, 2.0), (capi.AP_DP_SP, float("nan"), 0.0)):
        with pytest.raises(Exception):
            h.set_mixed_mode(*bad)


def test_cg_on_the_fp16_tier_of_a_laplacian_is_the_fp64_run():
    """ap[dp_hp], threshold 2 on the 7-pt Laplacian: every stored value is exact, the chains are the reference's, so CG's
    iterates are those of the fp64 exact path (same kernels for the vector passes, same work map for the partial sums)."""
    rp, col, val = MATRICES["lap3d_16"]()
    n = len(rp) - 1
    out = {}
    for fmt in (capi.FMT_CSR_EXACT, capi.FMT_MIXED):
        args = S.Args("conjugate-gradient", S.LoopParams(max_iters=500, tol=1e-10), S.Flags(), "", "none")
        sol = S.Solver("conjugate-gradient", device=0, spmv_format=fmt)
        try:
            sol.handle.set_mixed_mode(capi.AP_DP_HP, 2.0, 0.0)
            S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
            S.solve(args, sol)
            out[fmt] = (args.loop_params.iter_count, sol.x_star.copy(), args.normed_residuals[:args.loop_params.residual_count].copy())
        finally:
            sol.handle.close()
    assert out[capi.FMT_CSR_EXACT][0] == out[capi.FMT_MIXED][0]
    assert np.max(np.abs(out[capi.FMT_CSR_EXACT][1] - out[capi.FMT_MIXED][1])) <= 1e-12 * np.max(np.abs(out[capi.FMT_CSR_EXACT][1]))
