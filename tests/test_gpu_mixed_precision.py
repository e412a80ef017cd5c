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
