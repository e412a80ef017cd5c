"""Gauss-Seidel, option "gs_prefix" (csrc/its_api.hip trip_gs, its_handle.h): the sweep's chain is the residual's prefix.

With ascending columns a row of A is [its entries of L | diagonal | its entries of U].  The reference records
b - A x_new after every sweep (utility_funcs.hpp:444-447: spmv_crs_cpu's in-order chain over the whole row,
kernels.hpp:380-409) — and the sweep of that row has just run the first part of exactly that chain over L with the same x
(spltsv_crs, kernels.hpp:460-472).  With the option on (default) the sweep leaves its chain sums in place of the
right-hand side it has consumed and the pass that records the residual (and produces the next trip's b - U x) walks
D + U only, its full chain starting from those sums: the entries of L are read once per trip instead of twice.

Same operations in the same order, so the iterates must be BIT-identical to gs_prefix = 0 (the pass over A) on every
kernel family the sweep and the pass can land on, and equal the oracle's as before.  The recorded norms are sums of
per-workgroup partial sums whose grouping follows the image that is walked: 1e-13."""
import numpy as np
import pytest

import oracle_py as orc
from iterative_solvers_amd import capi, matgen, solver as S

pytestmark = pytest.mark.gpu

GEN = {
    "lap2d_64": lambda: matgen.laplace2d_5pt(64),          # natural order: runs of small levels in one workgroup
    "lap2d_200": lambda: matgen.laplace2d_5pt(200),
    "lap3d_16": lambda: matgen.laplace3d_7pt(16),
    "lap3d_32": lambda: matgen.laplace3d_7pt(32),          # multicolour: colours swept from ELL images of their own
    "lap3d_40": lambda: matgen.laplace3d_7pt(40),          # natural order: level-ordered copy, a launch per level
    "cd27_12": lambda: matgen.convdiff3d_27pt(12),
    "cd27_24": lambda: matgen.convdiff3d_27pt(24),         # byte-index ELL
    "pl_5000": lambda: matgen.powerlaw(5000),
    "pl_60000": lambda: matgen.powerlaw(60000),            # irregular rows: level lists / EPI_SWEEP on the irregular-row kernel
}


def run_gs(rp, col, val, order, prefix, trips=25, opts=(), x0=None):
    """prefix: the option's value — 0 off, 1 the default (the form only where it pays: irregular matrices), 2 forced"""
    n = len(rp) - 1
    args = S.Args("gauss-seidel", S.LoopParams(max_iters=trips, tol=1e-300), S.Flags(), "", "none")
    sol = S.Solver("gauss-seidel", device=0, gs_order=order)
    try:
        sol.handle.set_option("gs_prefix", prefix)
        for k, v in opts:
            sol.handle.set_option(k, v)
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.zeros(n) if x0 is None else x0)
        info = sol.handle.gs_info()
        S.solve(args, sol)
        return (args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy(), info,
                None if sol.perm is None else sol.perm.copy())
    finally:
        sol.handle.close()


def compare(a, b):
    assert len(a[0]) == len(b[0]) and len(a[0]) > 3
    assert np.array_equal(a[1], b[1])                                    # the iterate: bit for bit
    assert np.all(np.abs(a[0] - b[0]) <= 1e-13 * np.maximum(np.abs(b[0]), 1e-300))


@pytest.mark.parametrize("gen", sorted(GEN))
@pytest.mark.parametrize("order", [capi.GS_LEVEL, capi.GS_MULTICOLOR])
def test_prefix_form_gives_the_same_iterates(gen, order):
    rp, col, val = GEN[gen]()
    on = run_gs(rp, col, val, order, 2)
    off = run_gs(rp, col, val, order, 0)
    assert on[2]["prefix_ready"] and not off[2]["prefix_ready"]
    n = len(rp) - 1
    rpp, cp, _ = (rp, col, val) if on[3] is None else capi.permute_crs(rp, col, val, on[3])
    assert on[2]["nnz_du"] == int(((cp >= np.repeat(np.arange(n), np.diff(rpp)))).sum()) and on[2]["kernel"] != ""
    compare(on, off)


@pytest.mark.parametrize("gen,order", [("lap3d_16", capi.GS_LEVEL), ("lap3d_32", capi.GS_MULTICOLOR), ("pl_5000", capi.GS_LEVEL),
                                       ("cd27_12", capi.GS_MULTICOLOR)])
def test_prefix_form_equals_the_oracle(gen, order):
    """The chain is the reference's: x against the oracle's natural-order sweep (of the permuted matrix for the multicolour
    order), history at the bar of every Gauss-Seidel test."""
    rp, col, val = GEN[gen]()
    n = len(rp) - 1
    on = run_gs(rp, col, val, order, 2, trips=30)
    assert on[2]["prefix_ready"]
    orc.set_threads(1)
    if on[3] is not None:
        rpp, cp, vp = capi.permute_crs(rp, col, val, on[3])
    else:
        rpp, cp, vp = rp, col, val
    ref = orc.solve(rpp, cp, vp, np.ones(n), np.zeros(n), orc.GAUSS_SEIDEL, max_iters=30, tol=1e-300)
    x_ref = ref["x_star"]
    if on[3] is not None:
        x_ref = np.empty(n); x_ref[on[3]] = ref["x_star"]
    assert np.array_equal(on[1], x_ref)
    k = min(len(on[0]), len(ref["history"]))
    assert k > 3 and np.max(np.abs(on[0][:k] - ref["history"][:k])) <= 1e-10 * ref["r0_two"]


@pytest.mark.parametrize("opts", [
    (("spmv_ell", 0),),                                   # the pass over D + U on the CRS stream kernels
    (("sweep_ell_levels", 0),),                           # colours on the CRS stream kernels (EPI_SWEEP)
    (("sweep_stream", 0),),                               # every level on the level-list kernels
    (("gs_first_level_fused", 0),),                       # first colour swept by a launch of its own
    (("sweep_ell", 0),),                                  # no level-ordered copy
    (("sweep_run", 0),),                                  # a launch per level
])
@pytest.mark.parametrize("gen,order", [("lap3d_32", capi.GS_MULTICOLOR), ("lap2d_64", capi.GS_LEVEL), ("cd27_24", capi.GS_MULTICOLOR)])
def test_prefix_form_on_every_sweep_and_pass_kernel(gen, order, opts):
    rp, col, val = GEN[gen]()
    reset = {"spmv_ell": 1}
    try:
        on = run_gs(rp, col, val, order, 2, opts=opts)
        off = run_gs(rp, col, val, order, 0, opts=opts)
    finally:
        h = capi.Handle(0)
        for k, v in opts:
            if k in reset:
                h.set_option(k, reset[k])                 # (process-wide launch tunables)
        h.close()
    assert on[2]["prefix_ready"]
    compare(on, off)


def test_prefix_form_on_the_panel_image():
    """The power-law class with the column-panel image forced onto a small matrix: pass over the panel image of D + U plus
    the wave-per-row launch for its long rows, colours swept by the irregular-row kernel."""
    rp, col, val = matgen.powerlaw(60000)
    h = capi.Handle(0)
    try:
        h.set_option("spmv_panel_min_cols", 1); h.set_option("spmv_panel_log2pw", 9)
        res = {}
        for order in (capi.GS_MULTICOLOR, capi.GS_LEVEL):
            on = run_gs(rp, col, val, order, 2)
            off = run_gs(rp, col, val, order, 0)
            assert on[2]["prefix_ready"] and on[2]["kernel"] == "spmv_panel_kernel"
            compare(on, off)
            res[order] = on
    finally:
        h.set_option("spmv_panel_min_cols", 1 << 21); h.set_option("spmv_panel_log2pw", 16)
        h.close()


def test_rows_that_are_not_L_diagonal_U_keep_the_pass_over_A():
    """Unsorted columns inside the rows (the reference never sorts): crs_L / crs_U are still A's parts, but a row is no
    longer [L | diagonal | U] in that order — the set-up's device check refuses, the trips run as before."""
    rng = np.random.default_rng(5)
    rp, col, val = matgen.laplace2d_5pt(40)
    col = col.copy(); val = val.copy()
    for i in range(len(rp) - 1):
        p = rng.permutation(rp[i + 1] - rp[i])
        col[rp[i]:rp[i + 1]] = col[rp[i]:rp[i + 1]][p]; val[rp[i]:rp[i + 1]] = val[rp[i]:rp[i + 1]][p]
    on = run_gs(rp, col, val, capi.GS_LEVEL, 2)
    off = run_gs(rp, col, val, capi.GS_LEVEL, 0)
    assert not on[2]["prefix_ready"]
    assert np.array_equal(on[1], off[1]) and np.array_equal(on[0], off[0])


def test_nonzero_start_and_random_values():
    """Random values and a random start on a ragged sorted matrix with every diagonal present."""
    rng = np.random.default_rng(11)
    n = 3000
    rows = []
    for i in range(n):
        c = np.union1d(rng.choice(n, int(rng.integers(0, 40)), replace=False), [i])
        rows.append(np.sort(c))
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    val = rng.uniform(-1, 0, int(rp[-1]))
    for i in range(n):                                                   # strictly diagonally dominant
        j = rp[i] + int(np.searchsorted(col[rp[i]:rp[i + 1]], i))
        val[j] = 1.0 + np.abs(val[rp[i]:rp[i + 1]]).sum()
    x0 = rng.uniform(-1, 1, n)
    for order in (capi.GS_LEVEL, capi.GS_MULTICOLOR):
        on = run_gs(rp, col, val, order, 2, x0=x0)
        off = run_gs(rp, col, val, order, 0, x0=x0)
        assert on[2]["prefix_ready"]
        compare(on, off)


def test_default_takes_the_form_for_irregular_matrices_only():
    for gen, want in (("pl_5000", True), ("lap3d_16", False), ("cd27_12", False)):
        rp, col, val = GEN[gen]()
        assert run_gs(rp, col, val, capi.GS_LEVEL, 1, trips=5)[2]["prefix_ready"] == want
