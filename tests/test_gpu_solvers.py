"""GPU parity: whole solves through the C ABI (host mirror in iterative_solvers_amd/solver.py)
against the oracle, which is itself pinned bit-for-bit to the compiled reference.

Bar (BASELINE.json north_star): residual history within 1e-10 relative, iteration count ±1 %.
Reading used here (SURVEY.md §7 "hard parts"): |r_k^gpu − r_k^cpu| ≤ 1e-10·max(r_k^cpu, 1e-2·‖r_0‖)
for every recorded trip — 1e-10 relative over the first two decades of convergence and
1e-12·‖r_0‖ absolute below that (100x tighter than the |Δ|/‖r_0‖ ≤ 1e-10 reading SURVEY.md
proposes).  Below ≈1e-13·‖r_0‖ neither side has correct digits left: a recomputed residual
b − A x (Jacobi/GS every trip, GMRES at every restart) carries an absolute error of about
eps·‖A‖‖x‖, e.g. restarted GMRES(10) on lap2d_16 differs by 5e-7 RELATIVE at ‖r‖ = 1.4e-6,
which is 6e-14·‖r_0‖.

Ill-conditioned histories: some runs amplify rounding noise by many orders (restarted GMRES(10)
on lap2d_16: the reference restatement ITSELF moves by 3-4 % at trip ~100 when only its OpenMP
thread count, i.e. the order of the dot-product reduction, changes from 1 to 2/3/4 — measured).
There a fixed 1e-10 bound tests the noise, not the implementation.  check_history therefore
widens the bound at trip k to 10x the largest oracle-vs-oracle deviation seen up to trip k+8
(oracle at 1 thread vs 2..6 threads; the multi-thread combine order is itself not
deterministic, hence the margin): the GPU may deviate from the reference no more than the
reference deviates from itself under a reduction-order change.

Iteration counts: with the reference's default TOL=1e-14 the stopping value lies BELOW the
attainable accuracy (‖r‖/‖r_0‖ ≈ 1e-16), so the last ~10 % of trips are rounding noise: the
compiled reference itself needs 185 trips at 1 thread and 177 at 32 threads on lap2d_64
(measured).  The ±1 % bar is therefore asserted with TOL=1e-10 (above the noise floor); the
TOL=1e-14 runs assert the history bound and a count band of twice the spread the oracle itself shows between
1 and 2..6 threads (measured in the test, helpers.oracle_with_noise), not a constant.  The oracle runs with 1 thread
(= bit-identical to the compiled reference at OMP_NUM_THREADS=1).
"""
import numpy as np
import pytest

import oracle_py as orc
from helpers import MATRICES, oracle_with_noise
from iterative_solvers_amd import capi, solver as S

pytestmark = pytest.mark.gpu

KIND = {"-j": orc.JACOBI, "-gs": orc.GAUSS_SEIDEL, "-cg": orc.CG, "-gm": orc.GMRES}


def run_both(name, flag, scale="none", precond="", max_iters=400, restart=100, jacobi_mode=0, tol=1e-10,
             fmt=capi.FMT_CSR_EXACT, options=None, info=None):
    orc.set_threads(1)
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    # reference preprocessing on the COO (oracle) and in the product host code (numpy)
    Vo, rs_o = orc.scale_coo(n, rows, col, val, scale)
    rpo, colo, valo = orc.coo_to_crs(n, rows, col, Vo)
    bo = 1.0 / (rs_o * rs_o)
    kw = dict(jacobi_mode=jacobi_mode, precond=orc.PRECOND[precond], max_iters=max_iters, restart_len=restart, tol=tol)
    # the oracle, and the reference's own sensitivity to reduction order (see module docstring)
    ref = oracle_with_noise(orc, rpo, colo, valo, bo, bo, KIND[flag], **kw)
    args = S.Args(S.CLI_SOLVER[flag], S.LoopParams(max_iters=max_iters, tol=tol, gmres_restart_len=restart), S.Flags(),
                  {"": "", "gs": "gauss-seidel"}[precond], scale)
    sol = S.Solver(args.solver_type, args.preconditioner_type, device=0, spmv_format=fmt, jacobi_mode=jacobi_mode)
    try:
        for key, value in (options or {}).items():
            sol.handle.set_option(key, value)
        S.preprocessing(args, sol, n, rows.astype(np.int64), col.astype(np.int64), val.copy())
        S.solve(args, sol)
        xs = sol.x_star.copy()
        if info is not None and flag == "-gm":
            info.update(sol.handle.gmres_info())
    finally:
        sol.handle.close()
    return ref, args, xs


def check_history(ref, args, tol_rel=1e-10, count_band=0.01, may_stop_early=False):
    lp = args.loop_params
    assert abs(lp.stopping_criteria - ref["stopping"]) <= 1e-13 * abs(ref["stopping"])
    # +-1 % (the north_star bar), or twice what the reference's own count moves under a change of its reduction
    # order, whichever is larger (only runs below the attainable accuracy, TOL = 1e-14, ever need the latter)
    it_tol = max(1, int(round(count_band * ref["iters"])), 2 * ref.get("iters_spread", 0))
    if may_stop_early:       # the caller checks that an earlier stop is a legitimate one (true residual)
        assert lp.iter_count <= ref["iters"] + it_tol, (lp.iter_count, ref["iters"])
    else:
        assert abs(lp.iter_count - ref["iters"]) <= it_tol, (lp.iter_count, ref["iters"])
    assert args.flags.convergence_flag == ref["converged"]
    k = min(lp.residual_count, len(ref["history"]))
    got = args.normed_residuals[:k]; exp = ref["history"][:k]
    r0 = ref["r0_two"]
    bound = np.maximum(tol_rel * np.maximum(exp, 1e-2 * r0), 10.0 * ref["noise"][:k])
    bad = np.nonzero(np.abs(got - exp) > bound)[0]
    assert bad.size == 0, f"first mismatch at trip {bad[0]}: {got[bad[0]]!r} vs {exp[bad[0]]!r}"


@pytest.mark.parametrize("name", ["lap2d_16", "lap2d_64", "lap3d_16", "pl_200"])
@pytest.mark.parametrize("scale", ["none", "diag", "max"])
def test_cg(name, scale):
    ref, args, xs = run_both(name, "-cg", scale, max_iters=600)
    check_history(ref, args)
    assert np.max(np.abs(xs - ref["x_star"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["x_star"])))


@pytest.mark.parametrize("name", ["lap2d_64", "lap3d_16"])
def test_cg_reference_default_tol(name):
    """TOL=1e-14 (config.mk default) lies below the attainable accuracy: the trip at which the recurrence residual
    happens to dip under it is decided by rounding noise, and a MORE accurate summation gets there sooner (the
    oracle itself: 185 trips at 1-4 threads, 176-177 at 6-128 threads on lap2d_64; the GPU's tree reductions: 163).
    No constant band: the history must stay inside the measured noise envelope, the run must not take longer than
    the reference (+1 % / twice the oracle's own count spread), and if it stops EARLIER its solution must be as good
    as the reference's: true residual ||b - A x_star|| no larger than the reference's own at its stopping trip."""
    ref, args, xs = run_both(name, "-cg", "none", max_iters=600, tol=1e-14)
    check_history(ref, args, may_stop_early=True)
    rp, col, val = MATRICES[name]()
    b = np.ones(len(rp) - 1)
    true_gpu = np.linalg.norm(b - orc.spmv(rp, col, val, xs))
    true_ref = np.linalg.norm(b - orc.spmv(rp, col, val, ref["x_star"]))
    assert true_gpu <= 1.5 * true_ref + 1e-15 * np.linalg.norm(b), (true_gpu, true_ref, args.loop_params.iter_count, ref["iters"])


@pytest.mark.parametrize("fmt", [capi.FMT_CSR_VECTOR, capi.FMT_SELL])
def test_cg_reordered_formats(fmt):
    ref, args, xs = run_both("lap3d_16", "-cg", "none", fmt=fmt)
    check_history(ref, args)


@pytest.mark.parametrize("name,restart", [("lap2d_16", 100), ("lap2d_16", 10), ("cd27_12", 50), ("cd27_12", 10),
                                           ("lap3d_16", 50), ("pl_200", 10)])
def test_gmres(name, restart):
    ref, args, xs = run_both(name, "-gm", "none", restart=restart, max_iters=500)
    check_history(ref, args)
    assert np.max(np.abs(xs - ref["x_star"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["x_star"])))


@pytest.mark.parametrize("name", ["lap2d_16", "cd27_6"])
def test_gmres_gs_preconditioned(name):
    ref, args, xs = run_both(name, "-gm", "diag", precond="gs", restart=10, max_iters=300)
    check_history(ref, args)


@pytest.mark.parametrize("name,scale", [("lap2d_16", "none"), ("lap2d_16", "diag"), ("cd27_6", "diag"), ("pl_200", "diag"), ("pl_5000", "none")])
def test_gauss_seidel_level_scheduled(name, scale):
    """Natural-order GS by dependency levels: same arithmetic per row as the serial sweep."""
    ref, args, xs = run_both(name, "-gs", scale, max_iters=150)
    check_history(ref, args)
    assert np.max(np.abs(xs - ref["x_star"])) <= 1e-12 * max(1.0, np.max(np.abs(ref["x_star"])))


@pytest.mark.parametrize("name,scale", [("lap2d_16", "diag"), ("lap2d_16", "none"), ("pl_200", "diag"), ("cd27_6", "max")])
def test_jacobi_shipped_bug_compatible(name, scale):
    """Bug-compatible with methods/jacobi.hpp:191 (swapped SpMV arguments): stalls or diverges
    exactly like the shipped binary."""
    ref, args, xs = run_both(name, "-j", scale, max_iters=120)
    check_history(ref, args, tol_rel=1e-10)


@pytest.mark.parametrize("name", ["lap2d_16", "pl_200", "pl_5000", "cd27_6"])
def test_jacobi_ref_correct_mode(name):
    """Correct sweep = jacobi_iteration_ref_cpu (methods/jacobi.hpp:65-93)."""
    ref, args, xs = run_both(name, "-j", "none", max_iters=200, jacobi_mode=1)
    check_history(ref, args)
    assert np.max(np.abs(xs - ref["x_star"])) <= 1e-12 * max(1.0, np.max(np.abs(ref["x_star"])))


@pytest.mark.parametrize("name", ["lap2d_16", "lap3d_16", "cd27_12", "pl_5000"])
def test_gauss_seidel_multicolor(name):
    """Multicolour GS: the library sweeps P A P^T (rows colour by colour) level by level, so the
    levels are the colours; the oracle's natural-order sweep on the SAME permuted matrix is the
    identical arithmetic (SURVEY.md §7 "GS ordering")."""
    orc.set_threads(1)
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    perm, nc = capi.multicolor_order(rp, col)
    rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
    b = np.ones(n)
    ref = orc.solve(rpp, cp, vp, b, b.copy(), orc.GAUSS_SEIDEL, max_iters=120, tol=1e-10)
    ref["noise"] = np.zeros(len(ref["history"]))
    args = S.Args("gauss-seidel", S.LoopParams(max_iters=120, tol=1e-10), S.Flags(), "", "none")
    sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_MULTICOLOR)
    try:
        S.preprocessing_crs(args, sol, rp, col, val, b, b.copy())
        assert sol.n_colors == nc and sol.n_colors <= {"lap2d_16": 2, "lap3d_16": 2, "cd27_12": 8}.get(name, 64)
        S.solve(args, sol)
        xs = sol.x_star.copy()
    finally:
        sol.handle.close()
    check_history(ref, args)
    x_ref = np.empty(n); x_ref[perm] = ref["x_star"]
    assert np.max(np.abs(xs - x_ref)) <= 1e-12 * max(1.0, np.max(np.abs(x_ref)))


@pytest.mark.parametrize("name", ["lap3d_16", "cd27_12", "pl_200", "pl_5000", "lap2d_64"])
def test_shipped_jacobi_one_pass_per_trip_is_bit_identical(name):
    """The shipped Jacobi sweep (jacobi.hpp:41-62, the reference's default) works on the full product A x, the chain of
    the residual of the same vector: the pass that sweeps x_new also records ||b - A x_new|| (EPI_JACOBI_SHIPPED with
    partial sums).  Same bits as sweep + residual pass (fused_residual = 0); the iterate is read after an odd and an even
    number of trips (the three iterate buffers rotate)."""
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    out = []
    for fused in (1, 0):
        runs = []
        for trips in (30, 31):
            args = S.Args("jacobi", S.LoopParams(max_iters=trips, tol=1e-13), S.Flags(), "", "none")
            sol = S.Solver("jacobi", device=0, jacobi_mode=0)
            try:
                sol.handle.set_option("fused_residual", fused)
                S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
                S.solve(args, sol)
                runs.append((args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy()))
            finally:
                sol.handle.close()
        out.append(runs)
    for a, c in zip(out[0], out[1]):
        assert len(a[0]) > 3 and np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


@pytest.mark.parametrize("valdict", [0, 1])
@pytest.mark.parametrize("name", ["lap3d_16", "cd27_6", "pl_200", "pl_5000"])
@pytest.mark.parametrize("flag", ["-j", "-gs"])
def test_one_matrix_pass_per_trip_is_bit_identical(flag, name, valdict):
    """Jacobi / GS trips read A once: the residual pass also produces the next trip's product of the same x
    (EPI_JACOBI_FUSED / EPI_GS_FUSED, second in-order chain).  Same bits as the separate launches
    (option fused_residual = 0), on every exact kernel (stream, value-dictionary, irregular rows)."""
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    out = []
    for fused in (1, 0):
        args = S.Args(S.CLI_SOLVER[flag], S.LoopParams(max_iters=60, tol=1e-13), S.Flags(), "", "none")
        sol = S.Solver(args.solver_type, device=0, jacobi_mode=1)
        try:
            sol.handle.set_option("fused_residual", fused)
            sol.handle.set_option("gs_prefix", 0)       # (the pass over D + U groups its partial sums differently: tests/test_gpu_gs_prefix.py)
            sol.handle.set_option("spmv_valdict", valdict)
            S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
            S.solve(args, sol)
            out.append((args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy(),
                        args.loop_params.iter_count))
        finally:
            sol.handle.set_option("spmv_valdict", 0)
            sol.handle.close()
    assert out[0][2] == out[1][2] and out[0][2] > 3
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("flag,jacobi_mode", [("-j", 0), ("-j", 1), ("-gs", 0), ("-cg", 0)])
def test_hub_rows_on_the_wave_per_row_launch_give_the_same_iterates(flag, jacobi_mode):
    """Rows of more than SPMV_SOLO_LEN (256) entries leave the irregular-row plan for spmv_long_kernel (one wave per
    row); option spmv_irr_split = 0 keeps them in the plan.  Every row's chain is the same in-order sum either way:
    Jacobi / GS iterates bit-identical; the recorded norms are sums of per-workgroup partial sums, whose grouping follows
    the plan (1e-13), and CG's scalars come from such sums (its x within 1e-12)."""
    from iterative_solvers_amd import matgen
    rp, col, val = matgen.powerlaw(20000)
    n = len(rp) - 1
    lens = np.diff(rp)
    assert ((lens > 256) & (lens <= 1536)).sum() > 0 and lens.max() > 1536
    out = []
    for split in (1, 0):
        args = S.Args(S.CLI_SOLVER[flag], S.LoopParams(max_iters=30, tol=1e-13), S.Flags(), "", "none")
        sol = S.Solver(args.solver_type, device=0, jacobi_mode=jacobi_mode)
        try:
            sol.handle.set_option("spmv_irr_split", split)
            _, r0_two = S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.zeros(n))
            S.solve(args, sol)
            out.append((args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy()))
        finally:
            sol.handle.set_option("spmv_irr_split", 1)
            sol.handle.close()
    assert len(out[0][0]) == len(out[1][0]) and len(out[0][0]) > 3
    assert np.all(np.abs(out[0][0] - out[1][0]) <= 1e-13 * np.maximum(np.abs(out[1][0]), r0_two))   # (shipped Jacobi diverges here)
    if flag == "-cg":
        assert np.max(np.abs(out[0][1] - out[1][1])) <= 1e-12 * np.max(np.abs(out[1][1]))
    else:
        assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("gen", ["lap3d_32", "cd27_24", "pl_60000"])
def test_multicolor_sweep_on_stream_kernels_is_bit_identical(gen):
    """A colour that is one contiguous range of >= 4096 rows is swept by the exact SpMV kernels with the sweep
    epilogue (EPI_SWEEP); option sweep_stream = 0 keeps the level-list kernel.  Same operations per row in
    the same order (kernels.hpp:460-472): identical histories and x, and both match the oracle's natural-order
    sweep of the permuted matrix."""
    from iterative_solvers_amd import matgen
    rp, col, val = {"lap3d_32": lambda: matgen.laplace3d_7pt(32), "cd27_24": lambda: matgen.convdiff3d_27pt(24),
                    "pl_60000": lambda: matgen.powerlaw(60000)}[gen]()
    n = len(rp) - 1
    b = np.ones(n)
    out = []
    for stream in (1, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=40, tol=1e-13), S.Flags(), "", "none")
        sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_MULTICOLOR)
        try:
            sol.handle.set_option("sweep_stream", stream)
            S.preprocessing_crs(args, sol, rp, col, val, b, np.zeros(n))
            S.solve(args, sol)
            out.append((args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy(), sol.perm.copy()))
        finally:
            sol.handle.close()
    assert len(out[0][0]) > 3
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    orc.set_threads(1)
    perm = out[0][2]
    rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
    ref = orc.solve(rpp, cp, vp, b, np.zeros(n), orc.GAUSS_SEIDEL, max_iters=40, tol=1e-13)
    x_ref = np.empty(n); x_ref[perm] = ref["x_star"]
    assert np.max(np.abs(out[0][1] - x_ref)) <= 1e-12 * max(1.0, np.max(np.abs(x_ref)))
    k = min(len(out[0][0]), len(ref["history"]))
    assert np.max(np.abs(out[0][0][:k] - ref["history"][:k])) <= 1e-10 * ref["r0_two"]
    assert ref["r0_two"] > 1.0


@pytest.mark.parametrize("gen", ["lap3d_32", "lap2d_200", "cd27_40"])
def test_multicolor_sweep_on_level_ell_images_is_bit_identical(gen):
    """A colour of a stencil in multicolour order is swept from an ELL image of its own rows of L (nibble indices for the
    5-/7-point colours, byte indices for the 27-point ones); option sweep_ell_levels = 0 keeps the CRS stream kernels.
    Identical histories and iterate, and the iterate equals the oracle's natural-order sweep of the permuted matrix."""
    from iterative_solvers_amd import matgen
    rp, col, val = {"lap3d_32": lambda: matgen.laplace3d_7pt(32), "lap2d_200": lambda: matgen.laplace2d_5pt(200),
                    "cd27_40": lambda: matgen.convdiff3d_27pt(40)}[gen]()
    n = len(rp) - 1
    b = np.ones(n)
    out = []
    for lev in (1, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=25, tol=0.0), S.Flags(), "", "none")
        sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_MULTICOLOR)
        try:
            sol.handle.set_option("sweep_ell_levels", lev)
            S.preprocessing_crs(args, sol, rp, col, val, b, np.zeros(n))
            S.solve(args, sol)
            out.append((args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy(), sol.perm.copy()))
        finally:
            sol.handle.close()
    assert len(out[0][0]) > 3
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    orc.set_threads(1)
    perm = out[0][2]
    rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
    ref = orc.solve(rpp, cp, vp, b, np.zeros(n), orc.GAUSS_SEIDEL, max_iters=25, tol=0.0)
    x_ref = np.empty(n); x_ref[perm] = ref["x_star"]
    assert np.array_equal(out[0][1], x_ref)


@pytest.mark.parametrize("gen", ["lap3d_32", "lap2d_200", "cd27_40", "lap3d_33"])
def test_first_colour_folded_into_the_residual_pass_is_bit_identical(gen):
    """The first colour of a multicolour sweep reads nothing from L: its next iterate is (b - U x) / D, which the
    one-pass-per-trip form (EPI_GS_FUSED on the ELL kernels) stores itself, into the spare iterate buffer; the sweep
    continues there from the second colour and the buffers swap (option gs_first_level_fused = 0: the separate
    elementwise kernel).  Identical histories; the iterate is read in the MIDDLE of the run too (an odd and an even
    number of swaps), and equals the oracle's.  lap3d_33: colours of odd size, the form does not apply — same result."""
    from iterative_solvers_amd import matgen
    rp, col, val = {"lap3d_32": lambda: matgen.laplace3d_7pt(32), "lap2d_200": lambda: matgen.laplace2d_5pt(200),
                    "cd27_40": lambda: matgen.convdiff3d_27pt(40), "lap3d_33": lambda: matgen.laplace3d_7pt(33)}[gen]()
    n = len(rp) - 1
    b = np.ones(n)
    out = []
    for pre in (1, 0):
        xs = []
        for trips in (6, 7, 25):
            args = S.Args("gauss-seidel", S.LoopParams(max_iters=trips, tol=0.0), S.Flags(), "", "none")
            sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_MULTICOLOR)
            try:
                sol.handle.set_option("gs_first_level_fused", pre)
                S.preprocessing_crs(args, sol, rp, col, val, b, np.zeros(n))
                S.solve(args, sol)
                xs.append((args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy(), sol.perm.copy()))
            finally:
                sol.handle.close()
        out.append(xs)
    for a, c in zip(out[0], out[1]):
        assert len(a[0]) > 3 and np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    orc.set_threads(1)
    perm = out[0][2][2]
    rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
    ref = orc.solve(rpp, cp, vp, b, np.zeros(n), orc.GAUSS_SEIDEL, max_iters=25, tol=0.0)
    x_ref = np.empty(n); x_ref[perm] = ref["x_star"]
    assert np.array_equal(out[0][2][1], x_ref)


def test_gs_fusion_is_refused_for_a_foreign_crs_U():
    """The one-pass GS trip takes its b - U x from A's entries right of the diagonal, which is only the same
    thing if crs_U is that part of A: count, column sum and value hash are compared, and a crs_U that differs
    (one value doubled here) keeps the separate pass over U.  Both settings must then walk the same launches."""
    rp, col, val = MATRICES["lap3d_8"]()
    n = len(rp) - 1
    (Lp, Lc, Lv), (Up, Uc, Uv) = S.split_L_U(n, rp, col, val)
    Uv = Uv.copy(); Uv[5] *= 2.0
    out = []
    for fused in (1, 0):
        h = capi.Handle(0)
        try:
            h.set_option("fused_residual", fused)
            h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
            h.upload_crs(Lp, Lc, Lv, capi.MAT_L, capi.FMT_CSR_EXACT)
            h.upload_crs(Up, Uc, Uv, capi.MAT_U, capi.FMT_CSR_EXACT)
            h.upload_vectors(np.full(n, 6.0), np.ones(n), np.zeros(n))
            h.init_residual()
            h.solver_setup(capi.GAUSS_SEIDEL)
            out.append([h.iterate(i) for i in range(12)])
        finally:
            h.close()
    assert out[0] == out[1]


@pytest.mark.parametrize("flag", ["-cg", "-gm", "-j", "-gs"])
def test_odd_row_count(flag):
    """n = 225 (15 x 15 grid): the vector kernels take two elements per lane when their operands are 16-byte
    aligned; with an odd leading dimension every other Krylov vector is not, and the last element has no partner."""
    from iterative_solvers_amd import matgen
    MATRICES["lap2d_15"] = lambda: matgen.laplace2d_5pt(15)
    try:
        ref, args, xs = run_both("lap2d_15", flag, "none", max_iters=300, restart=7, jacobi_mode=1)
    finally:
        del MATRICES["lap2d_15"]
    check_history(ref, args)
    assert np.max(np.abs(xs - ref["x_star"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["x_star"])))


@pytest.mark.parametrize("gen", ["lap2d_256", "lap3d_48", "cd27_20", "lap2d_1100"])
def test_runs_of_small_levels_in_one_workgroup_are_bit_identical(gen):
    """Consecutive levels of at most 1024 rows of a natural-order sweep (all 511 of a 256 x 256 grid; the first and last
    ones of a 3-D grid; none of the middle ones of a 1100 x 1100 grid, whose levels reach 1100 rows) run as one launch of
    one workgroup with a barrier between levels (option sweep_run, default on) instead of a launch per level.  Same
    histories and iterates as sweep_run = 0 for GS and for GS-preconditioned GMRES (which sweeps in place, x == rhs),
    and the GS iterate equals the oracle's serial sweep."""
    from iterative_solvers_amd import matgen
    rp, col, val = {"lap2d_256": lambda: matgen.laplace2d_5pt(256), "lap3d_48": lambda: matgen.laplace3d_7pt(48),
                    "cd27_20": lambda: matgen.convdiff3d_27pt(20), "lap2d_1100": lambda: matgen.laplace2d_5pt(1100)}[gen]()
    n = len(rp) - 1
    out = []
    for run in (1, 0):
        res = []
        for solver, precond, iters in (("gauss-seidel", "", 12), ("gmres", "gauss-seidel", 12)):
            args = S.Args(solver, S.LoopParams(max_iters=iters, tol=1e-30, gmres_restart_len=5), S.Flags(), precond, "none")
            sol = S.Solver(solver, precond, device=0)
            try:
                sol.handle.set_option("sweep_run", run)
                S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
                S.solve(args, sol)
                res.append((args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy()))
            finally:
                sol.handle.close()
        out.append(res)
    for a, c in zip(out[0], out[1]):
        assert len(a[0]) > 3 and np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    if n <= 120000:
        orc.set_threads(1)
        ref = orc.solve(rp, col, val, np.ones(n), np.ones(n), orc.GAUSS_SEIDEL, max_iters=12, tol=1e-30)
        assert np.array_equal(out[0][0][1], ref["x_star"])


@pytest.mark.parametrize("name", ["lap3d_16", "lap2d_64", "cd27_6"])
def test_natural_order_sweep_level_ordered_copy_is_bit_identical(name):
    """Natural-order Gauss-Seidel on a stencil = many dependency levels of scattered short rows: they are swept from
    a level-ordered copy of L (option "sweep_ell", default on; two dependent memory round trips per level instead of
    four).  Same entries in the same order as the level-list kernel: identical histories and iterates, and equal to
    the oracle's serial sweep."""
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    out = []
    for ell in (1, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=40, tol=1e-13), S.Flags(), "", "none")
        sol = S.Solver("gauss-seidel", device=0)
        try:
            sol.handle.set_option("sweep_ell", ell)
            S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
            S.solve(args, sol)
            out.append((args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy()))
        finally:
            sol.handle.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    orc.set_threads(1)
    ref = orc.solve(rp, col, val, np.ones(n), np.ones(n), orc.GAUSS_SEIDEL, max_iters=40, tol=1e-13)
    assert np.array_equal(out[0][1], ref["x_star"])
