"""GPU parity of the persistent Gram-Schmidt launch of GMRES (csrc/its_mgs.hip, option "gmres_persist") through the
C ABI: one launch per trip that keeps w in registers instead of one launch per basis vector.

Same operations per element as the per-step kernels; the dot products are summed in another fixed order, so the
Hessenberg entries agree to rounding, not bit for bit.  Bar: the one every GMRES test has (test_gpu_solvers.py:
history within 1e-10 relative / the oracle's own reduction-order noise envelope, count +-1 %, x_star 1e-9).
The launch is the default for every vector length up to 2.9 M entries (the grid shrinks with the vectors: 8 workgroups
for the small matrices the oracle finishes in seconds, one per CU at 128^3, test_gpu_fullsize.py); "gmres_persist" 0 or
a larger "gmres_persist_min_n" selects one launch per basis vector.
"""
import numpy as np
import pytest

from helpers import MATRICES
from iterative_solvers_amd import capi, matgen, solver as S
from test_gpu_solvers import check_history, run_both

pytestmark = pytest.mark.gpu

FORCE = {"gmres_persist_min_n": 1}


@pytest.mark.parametrize("name,restart", [("lap2d_16", 100), ("lap2d_16", 10), ("lap2d_33", 30), ("cd27_12", 50),
                                           ("cd27_12", 10), ("lap3d_16", 50), ("lap2d_64", 60), ("pl_200", 10)])
def test_gmres_persistent_launch_vs_oracle(name, restart):
    info = {}
    ref, args, xs = run_both(name, "-gm", "none", restart=restart, max_iters=500, options=FORCE, info=info)
    assert info["persist_grid"] > 0 and info["fallbacks"] == 0, info      # the launch ran, every trip
    check_history(ref, args)
    assert np.max(np.abs(xs - ref["x_star"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["x_star"])))


@pytest.mark.parametrize("name", ["lap2d_16", "cd27_6"])
def test_gmres_gs_preconditioned_persistent_launch(name):
    info = {}
    ref, args, xs = run_both(name, "-gm", "diag", precond="gs", restart=10, max_iters=300, options=FORCE, info=info)
    assert info["persist_grid"] > 0 and info["fallbacks"] == 0, info
    check_history(ref, args)


def _solve(rp, col, val, restart, trips, options):
    n = len(rp) - 1
    args = S.Args("gmres", S.LoopParams(max_iters=trips, tol=1e-30, gmres_restart_len=restart), S.Flags(), "", "none")
    sol = S.Solver("gmres", device=0)
    try:
        for key, value in options.items():
            sol.handle.set_option(key, value)
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.zeros(n))
        S.solve(args, sol)
        return args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy(), sol.handle.gmres_info()
    finally:
        sol.handle.close()


@pytest.mark.parametrize("gen,restart", [("lap3d_40", 20), ("cd27_33", 12), ("lap2d_301", 25)])
def test_persistent_launch_agrees_with_per_step_kernels(gen, restart):
    """Both forms of the sweep on the same matrix (n = 64 000, 35 937 — odd —, 90 601 — odd): histories agree to
    1e-11 relative over 2.5 restart cycles, x_star to 1e-11; the default is the persistent launch, and a threshold
    above the vector length ("gmres_persist_min_n") is the per-step kernels."""
    rp, col, val = {"lap3d_40": lambda: matgen.laplace3d_7pt(40), "cd27_33": lambda: matgen.convdiff3d_27pt(33),
                    "lap2d_301": lambda: matgen.laplace2d_5pt(301)}[gen]()
    trips = restart * 5 // 2
    h1, x1, i1 = _solve(rp, col, val, restart, trips, FORCE)
    h0, x0, i0 = _solve(rp, col, val, restart, trips, {"gmres_persist": 0})
    hd, xd, idf = _solve(rp, col, val, restart, trips, {})
    hm, xm, im = _solve(rp, col, val, restart, trips, {"gmres_persist_min_n": 10**9})
    assert i1["persist_grid"] > 0 and i1["fallbacks"] == 0 and i0["persist_grid"] == 0 and im["persist_grid"] == 0
    assert idf["persist_grid"] == i1["persist_grid"] and idf["persist_grid"] % 8 == 0
    assert len(h1) == len(h0) > restart
    assert np.array_equal(h1, hd) and np.array_equal(x1, xd) and np.array_equal(h0, hm) and np.array_equal(x0, xm)
    assert np.max(np.abs(h1 - h0) / h0) <= 1e-11, np.max(np.abs(h1 - h0) / h0)
    assert np.max(np.abs(x1 - x0)) <= 1e-11 * np.max(np.abs(x0))


@pytest.mark.parametrize("failures,kind", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_launch_that_gives_up_is_repeated_by_the_per_step_kernels(failures, kind):
    """kind 1: workgroup 0 of the first `failures` launches never publishes (test hook), the other workgroups run into
    their limit.  kind 2: a FOLLOWER gives up at the last step of the sweep — where no later step exists in
    which workgroup 0 would miss it — by claiming its copy of the published sum; workgroup 0's publication of that step
    finds the word taken and leaves H[k+1,k] unwritten.  Either way the launch ends having written nothing that counts
    and the trip is redone with one launch per basis vector: those trips carry exactly the per-step path's bits; after
    two such trips the handle stays on that path."""
    rp, col, val = matgen.convdiff3d_27pt(16)
    restart, trips = 8, 12
    h1, x1, i1 = _solve(rp, col, val, restart, trips, dict(FORCE, gmres_persist_inject_kind=kind, gmres_persist_inject_fail=failures))
    h0, x0, i0 = _solve(rp, col, val, restart, trips, {"gmres_persist": 0})
    hp, xp, ip = _solve(rp, col, val, restart, trips, FORCE)
    assert i1["fallbacks"] == failures and i1["persist_grid"] == (0 if failures >= 2 else ip["persist_grid"])
    assert np.array_equal(h1[:failures], h0[:failures])
    if failures >= 2:
        assert np.array_equal(h1, h0) and np.array_equal(x1, x0)
    else:
        assert np.max(np.abs(h1 - hp) / hp) <= 1e-11 and np.max(np.abs(x1 - xp)) <= 1e-11 * np.max(np.abs(xp))


def test_persistent_launch_is_deterministic():
    rp, col, val = matgen.convdiff3d_27pt(20)
    a = _solve(rp, col, val, 15, 40, FORCE)
    b = _solve(rp, col, val, 15, 40, FORCE)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_persistent_launch_not_used_where_it_does_not_apply():
    """CGS-2 (its own kernels) and vectors beyond the registers of one resident grid stay on the per-step path."""
    rp, col, val = MATRICES["lap3d_16"]()
    _, _, info = _solve(rp, col, val, 10, 12, dict(FORCE, gmres_ortho=1))
    assert info["persist_grid"] == 0
    assert capi.lib().its_version() is not None
