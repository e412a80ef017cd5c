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


def test_gauss_seidel_natural_order_256cubed_iterate(lap256):
    """The reference's default sweep order at configs[1]'s size: 256 tiles of the line-pipelined sweep (csrc/its_lines.hip)
    resident at once, every tile edge polled across XCDs.  Three trips from a random start with a random right-hand side:
    the iterate == the oracle's sequential sweeps (kernels.hpp:451-473, gauss_seidel.hpp:227) bit for bit, and == the level
    schedule's; recorded norms 1e-10."""
    rp, col, val = lap256
    n = len(rp) - 1
    rng = np.random.default_rng(21)
    b = rng.uniform(-1, 1, n); x0 = rng.uniform(-1, 1, n)
    trips = 3
    orc.set_threads(1)
    ref = orc.solve(rp, col, val, b, x0, orc.GAUSS_SEIDEL, max_iters=trips, tol=1e-300)
    got = {}
    for lines in (1, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=trips, tol=1e-300), S.Flags(), "", "none")
        sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_LEVEL)
        try:
            sol.handle.set_option("sweep_lines", lines)
            S.preprocessing_crs(args, sol, rp, col, val, b, x0)
            info = sol.handle.sweep_info()
            assert (info["workgroups"] == 256) == bool(lines)
            S.solve(args, sol)
            got[lines] = (args.normed_residuals[:args.loop_params.residual_count].copy(), sol.x_star.copy())
        finally:
            sol.handle.close()
    assert np.array_equal(got[1][1], ref["x_star"])
    assert np.array_equal(got[1][1], got[0][1]) and np.array_equal(got[1][0], got[0][0])
    k = min(len(got[1][0]), len(ref["history"]))
    assert k >= trips and np.max(np.abs(got[1][0][:k] - ref["history"][:k])) <= 1e-10 * ref["r0_two"]


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


def test_sell_image_beyond_2_31_stored_elements(gpu_handle, monkeypatch):
    """A padded SELL-64 image of 2.4e9 stored elements (one row of 600 entries in every chunk of 64, sorting off): the
    chunk pointers are 64-bit and the col / val arrays are filled on the device, so neither the int32 limit nor a host
    copy of the image stands in the way (round 2: ITS_EUNSUPPORTED).  29 GB of device memory; against the oracle at 1e-12
    of sum |a_ij x_j| (reordered summation)."""
    monkeypatch.setenv("ITS_SELL_SIGMA", "1")
    n_chunks, long_len = 1 << 16, 600
    n = n_chunks * 64
    lens = np.ones(n, np.int64); lens[::64] = long_len
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    rng = np.random.default_rng(12)
    col = np.empty(nnz, np.int32)
    rows = np.repeat(np.arange(n), lens)
    first = np.repeat(rp[:-1].astype(np.int64), lens)
    # row i: its diagonal, and for the long rows 599 further columns i + 1 + 7 k (mod n), ascending within the wrap
    k = np.arange(nnz, dtype=np.int64) - first
    col[:] = ((rows + np.where(k > 0, 1 + 7 * (k - 1), 0)) % n).astype(np.int32)
    val = rng.uniform(-1, 1, nnz)
    x = rng.uniform(-1, 1, n)
    assert n_chunks * 64 * long_len > 2**31
    orc.set_threads(min(16, orc.num_threads()))
    ref = orc.spmv(rp, col, val, x)
    absrow = orc.spmv(rp, col, np.abs(val), np.abs(x))
    orc.set_threads(1)
    gpu_handle.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_SELL)
    info = gpu_handle.plan_info(capi.MAT_A, capi.FMT_SELL)
    assert 0 < info["sell_occupancy"] < nnz / 2.0**31
    got = gpu_handle.spmv_host(x, n, capi.MAT_A, capi.FMT_SELL)
    assert np.all(np.abs(got - ref) <= 1e-12 * absrow)


def _envelope(want, alt, got, tol_rel=1e-10):
    """the 1e-10 bar, widened to 10x the oracle's own 1-vs-8-thread deviation (running maximum, 8 trips ahead)"""
    k = min(len(want), len(alt), len(got))
    w, a, g = np.asarray(want[:k]), np.asarray(alt[:k]), np.asarray(got[:k])
    noise = np.maximum.accumulate(np.abs(a - w))
    noise = np.concatenate([noise[8:], np.full(min(8, k), noise[-1])])
    bound = np.maximum(tol_rel * np.maximum(w, 1e-2 * w[0]), 10.0 * noise)
    assert np.all(np.abs(g - w) <= bound), (np.abs(g - w) / w, bound / w)
    return k


def test_gmres50_27pt_128cubed_history():
    """configs[2]: restarted GMRES(50) on the 27-pt convection-diffusion matrix at 128^3 (the wide ELL image), 60 trips
    = one full cycle, the restart and ten trips of the next cycle, against the oracle."""
    rp, col, val = matgen.convdiff3d_27pt(128)
    n = len(rp) - 1
    b = np.ones(n)
    trips = 60
    kw = dict(fixed_iters=trips, max_iters=trips, restart_len=50)
    orc.set_threads(1)
    want = orc.solve(rp, col, val, b, b.copy(), orc.GMRES, **kw)["all_values"]
    orc.set_threads(8)
    alt = orc.solve(rp, col, val, b, b.copy(), orc.GMRES, **kw)["all_values"]
    orc.set_threads(1)
    args = S.Args("gmres", S.LoopParams(max_iters=10**9, tol=0.0, gmres_restart_len=50))
    sol = S.Solver("gmres", device=0)
    try:
        S.preprocessing_crs(args, sol, rp, col, val, b, b.copy())
        assert sol.handle.spmv_kernel_info(capi.MAT_A)[0] == "spmv_ellw_kernel"
        assert sol.handle.gmres_info()["persist_grid"] > 0           # the persistent Gram-Schmidt launch (its_mgs.hip)
        got = []
        for it in range(trips):
            got.append(sol.handle.iterate(it))
            if (it + 1) % 50 == 0:
                sol.restart_gmres(it)
        assert sol.handle.gmres_info()["fallbacks"] == 0
    finally:
        sol.handle.close()
    # (the oracle's own 1-vs-8-thread histories differ by 1e-11 at trip 0 and 1e-9 by trip 10 here: 2 M-term
    # sequential sums feeding the Arnoldi recurrences; the envelope is the whole statement)
    assert _envelope(want, alt, got) >= trips - 1
    assert np.all(np.abs(np.array(got[:3]) - want[:3]) <= 1e-10 * want[:3])


@pytest.fixture(scope="module")
def powerlaw2m():
    return matgen.powerlaw(2_200_000)       # >= 2^21 columns: the panel-blocked image is built (csrc/its_panel.hip)


def test_jacobi_powerlaw_2m_history_and_iterate(powerlaw2m):
    """configs[3], Jacobi leg at 2.2 M rows (longest row 8192: the panel-blocked kernels, one fused pass over A per trip):
    the iterate is the per-row in-order chain of jacobi_iteration_ref_cpu, so x after 12 trips equals the oracle's bit
    for bit; the recorded norms are tree sums against sequential ones: 1e-10."""
    rp, col, val = powerlaw2m
    n = len(rp) - 1
    b = np.ones(n)
    trips = 12
    orc.set_threads(16)
    x0 = np.zeros(n)                       # (x = 1 solves this system exactly: rows sum to 1)
    ref = orc.solve(rp, col, val, b, x0.copy(), orc.JACOBI, jacobi_mode=1, fixed_iters=trips, max_iters=trips)
    orc.set_threads(1)
    args = S.Args("jacobi", S.LoopParams(max_iters=10**9, tol=0.0))
    sol = S.Solver("jacobi", device=0, jacobi_mode=1)
    try:
        S.preprocessing_crs(args, sol, rp, col, val, b, x0.copy())
        assert sol.handle.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel"
        got = np.array([sol.handle.iterate(it) for it in range(trips)])
        xs = sol.save_x_star(trips).copy()
    finally:
        sol.handle.close()
    want = ref["all_values"][:trips]
    assert np.all(np.abs(got - want) <= 1e-10 * want), np.abs(got - want) / want
    assert np.array_equal(xs, ref["x_star"])


def test_gauss_seidel_multicolor_powerlaw_2m(powerlaw2m):
    """configs[3], GS leg: multicolour sweep (hub rows coloured first) = the oracle's natural-order sweep of the
    permuted matrix, row for row the same chain: identical iterate, norms to 1e-10."""
    rp, col, val = powerlaw2m
    n = len(rp) - 1
    b = np.ones(n)
    trips = 8
    perm, nc = capi.multicolor_order(rp, col)
    rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
    orc.set_threads(16)
    ref = orc.solve(rpp, cp, vp, b, b.copy(), orc.GAUSS_SEIDEL, fixed_iters=trips, max_iters=trips)
    orc.set_threads(1)
    args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
    sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_MULTICOLOR)
    try:
        S.preprocessing_crs(args, sol, rp, col, val, b, b.copy())
        assert sol.n_colors == nc and nc <= 32
        assert sol.handle.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel"
        got = np.array([sol.handle.iterate(it) for it in range(trips)])
        xs = sol.save_x_star(trips).copy()
    finally:
        sol.handle.close()
    want = ref["all_values"][:trips]
    assert np.all(np.abs(got - want) <= 1e-10 * want), np.abs(got - want) / want
    x_ref = np.empty(n); x_ref[perm] = ref["x_star"]
    assert np.array_equal(xs, x_ref)


def test_spmv_powerlaw_2m_panel_image_bit_exact(gpu_handle, powerlaw2m):
    """the plain product, the CG form and the residual form on the panel image at its real sizes (2^16-column panels,
    4096-row tiles, 128-entry cut): bit-exact against the oracle and the irregular-row kernel"""
    rp, col, val = powerlaw2m
    n = len(rp) - 1
    h = gpu_handle
    x = np.random.default_rng(10).uniform(-1, 1, n)
    orc.set_threads(min(16, orc.num_threads()))
    ref = orc.spmv(rp, col, val, x)
    orc.set_threads(1)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    assert h.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel"
    assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref)
    px = h.to_dev(x); py = h.dev_alloc(8 * n + 64)
    try:
        outs = {}
        for pan in (1, 0):
            h.set_option("spmv_panel", pan)
            for mode in (1, 2):
                h.spmv_timed(px, py, capi.FMT_CSR_EXACT, reps=1, with_dot=mode)
                outs[(pan, mode)] = h.from_dev(py, n)
        assert np.array_equal(outs[(1, 1)], ref) and np.array_equal(outs[(0, 1)], ref)
        assert np.array_equal(outs[(1, 2)], x - ref) and np.array_equal(outs[(0, 2)], x - ref)
    finally:
        h.set_option("spmv_panel", 1)
        h.dev_free(px); h.dev_free(py)


def test_zz_powerlaw_10m_the_config_itself():
    """BASELINE configs[3] at ITS size — the 10 M-row power-law matrix bench.py times (84 M entries, longest row 8192) —
    checked, not only timed: exact SpMV == oracle; Jacobi (correct sweep, one fused pass over A per
    trip) 3 trips and multicolour Gauss-Seidel 2 trips: iterate == oracle bit for bit (per-row in-order chains), recorded
    norms to 1e-10; kernel names asserted.  Last in this file: it holds 3 GB of host arrays and takes about half a minute."""
    rp, col, val = matgen.powerlaw(10_000_000)
    n = len(rp) - 1
    assert n == 10_000_000 and int(np.diff(rp).max()) == 8192
    b = np.ones(n)
    x = np.random.default_rng(21).uniform(-1, 1, n)
    orc.set_threads(min(16, orc.num_threads()))
    try:
        ref_y = orc.spmv(rp, col, val, x)
        h = capi.Handle(0)
        try:
            h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
            assert h.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel"
            assert np.array_equal(h.spmv_host(x, n, capi.MAT_A, capi.FMT_CSR_EXACT), ref_y)
        finally:
            h.close()
        del ref_y
        # ---- Jacobi, 3 trips from x0 = 0
        trips = 3
        ref = orc.solve(rp, col, val, b, np.zeros(n), orc.JACOBI, jacobi_mode=1, fixed_iters=trips, max_iters=trips)
        args = S.Args("jacobi", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("jacobi", device=0, jacobi_mode=1)
        try:
            S.preprocessing_crs(args, sol, rp, col, val, b, np.zeros(n))
            assert sol.handle.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel"
            got = np.array([sol.handle.iterate(it) for it in range(trips)])
            xs = sol.save_x_star(trips).copy()
        finally:
            sol.handle.close()
        want = ref["all_values"][:trips]
        assert np.all(np.abs(got - want) <= 1e-10 * want), np.abs(got - want) / want
        assert np.array_equal(xs, ref["x_star"])
        del ref, xs
        # ---- multicolour Gauss-Seidel, 2 trips: the oracle's natural-order sweep of the permuted matrix
        trips = 2
        perm, nc = capi.multicolor_order(rp, col)
        rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
        ref = orc.solve(rpp, cp, vp, b, b.copy(), orc.GAUSS_SEIDEL, fixed_iters=trips, max_iters=trips)
        del rpp, cp, vp
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_MULTICOLOR)
        try:
            S.preprocessing_crs(args, sol, rp, col, val, b, b.copy())
            assert sol.n_colors == nc and nc <= 32
            assert sol.handle.spmv_kernel_info(capi.MAT_A)[0] == "spmv_panel_kernel"
            got = np.array([sol.handle.iterate(it) for it in range(trips)])
            xs = sol.save_x_star(trips).copy()
        finally:
            sol.handle.close()
        want = ref["all_values"][:trips]
        assert np.all(np.abs(got - want) <= 1e-10 * want), np.abs(got - want) / want
        x_ref = np.empty(n); x_ref[perm] = ref["x_star"]
        assert np.array_equal(xs, x_ref)
    finally:
        orc.set_threads(1)
