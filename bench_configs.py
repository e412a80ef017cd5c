#!/usr/bin/env python3
"""bench_configs.py — the other BASELINE.json configs on one MI355X (parity-test cases measured
for DESIGN.md; the driver's headline line comes from bench.py):

  C3  restarted GMRES(50), nonsymmetric 27-pt convection-diffusion 128^3 (MGS Arnoldi on device)
  C4  power-law irregular matrix, 10 M rows: SpMV scalar-CSR (exact) vs vector-CSR vs SELL-64-sigma,
      Jacobi (correct sweep) and Gauss-Seidel (multicolour sweep)

  python bench_configs.py [c3] [c4] [--small] [--no-cpu]   (--small: 48^3 / 200k rows, for a quick check;
                                                            --no-cpu: skip the oracle legs, for profiler passes)
One JSON line per measurement.  No torch in this process (see iterative_solvers_amd/rendezvous.py).
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from iterative_solvers_amd import capi, matgen, solver as S  # noqa: E402

HBM = 8000.0


def spmv_bytes(n, nnz):
    return 12 * nnz + 4 * (n + 1) + 16 * n


def emit(**kw):
    print(json.dumps(kw), flush=True)


def time_trips(h, it0, trips):
    h.synchronize()
    t0 = time.perf_counter()
    it = it0
    for _ in range(trips):
        h.iterate(it); it += 1
    h.synchronize()
    return (time.perf_counter() - t0) / trips, it


def spmv_formats(h, n, nnz, tag):
    xd = h.to_dev(np.random.default_rng(0).uniform(-1, 1, n)); yd = h.dev_alloc(8 * n)
    for nm, f in (("exact", capi.FMT_CSR_EXACT), ("vector", capi.FMT_CSR_VECTOR), ("sell", capi.FMT_SELL)):
        h.spmv_timed(xd, yd, f, reps=3)
        ms = h.spmv_timed(xd, yd, f, reps=20)
        info = h.plan_info()
        by = spmv_bytes(n, nnz)
        emit(config=tag, what=f"spmv_{nm}", ms=ms, GBps_algorithmic=by / ms / 1e6, frac_hbm_peak=by / ms / 1e6 / HBM,
             lanes_per_row=info["lanes_per_row"], sell_occupancy=info["sell_occupancy"])
    h.dev_free(xd); h.dev_free(yd)


NO_CPU = "--no-cpu" in sys.argv


def cpu_trips(rp, col, val, solver, trips, **kw):
    if NO_CPU:
        return None, 0
    import oracle_py as orc
    n = len(rp) - 1
    th = min(orc.num_threads(), os.cpu_count() or 1)
    orc.set_threads(th)
    r = orc.solve(rp, col, val, np.ones(n), np.ones(n), solver, fixed_iters=trips, max_iters=trips, **kw)
    return trips / r["seconds"], th


def c3(small):
    N = 48 if small else 128
    rp, col, val = matgen.convdiff3d_27pt(N)
    n, nnz = len(rp) - 1, len(col)
    args = S.Args("gmres", S.LoopParams(max_iters=10**9, tol=0.0, gmres_restart_len=50))
    sol = S.Solver("gmres", device=0)
    S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
    h = sol.handle
    it = 0
    res = None
    # one full restart cycle (50 trips, inner index k = 0..49) as warm-up, two as the timed region
    for _ in range(50):
        res = h.iterate(it)
        if (it + 1) % 50 == 0:
            h.gmres_restart(it)
        it += 1
    h.synchronize()
    t0 = time.perf_counter()
    trips = 100
    for _ in range(trips):
        res = h.iterate(it)
        if (it + 1) % 50 == 0:
            h.gmres_restart(it)
        it += 1
    h.synchronize()
    dt = (time.perf_counter() - t0) / trips
    # algorithmic bytes per trip, mean over k = 0..49: SpMV + (5(k+1)+3) * 8n
    by = spmv_bytes(n, nnz) + np.mean([(5 * (k + 1) + 3) * 8 * n for k in range(50)])
    # fused MGS steps move 4 passes per j instead of 5: bytes actually requested
    by_fused = spmv_bytes(n, nnz) + np.mean([(4 * (k + 1) + 4) * 8 * n for k in range(50)])
    import oracle_py as orc
    cpu, th = cpu_trips(rp, col, val, orc.GMRES, 50, restart_len=50)
    emit(config="C3 GMRES(50) convdiff 27-pt %d^3" % N, n=n, nnz=nnz, ms_per_trip=dt * 1e3, trips_per_s=1 / dt,
         GBps_algorithmic=by / dt / 1e9, frac_hbm_peak=by / dt / 1e9 / HBM, GBps_fused_bytes=by_fused / dt / 1e9,
         last_residual=res, cpu_trips_per_s=cpu, cpu_threads=th)
    sol.handle.close()


def c4(small):
    n = 200_000 if small else 10_000_000
    t0 = time.time()
    rp, col, val = matgen.powerlaw(n)
    nnz = len(col)
    emit(config="C4 power-law", n=n, nnz=nnz, mean_row=nnz / n, max_row=int(np.diff(rp).max()), gen_seconds=time.time() - t0)
    h = capi.Handle(0)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_SELL)
    spmv_formats(h, n, nnz, "C4 power-law %d rows" % n)
    h.close()
    import oracle_py as orc
    b = np.ones(n)
    for name, kw, okw in (("jacobi", dict(jacobi_mode=capi.JACOBI_REF), dict(jacobi_mode=1)),
                          ("gauss-seidel", dict(gs_order=capi.GS_MULTICOLOR), {})):
        args = S.Args(name, S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver(name, device=0, **kw)
        t0 = time.time()
        S.preprocessing_crs(args, sol, rp, col, val, b, b.copy())
        setup = time.time() - t0
        dt, it = time_trips(sol.handle, 0, 5)
        dt, it = time_trips(sol.handle, it, 30)
        res = sol.handle.iterate(it)
        cpu, th = cpu_trips(rp, col, val, orc.JACOBI if name == "jacobi" else orc.GAUSS_SEIDEL, 10, **okw)
        # trip = sweep + recorded residual (second SpMV), as in the reference loop
        by = 2 * spmv_bytes(n, nnz) + (16 * n if name == "jacobi" else 40 * n)
        emit(config="C4 power-law %d rows" % n, what=name + (" (multicolour, %d colours)" % sol.n_colors if sol.n_colors else " (correct sweep)"),
             ms_per_trip=dt * 1e3, trips_per_s=1 / dt, GBps_algorithmic=by / dt / 1e9, frac_hbm_peak=by / dt / 1e9 / HBM,
             residual=res, setup_seconds=setup, cpu_trips_per_s=cpu, cpu_threads=th)
        sol.handle.close()


def f4(small):
    """Opt-in communication-reducing variants on one GPU: what the extra vector traffic of
    pipelined CG and the batched CGS-2 orthogonalisation cost per trip when there is no all-reduce to save."""
    N = 64 if small else 256
    rp, col, val = matgen.laplace3d_7pt(N)
    n = len(rp) - 1
    for variant in (0, 1):
        args = S.Args("conjugate-gradient", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("conjugate-gradient", device=0)
        sol.handle.set_option("cg_variant", variant)
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        dt, it = time_trips(sol.handle, 0, 20)
        dt, it = time_trips(sol.handle, it, 200)
        emit(config="CG 7-pt %d^3" % N, what="pipelined CG (cg_variant 1)" if variant else "reference CG", ms_per_trip=dt * 1e3,
             reductions_per_trip=1 if variant else 2)
        sol.handle.close()
    del rp, col, val
    N = 48 if small else 128
    rp, col, val = matgen.convdiff3d_27pt(N)
    n = len(rp) - 1
    for variant in (0, 1):
        args = S.Args("gmres", S.LoopParams(max_iters=10**9, tol=0.0, gmres_restart_len=50))
        sol = S.Solver("gmres", device=0)
        sol.handle.set_option("gmres_ortho", variant)
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        h = sol.handle
        it = 0
        for phase in range(2):
            h.synchronize(); t0 = time.perf_counter()
            for _ in range(50 if phase == 0 else 100):
                h.iterate(it)
                if (it + 1) % 50 == 0:
                    h.gmres_restart(it)
                it += 1
            h.synchronize(); dt = (time.perf_counter() - t0) / 100
        emit(config="GMRES(50) 27-pt %d^3" % N, what="CGS-2 (gmres_ortho 1)" if variant else "modified Gram-Schmidt (reference)",
             ms_per_trip=dt * 1e3, reductions_per_trip="3" if variant else "k + 2 (mean 27.5)")
        h.close()


if __name__ == "__main__":
    small = "--small" in sys.argv
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c3", "c4"]
    if "c3" in which:
        c3(small)
    if "c4" in which:
        c4(small)
    if "f4" in which:
        f4(small)
