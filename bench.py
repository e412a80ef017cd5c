#!/usr/bin/env python3
"""bench.py — headline benchmark of the solver hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 256] [--global-grid G] [--format exact|vector|sell]

Workload (BASELINE.json configs[1], SURVEY.md §8d "C2"): conjugate gradient, fp64, 3-D 7-point
Laplacian, 16.8 M rows PER GPU, b = 1, x0 = 1, no scaling.  Two named grids:
  * weak series (default at N = 1, 2, 4): every rank owns a 256^3 slab of one 256 x 256 x 256N grid;
  * configs[4] (default at N = 8; `--global-grid 512` at any N that divides 512): the ranks share ONE 512^3 grid
    (134 M rows), rank r owns planes [512 r / N, 512 (r + 1) / N) — at N = 8 that is 64 planes of 512^2 per GPU, the
    same 16.8 M rows as a 256^3 slab, but halo planes of 2 MiB and a far band 262 144 rows away.
    `--global-grid 0` forces the weak series.
One "step" = one CG iteration through the reference's loop
body (Solver::iterate + record_residual_norm + exchange_arrays, solver_harness.hpp:40-116),
INCLUDING the per-trip 8-byte residual read-back the harness needs for its stopping test.
The matrix and all vectors are resident in HBM before the timed region starts.
Timing: W warm-up trips, then REGIONS (5) regions of exactly K trips, each bracketed by a barrier and a device
synchronize on both sides, each taken as the MAX over ranks; `value` / `ms_per_step` are the MEDIAN region,
`ms_per_step_min` / `_max` the fastest and slowest.

Prints ONE JSON line (rank 0): value = CG iterations/s x (number of slabs) — whole-job
throughput in slab-iterations per second, so weak scaling is value(N) / (N value(1)).
`roofline` prices the dominant kernel (the CSR SpMV fused with the p.Ap partial sums) against
the ~8 TB/s HBM peak; `cpu_baseline` times the oracle (CPU restatement of the reference loop,
arithmetic pinned to the compiled reference) on the box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def spmv_bytes(n, nnz):
    """Algorithmic bytes of one CSR SpMV: 12 nnz + 4 (n+1) + 8 n (x) + 8 n (y)."""
    return 12 * nnz + 4 * (n + 1) + 16 * n


def cg_iter_bytes(n, nnz):
    """SURVEY.md §8d's accounting of one CG trip: SpMV + 9 vector passes (x,p->x; r,Ap->r with fused dot; r,p->p) =
    SpMV + 72 n.  (The implementation moves 64 n: `cg_GBps_streamed` in the JSON line prices that.)"""
    return spmv_bytes(n, nnz) + 72 * n


def kernel_sources_hash():
    """sha256 over the HIP sources of the library: what a PMC summary must have been taken with to describe the
    kernels this run launches (tests/profile_run.sh stamps it into the summary it writes)."""
    import glob
    import hashlib
    hsh = hashlib.sha256()
    # the files that hold device code (host-side API changes do not invalidate a counter profile)
    names = ("its_kernels.hip", "its_ell.hip", "its_ellg.hip", "its_ellm.hip", "its_lines.hip", "its_panel.hip", "its_mgs.hip", "its_epilogue.h", "its_device.h", "its_kernels.h", "its_ap.h")
    for f in sorted(os.path.join(ROOT, "iterative_solvers_amd", "csrc", nm) for nm in names):
        hsh.update(os.path.basename(f).encode()); hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def pmc_traffic(kernel_substr):
    """(bytes per launch, source note) of the kernel from the newest committed PMC summary (profiles/*pmc_summary.txt,
    produced by tests/profile_run.sh: separate FETCH_SIZE / WRITE_SIZE passes).  gfx950 correction from
    MI355X_MICROARCH.md: FETCH_SIZE (KB) counts half the bytes of wide coalesced reads -> doubled; WRITE_SIZE (KB)
    is exact.  The summary carries the hash of the kernel sources it was taken with: when the sources here differ
    (or the summary has no stamp, or does not list the kernel) the traffic is None and the note says why — a
    counter value from other code is never reported as this run's."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.txt")))
    if not files:
        return None, "no profiles/*pmc_summary.txt"
    # the newest summary taken with THESE kernel sources that lists the kernel (several summaries share a round's prefix:
    # the headline run's, the configs', the 512-slab rehearsal's), else the newest by name for the note
    want = kernel_sources_hash()
    def stamp_of(f):
        for line in open(f):
            if line.startswith("# kernel_sources_sha256="):
                return line.split("=", 1)[1].strip()
        return None
    match = [f for f in files if stamp_of(f) == want and any(kernel_substr in l and l.startswith("FETCH_SIZE") for l in open(f))]
    if match:
        files = match
    fetch = write = None
    stamp = None
    for line in open(files[-1]):
        if line.startswith("# kernel_sources_sha256="):
            stamp = line.split("=", 1)[1].strip()
        if kernel_substr in line and "avg=" in line:
            # the instantiation the CG loop launches: plain store epilogue fused with the p.Ap partial sums
            # (spmv_ell_kernel<W, 0, true>, spmv_stream4_kernel<0, true, V>), not the Jacobi / residual forms
            targs = line.split(kernel_substr, 1)[1].split(">", 1)[0].replace(" ", "").split(",")
            if "true" not in targs or "0" not in targs:
                continue
            kb = float(line.split("avg=")[1].split("KB")[0])
            if line.startswith("FETCH_SIZE"):
                fetch = kb
            elif line.startswith("WRITE_SIZE"):
                write = kb
    name = os.path.relpath(files[-1], ROOT)
    if stamp != kernel_sources_hash():
        return None, f"{name} was taken with other kernel sources ({stamp}); re-run tests/profile_run.sh"
    if fetch is None or write is None:
        return None, f"{name} does not list {kernel_substr}"
    return 2.0 * fetch * 1024 + write * 1024, f"{name} (kernel sources {stamp}; 2 x FETCH_SIZE + WRITE_SIZE per launch)"


def cpu_baseline(rp, col, val, n, budget_s=10.0):
    """Oracle CG on the host cores, bounded sample (kind 'port': our restatement of the loop)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as orc
    threads = int(os.environ.get("ITS_CPU_THREADS", "0")) or min(orc.num_threads(), os.cpu_count() or 1)
    orc.set_threads(threads)
    b = np.ones(n); x0 = np.ones(n)
    probe = orc.solve(rp, col, val, b, x0, orc.CG, fixed_iters=2, max_iters=2)
    per_it = max(probe["seconds"] / 2, 1e-6)
    iters = int(max(3, min(200, budget_s / per_it)))
    res = orc.solve(rp, col, val, b, x0, orc.CG, fixed_iters=iters, max_iters=iters)
    out = {"value": iters / res["seconds"], "unit": "it/s", "cores": threads, "kind": "port",
           "sample": f"{iters} CG iterations (reference loop as written: 1 SpMV + 15 vector passes) "
                     f"on the same {n}-row system, oracle/libits_oracle.so with {threads} OpenMP threads"}
    try:
        out["calibration"] = cpu_calibration(orc, threads)
    except Exception as e:                                   # the calibration is an extra: never fails the bench line
        out["calibration"] = {"error": repr(e)[:200]}
    return out


def cpu_calibration(orc, threads, grid=128):
    """`kind` is "port": what is timed above is the oracle's restatement of the loop, not the reference binary.  This leg
    shows that the port runs at the reference's speed: the COMPILED REFERENCE (oracle/_ref/iterative_solvers_cpu_m100, built
    from /root/reference by oracle/Makefile with plain g++, unmodified) and the port solve the same generated 7-pt
    Laplacian `grid`^3 .mtx with `-cg -scale none` to the reference's compiled-in TOL = 1e-14, same thread count; the
    reference's own timers (solver_harness.hpp:41,52 -> "CG Solver time", "Solver Harness time") are parsed."""
    import re
    import subprocess
    import tempfile
    from iterative_solvers_amd import matgen
    exe = os.path.join(ROOT, "oracle", "_ref", "iterative_solvers_cpu_m100")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/iterative_solvers_cpu_m100 not built (make -C oracle ref, needs /root/reference)"}
    rp, col, val = matgen.laplace3d_7pt(grid)
    n = len(rp) - 1
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, f"lap{grid}.mtx")
        t0 = time.time()
        orc.write_mtx(path, rp, col, val)
        t_write = time.time() - t0
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PLACES="cores", OMP_PROC_BIND="close")
        t0 = time.time()
        r = subprocess.run([exe, path, "-cg", "-scale", "none"], env=env, capture_output=True, text=True, timeout=600, cwd=td)
        t_ref = time.time() - t0
    if r.returncode != 0:
        return {"error": f"reference exited with {r.returncode}: {r.stderr[-200:]}"}
    m = re.search(r"solver converged in: (\d+) iterations", r.stdout)
    its_ref = int(m.group(1)) if m else 5000                 # MAX_ITERS the reference was compiled with
    t_cg = float(re.search(r"CG Solver time:\s+(\S+)\[s\]", r.stdout).group(1))
    t_har = float(re.search(r"Solver Harness time:\s+(\S+)\[s\]", r.stdout).group(1))
    b = np.ones(n)
    port = orc.solve(rp, col, val, b, b.copy(), orc.CG, max_iters=5000, tol=1e-14)
    ref_its, port_its = its_ref / t_har, port["iters"] / port["seconds"]
    return {"system": f"7-pt Laplacian {grid}^3 ({n} rows, {len(col)} entries) as a generated .mtx, -cg -scale none, TOL 1e-14, {threads} threads",
            "reference_iterations": its_ref, "reference_harness_seconds": t_har, "reference_cg_solver_seconds": t_cg,
            "reference_it_per_s": ref_its, "reference_it_per_s_iterate_only": its_ref / t_cg,
            "port_iterations": port["iters"], "port_seconds": port["seconds"], "port_it_per_s": port_its,
            "port_over_reference": port_its / ref_its,
            "note": "reference_it_per_s = iterations / 'Solver Harness time' (iterate + record_residual_norm + exchange_arrays, what the "
                    "port's clock covers); the reference binary's wall time here was %.1f s (%.1f s writing the .mtx apart), most of it fscanf" % (t_ref, t_write)}


GATHER_PEAK = 56.5e9      # random 8-byte gathers/s from an 80 MB table (Infinity Cache resident, not L2): measured ceiling
                          # of MI355X, independent of occupancy and unrolling (tools/gatherlab.hip, profiles/r02_gatherlab.txt).
                          # The column-panel blocked kernels (csrc/its_panel.hip) serve part of the gathers from L2 instead
                          # (200 G/s from an L2-resident table): frac_gather_peak above 1 means exactly that.


def other_configs(capi, matgen, S):
    """BASELINE.json configs[2] (restarted GMRES(50), nonsymmetric 27-pt convection-diffusion 128^3) and configs[3]
    (Jacobi + multicolour Gauss-Seidel on the 10 M-row power-law matrix) through the same loop body as the headline
    (its_iterate with the per-trip read-back), each priced with SURVEY.md §8d's algorithmic bytes against the 8 TB/s
    HBM peak; the power-law legs also against what bounds them, the random-gather rate."""
    res = {}
    # ---- C3
    t0 = time.time()
    rp, col, val = matgen.convdiff3d_27pt(128)
    n, nnz = len(rp) - 1, len(col)
    gen3 = time.time() - t0
    args = S.Args("gmres", S.LoopParams(max_iters=10**9, tol=0.0, gmres_restart_len=50))
    sol = S.Solver("gmres", device=0)
    t0 = time.time()
    S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
    setup = time.time() - t0
    h = sol.handle
    it = 0

    def cycle(trips):
        nonlocal it
        for _ in range(trips):
            h.iterate(it)
            if (it + 1) % 50 == 0:
                h.gmres_restart(it)
            it += 1
    cycle(50)                                      # one restart cycle as warm-up, two as the timed region
    h.synchronize()
    t0 = time.perf_counter()
    cycle(100)
    h.synchronize()
    dt = (time.perf_counter() - t0) / 100
    by = spmv_bytes(n, nnz) + float(np.mean([(5 * (k + 1) + 3) * 8 * n for k in range(50)]))
    info = h.gmres_info()
    kname, streamed = h.spmv_kernel_info(capi.MAT_A)
    # what the kernels are asked to move: the SpMV's storage form + y, then either one persistent Gram-Schmidt launch
    # (csrc/its_mgs.hip: w once, each basis vector once, v_{k+1} once) or k + 2 step launches of 32 n and the normalisation
    persist = info["persist_grid"] > 0
    moved = streamed + float(np.mean([((k + 1) + 2) * 8 * n if persist else (4 * (k + 2) + 2) * 8 * n for k in range(50)]))
    res["c3_gmres50_convdiff27_128"] = {
        "workload": "restarted GMRES(50), 27-pt convection-diffusion 128^3, MGS Arnoldi on device, Givens on host",
        "rows": n, "nnz": nnz, "ms_per_trip": dt * 1e3, "trips_per_s": 1 / dt,
        # the config's roofline fraction prices the bytes the kernels are ASKED FOR: SURVEY 8d prices a
        # Gram-Schmidt step at 5 vector passes, the persistent launch reads each basis vector once with w in registers
        "frac": moved / dt / 1e9 / HBM_PEAK_GBS, "moved_bytes_per_trip": moved, "GBps_moved": moved / dt / 1e9,
        "bound": "latency of the k + 2 grid-wide sums of a trip (3.3 us each), not bandwidth" if persist else "hbm",
        "survey_8d_accounting": {"algorithmic_bytes_per_trip": by, "GBps_algorithmic": by / dt / 1e9, "frac_hbm_peak": by / dt / 1e9 / HBM_PEAK_GBS,
                                 "note": "5 vector passes per Gram-Schmidt step that the persistent launch no longer makes: can exceed 1.0"},
        "gram_schmidt": f"one persistent launch per trip, {info['persist_grid']} workgroups (fallbacks: {info['fallbacks']})" if persist
                        else "one launch per basis vector",
        "spmv_kernel": kname, "setup_seconds": setup, "generate_seconds": gen3}
    h.close()
    del rp, col, val
    # ---- C4
    t0 = time.time()
    rp, col, val = matgen.powerlaw(10_000_000)
    n, nnz = len(rp) - 1, len(col)
    gen = time.time() - t0
    b = np.ones(n)
    for name, kw in (("jacobi", dict(jacobi_mode=capi.JACOBI_REF)), ("gauss-seidel", dict(gs_order=capi.GS_MULTICOLOR))):
        args = S.Args(name, S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver(name, device=0, **kw)
        t0 = time.time()
        S.preprocessing_crs(args, sol, rp, col, val, b, b.copy())
        setup = time.time() - t0
        h = sol.handle
        k = 0
        for _ in range(5):
            h.iterate(k); k += 1
        h.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            h.iterate(k); k += 1
        h.synchronize()
        dt = (time.perf_counter() - t0) / 30
        # one Jacobi trip (correct mode, sweep fused with the recorded residual) = one pass over A + b
        # = 12 nnz + 28 n; a GS trip = b - U x, the sweep over L and the recorded residual: two passes over the entries
        # (U + L, then A) + the vectors they touch.  Gathers per trip: one per entry and pass.
        by = (spmv_bytes(n, nnz) + 8 * n) if name == "jacobi" else (2 * spmv_bytes(n, nnz) + 40 * n)
        gathers = nnz if name == "jacobi" else 2 * nnz           # Jacobi: one fused pass over A; GS: U + L, then A
        res["c4_powerlaw_" + name.replace("-", "_")] = {
            "workload": f"{name} on the power-law matrix (10 M rows, {nnz} entries, longest row {int(np.diff(rp).max())})"
                        + (f", multicolour sweep with {sol.n_colors} colours" if sol.n_colors else ", correct sweep fused with the residual pass"),
            "ms_per_trip": dt * 1e3, "trips_per_s": 1 / dt, "algorithmic_bytes_per_trip": by,
            "GBps_algorithmic": by / dt / 1e9, "frac_hbm_peak": by / dt / 1e9 / HBM_PEAK_GBS,
            "bound": "random gathers of x (80 MB: Infinity Cache, not L2; the panel-blocked passes get part of them from L2)",
            "kernel_of_the_pass_over_A": h.spmv_kernel_info(capi.MAT_A)[0], "gathers_per_trip": gathers,
            "frac_gather_peak": gathers / dt / GATHER_PEAK, "setup_seconds": setup, "generate_seconds": gen}
        if name == "jacobi":
            xd = h.to_dev(np.ones(n)); yd = h.dev_alloc(8 * n)
            h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=3)
            ms = h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=20)
            res["c4_powerlaw_spmv_exact"] = {"ms": ms, "kernel": h.spmv_kernel_info(capi.MAT_A)[0], "GBps_algorithmic": spmv_bytes(n, nnz) / ms / 1e6,
                                             "frac_hbm_peak": spmv_bytes(n, nnz) / ms / 1e6 / HBM_PEAK_GBS,
                                             "frac_gather_peak": nnz / (ms * 1e-3) / GATHER_PEAK}
            # configs[3]'s "SELL-C-sigma vs vector-CSR" on the same resident matrix (CUDA_tests/spmv.cpp:264,548-552: GF/s and
            # the occupancy nnz / stored elements of the SELL image); reordered summation, 1e-12 of the exact result
            h.spmv_timed(xd, yd, capi.FMT_CSR_VECTOR, reps=3)
            ms = h.spmv_timed(xd, yd, capi.FMT_CSR_VECTOR, reps=20)
            res["c4_powerlaw_spmv_vector"] = {"ms": ms, "kernel": "spmv_vector_kernel", "lanes_per_row": h.plan_info(capi.MAT_A)["lanes_per_row"],
                                              "GBps_algorithmic": spmv_bytes(n, nnz) / ms / 1e6, "GFps": 2.0 * nnz / ms / 1e6,
                                              "frac_hbm_peak": spmv_bytes(n, nnz) / ms / 1e6 / HBM_PEAK_GBS,
                                              "frac_gather_peak": nnz / (ms * 1e-3) / GATHER_PEAK}
        h.close()
    # SELL-64-sigma image of the same matrix (its own upload: the image is built on request)
    hs = capi.Handle(0)
    t0 = time.time()
    hs.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_SELL)
    setup = time.time() - t0
    xd = hs.to_dev(np.ones(n)); yd = hs.dev_alloc(8 * n)
    hs.spmv_timed(xd, yd, capi.FMT_SELL, reps=3)
    ms = hs.spmv_timed(xd, yd, capi.FMT_SELL, reps=20)
    occ = hs.plan_info(capi.MAT_A, capi.FMT_SELL)["sell_occupancy"]
    stored = nnz / occ if occ > 0 else float("nan")
    sell_bytes = 12 * stored + 8 * (n / 64) + 16 * n              # padding counted as cost
    res["c4_powerlaw_spmv_sell"] = {"ms": ms, "kernel": "spmv_sell_kernel", "sell_occupancy": occ, "C": 64,
                                    "GBps_algorithmic": spmv_bytes(n, nnz) / ms / 1e6, "GFps": 2.0 * nnz / ms / 1e6,
                                    "frac_hbm_peak": spmv_bytes(n, nnz) / ms / 1e6 / HBM_PEAK_GBS, "sell_bytes": sell_bytes,
                                    "GBps_sell_bytes": sell_bytes / ms / 1e6, "frac_gather_peak": nnz / (ms * 1e-3) / GATHER_PEAK,
                                    "setup_seconds": setup}
    hs.close()
    return res


def spawn_ranks(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (rank r -> GPU r), each with
    the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* variables torch.distributed.run would set, relay rank 0's one
    JSON line, and fail if any rank fails.  This parent never touches the GPU (no HIP call, no library load), so
    no process that has initialised a device is ever replaced or forked."""
    import signal
    import socket
    import subprocess
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    timeout = float(os.environ.get("ITS_BENCH_SPAWN_TIMEOUT", "1500"))
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   ITS_BENCH_CHILD="1")
        env.pop("NCCL_DEBUG", None)                     # stdout carries exactly one line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    deadline = time.time() + timeout
    rc = 0
    out0 = b""
    try:
        out0, _ = procs[0].communicate(timeout=timeout)
        for p in procs:
            p.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        sys.stderr.write(f"bench.py: ranks did not finish within {timeout:.0f} s "
                         "(stuck in communicator set-up? every rank needs a GPU of its own unless ITS_BENCH_TRANSPORT=host)\n")
        rc = 124
    for r, p in enumerate(procs):
        if p.poll() is None:
            p.send_signal(signal.SIGKILL)               # exactly the processes started above
            p.wait()
        if p.returncode != 0 and rc == 0:
            sys.stderr.write(f"bench.py: rank {r} exited with status {p.returncode}\n")
            rc = p.returncode if p.returncode > 0 else 1
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, got {len(lines)}\n")
        rc = 1
    if rc == 0:
        print(lines[0], flush=True)
    raise SystemExit(rc)


REGIONS = 5                     # timed regions of --steps trips each; `value` is the median region


def timed_regions(step, steps, regions, barrier, dist):
    """`regions` regions of exactly `steps` trips, each bracketed by barrier + device synchronize on both sides and taken
    as the max over ranks; returns the region times in seconds."""
    dts = []
    for _ in range(regions):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = dist.allreduce_max(dt)
        dts.append(dt)
    return dts


def region_stats(dts, steps):
    ms = sorted(d / steps * 1e3 for d in dts)
    return {"median": ms[len(ms) // 2], "min": ms[0], "max": ms[-1]}


def generic_spmv_entries(h, capi, rp, col, val, n, nnz, xd, yd, reps):
    """north_star's ">= 80 % of HBM bandwidth" beyond the stencil dictionaries: the exact SpMV of the
    SAME C2 matrix with every dictionary form switched off — the explicit-column sliced-ELL image any matrix with rows of
    <= 8 entries gets (csrc/its_ellg.hip) — and of the same matrix with randomly permuted rows (P A: no locality left, 32-bit
    column deltas, x gathers from all over the vector).  CG form (store + fused p.Ap partial sums), same accounting as
    `roofline`.  Leaves the handle with the default image of the original matrix."""
    out = {}
    h.set_option("spmv_ell_dict", 0)
    try:
        # (ITS_BENCH_NO_PERMUTED=1: counter passes, where both matrices would launch the same kernel instantiation)
        tags = ("exact_generic",) if os.environ.get("ITS_BENCH_NO_PERMUTED", "0") == "1" else ("exact_generic", "exact_generic_rows_permuted")
        for tag in tags:
            if tag.endswith("permuted"):
                perm = np.random.default_rng(7).permutation(n)
                ln = np.diff(rp)[perm]
                rq = np.concatenate([[0], np.cumsum(ln)]).astype(np.int32)
                idx = (np.repeat(rp[:-1][perm].astype(np.int64) - rq[:-1], ln) + np.arange(nnz, dtype=np.int64))
                t0 = time.time()
                h.upload_crs(rq, col[idx], val[idx], capi.MAT_A, capi.FMT_CSR_EXACT)
                del idx, perm
            else:
                t0 = time.time()
                h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
            setup = time.time() - t0
            kname, streamed = h.spmv_kernel_info(capi.MAT_A, True)
            info = h.ell_info()
            h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=3, with_dot=True)
            ms = h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=reps, with_dot=True)
            ms_plain = h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=reps)
            by = spmv_bytes(n, nnz)
            traffic, tnote = (pmc_traffic(kname + "<") if tag == "exact_generic" else (None, "not profiled"))
            out[tag] = {"kernel": "its::" + kname, "ms": ms, "ms_plain_store": ms_plain, "algorithmic_bytes": by, "streamed_bytes": streamed,
                        "GBps_algorithmic": by / ms / 1e6, "frac": by / ms / 1e6 / HBM_PEAK_GBS, "GBps_streamed": streamed / ms / 1e6,
                        "traffic": traffic, "traffic_source": tnote, "upload_seconds": setup,
                        "image": {"slots_per_row": info["width"], "slices_with_8_16_32_bit_deltas": info["slices_by_delta_width"],
                                  "index_bytes_per_row": 4.0 * info["index_words"] / n}}
    finally:
        h.set_option("spmv_ell_dict", 1)
    h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
    return out


def gauss_seidel_entries(capi, S, rp, col, val, n, nnz):
    """Gauss-Seidel on the headline matrix, in the reference's natural order (methods/gauss_seidel.hpp:227, kernels.hpp:451-473:
    the sweep of a stencil is one line-pipelined launch, csrc/its_lines.hip) and colour by colour (GS_ORDER = multicolor: the
    same sweep on P A P^T).  A trip = b - U x, the sweep, the recorded residual.  Never `value`."""
    out = {}
    for key, order in (("natural_order", capi.GS_LEVEL), ("multicolour", capi.GS_MULTICOLOR)):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("gauss-seidel", device=0, gs_order=order)
        t0 = time.time()
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        setup = time.time() - t0
        h = sol.handle
        k = 0
        for _ in range(3):
            h.iterate(k); k += 1
        h.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            h.iterate(k); k += 1
        h.synchronize()
        dt = (time.perf_counter() - t0) / 20
        sw = h.sweep_info()
        by = 2 * spmv_bytes(n, nnz) + 40 * n                     # SURVEY.md §8d, as for configs[3]: U + L, then A, + the vectors
        out[key] = {"ms_per_trip": dt * 1e3, "trips_per_s": 1 / dt, "algorithmic_bytes_per_trip": by, "GBps_algorithmic": by / dt / 1e9,
                    "frac_hbm_peak": by / dt / 1e9 / HBM_PEAK_GBS, "dependency_levels": sw["n_levels"],
                    "sweep": (f"one launch pipelined over grid lines: {sw['workgroups']} tiles, lines of {sw['line_length']} rows in chunks of {sw['rows_per_chunk']}"
                              if sw["workgroups"] else (f"{sol.n_colors} colours, a launch each" if sol.n_colors else "a launch per level")),
                    "bound": "latency: the dependency chain of the sweep, not bandwidth" if order == capi.GS_LEVEL else "hbm",
                    "setup_seconds": setup}
        h.close()
    return out


def mixed_precision_entries(capi, S, device, rp, col, val, n, nnz, steps, reps):
    """SURVEY.md §8f.3 (the reference's USE_AP idea, config.mk:56-63), opt-in, never `value`: the SAME C2 system with its
    entries stored by magnitude — AP_THRESHOLD_1 = 2: the diagonal (6) in fp64, the off-diagonals (-1) in fp32 (ap[dp_sp])
    or fp16 (ap[dp_hp]) — on the tiered sliced-ELL image (csrc/its_ellm.hip).  x, y and the accumulation stay fp64 and the
    chain order is the reference's, so for this matrix (-1 is exact in every type) the iterates are those of the fp64 run."""
    out = {}
    for tag, mode in (("ap[dp_sp]", capi.AP_DP_SP), ("ap[dp_hp]", capi.AP_DP_HP)):
        sol = S.Solver("conjugate-gradient", device=device, spmv_format=capi.FMT_MIXED)
        h = sol.handle
        try:
            h.set_mixed_mode(mode, 2.0, 0.0)
            args = S.Args("conjugate-gradient", S.LoopParams(max_iters=10**9, tol=0.0))
            S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
            tiers = h.mixed_tiers()
            xd = h.to_dev(np.ones(n)); yd = h.dev_alloc(8 * n)
            h.spmv_timed(xd, yd, capi.FMT_MIXED, reps=3, with_dot=True)
            ms = h.spmv_timed(xd, yd, capi.FMT_MIXED, reps=reps, with_dot=True)
            h.iterate_n(0, 20)
            res = h.iterate_n(20, steps)
            streamed = tiers["value_bytes"] + 4 * n + 16 * n            # values + one index word per row + x once + y once
            out[tag] = {"threshold_1": 2.0, "form": "tiered sliced ELL" if tiers["ell_form"] else "CRS split",
                        "entries_fp64_fp32_fp16": tiers["nnz"], "matrix_value_bytes": tiers["value_bytes"],
                        "spmv_ms": ms, "streamed_bytes": streamed, "GBps_streamed": streamed / ms / 1e6,
                        "GBps_csr_equivalent": spmv_bytes(n, nnz) / ms / 1e6,
                        "cg_ms_per_step": h.last_region_ms() / steps, "last_residual": float(res[-1])}
            h.dev_free(xd); h.dev_free(yd)
        finally:
            h.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=int, default=256, help="weak series: grid edge N, N^3 rows per GPU")
    ap.add_argument("--global-grid", type=int, default=None,
                    help="G > 0: the ranks share ONE G^3 grid, rank r owns planes [G r / N, G (r + 1) / N) (BASELINE configs[4] is "
                         "G = 512 at N = 8: the default there); 0: the weak series (default at every other N)")
    ap.add_argument("--format", default="exact", choices=["exact", "vector", "sell"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 (GMRES(50) 27-pt 128^3) and C4 (power-law 10 M rows) legs")
    ap.add_argument("--no-generic", action="store_true", help="skip the spmv.exact_generic entries (the C2 matrix without dictionary forms / with permuted rows)")
    ap.add_argument("--print-source-hash", action="store_true", help="print the kernel-source hash PMC summaries are stamped with")
    a = ap.parse_args()
    if a.print_source_hash:
        print(kernel_sources_hash())
        return

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a.gpus, sys.argv[1:])               # never returns

    # multi-process GPU work on this host driver needs dmabuf IPC (RCCL fails with `hipIpcGetMemHandle: invalid
    # argument` otherwise); set before the HIP runtime is loaded, whatever the launcher exported
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # NOTE: this process never imports torch — PyTorch's wheel bundles a second ROCm runtime
    # (7.0) that must not share an address space with the /opt/rocm runtime libits_hip.so uses
    # (iterative_solvers_amd/rendezvous.py explains).  The launcher may still be
    # `python -m torch.distributed.run`: only its RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* are read.
    from iterative_solvers_amd import capi, matgen, solver as S
    from iterative_solvers_amd.rendezvous import Rendezvous

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and not (world == 1 and a.gpus <= 1):
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE={world}")
    rdzv = Rendezvous(rank, world)
    dist = rdzv if world > 1 else None
    fmt = {"exact": capi.FMT_CSR_EXACT, "vector": capi.FMT_CSR_VECTOR, "sell": capi.FMT_SELL}[a.format]

    # ---- which grid.  NX = plane edge, PL = planes this rank owns, NZ = planes of the whole grid
    # ITS_BENCH_FORCE_DIST=1: take the multi-GPU code path (RCCL communicators, split SpMV, all-reduced
    # scalars) even at world = 1, to see what that path costs per trip without a second GPU
    # ITS_BENCH_FORCE_DIST=2: the same, and the rank is its own z-neighbour on both sides (the middle slab of three
    # with the outer two folded onto it), so the halo planes really travel through pack kernel -> RCCL send/recv ->
    # boundary SpMV: the whole per-trip cost of the N > 1 path except the xGMI wire (never a reported number).
    # With --global-grid G the rehearsed slab is one of ITS_BENCH_REHEARSE_RANKS (default 8) slabs of the G^3 grid.
    force_dist = os.environ.get("ITS_BENCH_FORCE_DIST", "0") in ("1", "2")
    loopback = os.environ.get("ITS_BENCH_FORCE_DIST", "0") == "2"
    gg = a.global_grid if a.global_grid is not None else (512 if a.gpus == 8 else 0)
    if gg > 0:
        parts = world if world > 1 else (int(os.environ.get("ITS_BENCH_REHEARSE_RANKS", "8")) if force_dist else 1)
        if gg % parts:
            raise SystemExit(f"--global-grid {gg} is not divisible into {parts} z-slabs")
        NX, PL, NZ = gg, gg // parts, gg
        grid_name = f"{gg}^3 grid shared by {parts} z-slabs of {PL} planes ({NX}x{NX}x{PL} rows per GPU)" if parts > 1 else f"{gg}^3 grid on one GPU"
        slab_name = f"{NX}x{NX}x{PL}-slab" if parts > 1 else f"{gg}^3"
        mode = "global-grid"
    else:
        NX, PL, NZ = a.grid, a.grid, a.grid * world
        grid_name = f"{a.grid}^3 rows per GPU (grid {a.grid}x{a.grid}x{a.grid * world})"
        slab_name = f"{a.grid}^3-slab"
        mode = "weak-series"
    P = NX * NX

    t_gen = time.time()
    t_matgen = None
    # rehearsal of the N > 1 path on a one-GPU box: ITS_BENCH_TRANSPORT=host puts every rank on GPU 0 and
    # moves halo entries / scalars through the TCP rendezvous instead of RCCL (never a reported number)
    host_transport = os.environ.get("ITS_BENCH_TRANSPORT", "rccl") == "host"
    # rank -> GPU: the local rank, or (a launcher that masks the devices per rank) whatever single device is visible
    n_dev = capi.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no HIP device visible to rank %d" % rank)
    if world > 1 and not host_transport and 1 < n_dev < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        raise SystemExit("bench.py: %d ranks on this node but only %d GPUs visible: RCCL needs one GPU per rank" %
                         (int(os.environ.get("LOCAL_WORLD_SIZE", world)), n_dev))
    sol = S.Solver("conjugate-gradient", device=0 if host_transport else local_rank % n_dev, spmv_format=fmt)
    h = sol.handle
    rp = col = val = None
    if world == 1 and not force_dist:
        rp, col, val = matgen.laplace3d_7pt(NX, PL) if PL != NX else matgen.laplace3d_7pt(NX)
        n = len(rp) - 1
        nnz = len(col)
        t_matgen = time.time() - t_gen
        args = S.Args("conjugate-gradient", S.LoopParams(max_iters=10**9, tol=0.0))
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
    else:
        from iterative_solvers_amd import distributed as D
        if loopback and world == 1:
            rp, colg, val = matgen.laplace3d_7pt_slab(NX, PL, 2 * PL, 3 * PL)
        elif world == 1:
            rp, colg, val = matgen.laplace3d_7pt_slab(NX, 0, PL, PL)
        else:
            rp, colg, val = matgen.laplace3d_7pt_slab(NX, rank * PL, (rank + 1) * PL, NZ)
        n = len(rp) - 1
        nnz = len(colg)
        if host_transport:
            D.init_comm_host(h, rank, world, rdzv)
        else:
            D.init_comm(h, rank, world, rdzv)
        if loopback and world == 1:
            import ctypes as C
            colg = np.asarray(colg, dtype=np.int64)
            coll = np.where(colg < n, n + (colg - (n - P)),                    # plane below -> halo [n, n+P)
                            np.where(colg >= 2 * n, n + P + (colg - 2 * n),        # plane above -> halo [n+P, n+2P)
                                     colg - n)).astype(np.int32)
            h.upload_crs(rp, coll, val, capi.MAT_A, fmt, n_cols=n + 2 * P)
            send_idx = np.concatenate([np.arange(n - P, n), np.arange(0, P)]).astype(np.int32)   # own top, own bottom
            nbr = np.zeros(2, np.int32); off = np.array([0, P, 2 * P], np.int32)   # two "neighbours", one plane each, as at N > 1
            h.check(h.L.its_dist_setup(h.h, 2, nbr.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                       send_idx.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p)), "its_dist_setup")
        else:
            D.upload_slab(h, rdzv, world * n, rank * n, rp, colg, val, fmt)
        diag = np.full(n, 6.0)
        h.upload_vectors(diag, np.ones(n), np.ones(n))
        h.init_residual()
        h.solver_setup(capi.CG)
    t_gen = time.time() - t_gen

    def barrier():
        # device-wide sync (hipStreamSynchronize on the only stream that has work) on both
        # sides of the cross-rank barrier
        h.synchronize()
        if dist is not None:
            dist.barrier()
        h.synchronize()

    it = 0

    def cg_step():
        nonlocal it
        h.iterate(it); it += 1
    for _ in range(a.warmup):
        cg_step()
    dts = timed_regions(cg_step, a.steps, REGIONS, barrier, dist)
    st = region_stats(dts, a.steps)
    ms_per_step = st["median"]
    its_per_s = 1e3 / ms_per_step

    # same loop without the per-trip read-back (one sync at the end) — reported beside `value`
    res = h.iterate_n(it, a.steps); it += a.steps
    async_ms = h.last_region_ms() / a.steps
    inloop_spmv_ms = h.last_spmv_ms()          # HIP event pairs around the SpMV launches of that loop

    c2_single = world == 1 and not force_dist and mode == "weak-series" and a.grid == 256
    out = None
    if rank == 0:
        # dominant kernel: the in-loop SpMV (+ fused p.Ap partials), HIP events on the handle's stream
        xd = h.to_dev(np.ones(n + 2 * P)); yd = h.dev_alloc(8 * n)
        roof = None
        spmv_all = {}
        if inloop_spmv_ms > 0:
            # the kernel as it runs INSIDE the CG loop (caches in the loop's state): its average
            # launch duration is what rocprofv3 --stats reports for the same command
            k_ms = inloop_spmv_ms
            ach = spmv_bytes(n, nnz) / (k_ms * 1e-3) / 1e9
            kname, streamed = h.spmv_kernel_info(capi.MAT_A, True) if fmt == capi.FMT_CSR_EXACT else (a.format, spmv_bytes(n, nnz))
            traffic, tnote = pmc_traffic(kname + "<") if (fmt == capi.FMT_CSR_EXACT and c2_single) else (None, "not the C2 single-GPU exact run")
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": tnote,
                    "kernel": f"its::{kname} (CSR SpMV fused with the p.Ap partial sums)",
                    "kernel_ms": k_ms, "algorithmic_bytes": spmv_bytes(n, nnz),
                    # what the kernel is asked to move in its own (lossless) storage form of the same matrix
                    "streamed_bytes": streamed}
            for nm, f in ((("exact", capi.FMT_CSR_EXACT), ("vector", capi.FMT_CSR_VECTOR)) if (world == 1 and not force_dist) else ()):
                h.spmv_timed(xd, yd, f, reps=3)
                ms = h.spmv_timed(xd, yd, f, reps=a.spmv_reps)
                spmv_all[nm] = {"ms": ms, "GB/s": spmv_bytes(n, nnz) / (ms * 1e-3) / 1e9}
        opt_in = None
        if world == 1 and not force_dist and fmt == capi.FMT_CSR_EXACT:
            # opt-in (never `value`): the 7-pt matrix also has only 2 distinct values, so val can stream as a
            # 1-byte dictionary index too (lossless, bit-identical results; its_set_option "spmv_valdict")
            h.set_option("spmv_valdict", 1)
            h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=3)
            ms_vd = h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=a.spmv_reps)
            h.iterate_n(it, a.steps); it += a.steps
            opt_in = {"spmv_valdict": {"spmv_ms": ms_vd, "cg_ms_per_step": h.last_region_ms() / a.steps,
                                       "streamed_bytes": spmv_bytes(n, nnz) - 10 * nnz}}
            h.set_option("spmv_valdict", 0)
            if c2_single and not a.no_generic:
                try:
                    opt_in["mixed_precision"] = mixed_precision_entries(capi, S, local_rank % n_dev, rp, col, val, n, nnz, a.steps, a.spmv_reps)
                except Exception as e:                             # an extra: never fails the bench line
                    opt_in["mixed_precision"] = {"error": repr(e)[:200]}
        cpu = None
        if not a.no_cpu and world == 1 and not force_dist:
            cpu = cpu_baseline(rp, col, val, n)
        out = {
            "metric": "cg_iters_per_sec", "value": its_per_s * world, "unit": f"it/s ({slab_name} iterations)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "ms_per_step_min": st["min"], "ms_per_step_max": st["max"], "regions": REGIONS,
            "higher_is_better": True, "scaling": "weak" if (mode == "weak-series" or world == 8) else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"CG fp64, 3-D 7-pt Laplacian, {grid_name}, b=1, x0=1, -scale none",
                       "grid_mode": mode + (" (BASELINE configs[4])" if (mode == "global-grid" and gg == 512 and world == 8) else ""),
                       "rows_per_gpu": n, "nnz_per_gpu": nnz, "spmv_format": a.format,
                       "parallelism": (f"row-block z-slabs x{world}" + (" (HOST rehearsal transport, one GPU)" if host_transport else " over RCCL")) if world > 1
                                      else ("single GPU, RCCL loopback rehearsal of the N > 1 path" if loopback else "single GPU")},
            "cg_GBps_algorithmic": cg_iter_bytes(n, nnz) * its_per_s / 1e9,
            "cg_frac_of_hbm_peak": cg_iter_bytes(n, nnz) * its_per_s / 1e9 / HBM_PEAK_GBS,
            # bytes the trip actually asks for: the SpMV in its own storage form + 8 vector passes (x += alpha p runs
            # inside the p pass, so a trip is 64 n, not the 72 n of SURVEY.md §8d's accounting above)
            "cg_GBps_streamed": ((roof["streamed_bytes"] + 64 * n) * its_per_s / 1e9) if roof else None,
            "async_loop_ms_per_step": async_ms,
            "last_residual": float(res[-1]),
            "spmv": spmv_all,
            # matrix generation (numpy) apart from what the library needs: upload, plans, ELL image, first residual
            "generate_seconds": t_matgen, "setup_seconds": (t_gen - t_matgen) if t_matgen is not None else t_gen,
            "opt_in": opt_in, "roofline": roof, "cpu_baseline": cpu,
        }
    # second solver of BASELINE.json configs[4] ("CG + Jacobi"): the reference's correct Jacobi sweep
    # (jacobi.hpp:65-93) on the same resident system, same loop body and timing protocol; reported beside `value`
    h.upload_vectors(np.full(n, 6.0), np.ones(n), np.ones(n))
    h.init_residual()
    h.solver_setup(capi.JACOBI, jacobi_mode=capi.JACOBI_REF)
    jt = 0

    def jacobi_step():
        nonlocal jt
        h.iterate(jt); jt += 1
    for _ in range(a.warmup):
        jacobi_step()
    jst = region_stats(timed_regions(jacobi_step, a.steps, 3, barrier, dist), a.steps)
    if rank == 0:
        # one pass over A per trip (sweep fused with the recorded residual): SpMV bytes + b
        jb = spmv_bytes(n, nnz) + 8 * n
        out["jacobi"] = {"value": 1e3 / jst["median"] * world, "unit": out["unit"], "ms_per_step": jst["median"],
                         "ms_per_step_min": jst["min"], "ms_per_step_max": jst["max"],
                         "GBps_algorithmic_per_gpu": jb / jst["median"] / 1e6}
    # row-partitioned runs only, opt-in and never `value`: pipelined CG (its_set_option "cg_variant" 1) — one
    # reduction point per trip instead of two — on the same resident system with the same timing protocol, so that the
    # scaling record shows what the second all-reduce of a trip costs on this fabric.  ITS_BENCH_NO_PIPELINED=1 skips it.
    if (world > 1 or force_dist) and os.environ.get("ITS_BENCH_NO_PIPELINED", "0") != "1":
        h.set_option("cg_variant", 1)
        h.upload_vectors(np.full(n, 6.0), np.ones(n), np.ones(n))
        h.init_residual()
        h.solver_setup(capi.CG)
        pt = 0

        def pcg_step():
            nonlocal pt
            h.iterate(pt); pt += 1
        for _ in range(a.warmup):
            pcg_step()
        pst = region_stats(timed_regions(pcg_step, a.steps, 3, barrier, dist), a.steps)
        if rank == 0:
            out["opt_in"] = dict(out.get("opt_in") or {}, pipelined_cg={
                "value": 1e3 / pst["median"] * world, "unit": out["unit"], "ms_per_step": pst["median"],
                "note": "Ghysels-Vanroose recurrence, outside the parity contract: one reduction point per trip"})
    if dist is not None:
        dist.barrier()
    if rank == 0 and c2_single and fmt == capi.FMT_CSR_EXACT and not a.no_generic:
        # the exact SpMV of the same matrix without any dictionary form, and with its rows randomly permuted
        out["spmv"].update(generic_spmv_entries(h, capi, rp, col, val, n, nnz, xd, yd, a.spmv_reps))
    h.close()
    if rank == 0:
        if c2_single and not a.no_extra:
            # BASELINE.json configs[2] and configs[3] on the same GPU, as sub-objects (never `value`)
            out["gauss_seidel"] = gauss_seidel_entries(capi, S, rp, col, val, n, nnz)
            del rp, col, val
            out["configs"] = other_configs(capi, matgen, S)
        print(json.dumps(out), flush=True)
    rdzv.close()


if __name__ == "__main__":
    main()
