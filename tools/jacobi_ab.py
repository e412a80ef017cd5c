#!/usr/bin/env python3
"""Jacobi / Gauss-Seidel trip time on the N^3 7-pt Laplacian: one matrix pass per trip (fused_residual = 1,
default) against the separate sweep + residual launches (0).  Usage: jacobi_ab.py [jacobi] [gauss-seidel]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
N = int(os.environ.get("N", "256"))
TRIPS = int(os.environ.get("TRIPS", "50"))
rp, col, val = matgen.laplace3d_7pt(N)
n = len(rp) - 1
for name in sys.argv[1:] or ["jacobi"]:
    orders = (capi.GS_LEVEL, capi.GS_MULTICOLOR) if name == "gauss-seidel" else (0,)
    if os.environ.get("ORDER"): orders = (int(os.environ["ORDER"]),)
    for order in orders:
        for fused in ((1, 0) if not os.environ.get("FUSED") else (int(os.environ["FUSED"]),)):
            args = S.Args(name, S.LoopParams(max_iters=10**9, tol=0.0))
            sol = S.Solver(name, device=0, jacobi_mode=capi.JACOBI_REF, gs_order=order)
            sol.handle.set_option("fused_residual", fused)
            if os.environ.get("SWEEP_ELL"): sol.handle.set_option("sweep_ell", int(os.environ["SWEEP_ELL"]))
            S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
            h = sol.handle
            for it in range(5): h.iterate(it)
            h.synchronize(); t0 = time.perf_counter()
            for it in range(5, 5 + TRIPS): h.iterate(it)
            h.synchronize()
            print("%s order %d fused %d: %.4f ms/trip" % (name, order, fused, (time.perf_counter() - t0) / TRIPS * 1e3), flush=True)
            h.close()
