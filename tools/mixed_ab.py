#!/usr/bin/env python3
"""SpMV / CG step on the tiered sliced-ELL image of the 7-pt N^3 Laplacian (ITS_FMT_MIXED, threshold 2) against the fp64
image.  usage: mixed_ab.py [N]   (ITS_SPMV_ELLM_GRID selects the grid)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rp, col, val = matgen.laplace3d_7pt(N)
n = len(rp) - 1
for tag, fmt, mode in (("fp64", capi.FMT_CSR_EXACT, None), ("ap[dp_sp]", capi.FMT_MIXED, capi.AP_DP_SP), ("ap[dp_hp]", capi.FMT_MIXED, capi.AP_DP_HP)):
    sol = S.Solver("conjugate-gradient", device=0, spmv_format=fmt)
    h = sol.handle
    if mode is not None: h.set_mixed_mode(mode, 2.0, 0.0)
    args = S.Args("conjugate-gradient", S.LoopParams(max_iters=10**9, tol=0.0))
    S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
    xd = h.to_dev(np.ones(n)); yd = h.dev_alloc(8 * n)
    h.spmv_timed(xd, yd, fmt, reps=3, with_dot=True)
    ms = h.spmv_timed(xd, yd, fmt, reps=50, with_dot=True)
    h.iterate_n(0, 20); h.iterate_n(20, 200)
    print("%-10s grid %s: spmv %.4f ms, CG %.4f ms/step" % (tag, os.environ.get("ITS_SPMV_ELLM_GRID", "512"), ms, h.last_region_ms() / 200), flush=True)
    h.close()
