#!/usr/bin/env python3
"""Exact-SpMV time per epilogue (its_spmv_timed modes 0..7) on the power-law matrix (MAT=pl, N rows) or the
7-pt Laplacian (MAT=lap7, N^3 rows); PERM=1 applies the multicolour ordering first."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen
MAT = os.environ.get("MAT", "pl")
N = int(os.environ.get("N", "10000000" if MAT == "pl" else "256"))
rp, col, val = matgen.powerlaw(N) if MAT == "pl" else matgen.laplace3d_7pt(N)
if os.environ.get("PERM", "0") == "1":
    perm, nc = capi.multicolor_order(rp, col)
    rp, col, val = capi.permute_crs(rp, col, val, perm)
n = len(rp) - 1
h = capi.Handle(0)
h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
xd = h.to_dev(np.ones(n)); yd = h.dev_alloc(8 * n)
names = ["store", "store+dot", "residual stored", "residual norm", "jacobi ref", "gs rhs (all of A)", "jacobi fused", "gs fused"]
for mode, nm in enumerate(names):
    h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=3, with_dot=mode)
    print("%-20s %.4f ms" % (nm, h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=int(os.environ.get("REPS", "20")), with_dot=mode)), flush=True)
h.close()
