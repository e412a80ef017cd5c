#!/usr/bin/env python3
"""SpMV time of every storage format on one matrix (LAP=N 7-pt, CD27=N 27-pt, PL=N power-law)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen
rp, col, val = (matgen.laplace3d_7pt(int(os.environ["LAP"])) if os.environ.get("LAP") else
                matgen.convdiff3d_27pt(int(os.environ["CD27"])) if os.environ.get("CD27") else matgen.powerlaw(int(os.environ.get("PL", "1000000"))))
n, nnz = len(rp) - 1, len(col)
by = 12 * nnz + 4 * (n + 1) + 16 * n
for name, fmt in (("exact", capi.FMT_CSR_EXACT), ("vector", capi.FMT_CSR_VECTOR), ("sell", capi.FMT_SELL)):
    h = capi.Handle(0)
    h.upload_crs(rp, col, val, capi.MAT_A, fmt)
    xd = h.to_dev(np.random.default_rng(0).uniform(-1, 1, n)); yd = h.dev_alloc(8 * n)
    h.spmv_timed(xd, yd, fmt, reps=5)
    ms = min(h.spmv_timed(xd, yd, fmt, reps=40) for _ in range(2))
    print("%-7s %.4f ms  %.0f GB/s algorithmic  %s" % (name, ms, by / ms / 1e6, h.plan_info(capi.MAT_A, fmt)), flush=True)
    h.close()
