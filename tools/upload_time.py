#!/usr/bin/env python3
"""Wall time of its_upload_crs (validation, H2D copies, dictionaries, plans) for the N^3 7-pt Laplacian."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen
N = int(os.environ.get("N", "256"))
t0 = time.time(); rp, col, val = matgen.laplace3d_7pt(N); print("matgen %.2f s" % (time.time() - t0))
h = capi.Handle(0)
for fmt, nm in ((capi.FMT_CSR_EXACT, "exact"), (capi.FMT_SELL, "sell")):
    t0 = time.time(); h.upload_crs(rp, col, val, capi.MAT_A, fmt); print("upload_crs %s %.2f s" % (nm, time.time() - t0), flush=True)
n = len(rp) - 1
t0 = time.time(); h.upload_vectors(np.full(n, 6.0), np.ones(n), np.ones(n)); print("upload_vectors %.2f s" % (time.time() - t0))
h.close()
