#!/usr/bin/env python3
"""A/B timing of the exact SpMV launch tunables on the 256^3 7-pt Laplacian (one GPU).
usage: python tools/spmv_ab.py "grid=1024,map=1,nt=0" "grid=1024,map=1,nt=1" ...   [N=256]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen

N = int(os.environ.get("N", "256"))
REPS = int(os.environ.get("REPS", "40"))
ROUNDS = int(os.environ.get("ROUNDS", "2"))
NZ = int(os.environ.get("NZ", "0")) or None
MAT = os.environ.get("MAT", "lap7")
if MAT == "pl":
    rp, col, val = matgen.powerlaw(N)
    print("power-law %d rows" % N, flush=True)
elif MAT == "cd27":
    rp, col, val = matgen.convdiff3d_27pt(N)
    print("27-pt convection-diffusion %d^3" % N, flush=True)
else:
    rp, col, val = matgen.laplace3d_7pt(N, NZ)
    print("grid %d x %d x %d" % (N, N, NZ or N), flush=True)
n, nnz = len(rp) - 1, len(col)
by = 12 * nnz + 4 * (n + 1) + 16 * n
h = capi.Handle(0)
h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
xd = h.to_dev(np.random.default_rng(0).uniform(-1, 1, n)); yd = h.dev_alloc(8 * n)
cfgs = sys.argv[1:] or ["grid=1024,map=1,nt=0"]
for rnd in range(ROUNDS):
    for cfg in cfgs:
        reup = False
        for kv in cfg.split(","):
            k, v = kv.split("=")
            h.set_option("spmv_" + k, int(v))
            reup |= k in ("panel_log2pw", "panel_min_cols", "panel_lcut")        # read when the image is built
        if reup:
            h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
        h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=5)
        ms = h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=REPS)
        msd = h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=REPS, with_dot=True)
        line = "%-32s plain %.4f ms (%.0f GB/s)   with-dot %.4f ms (%.0f GB/s)" % (cfg, ms, by / ms / 1e6, msd, by / msd / 1e6)
        if os.environ.get("RES"):
            # residual epilogue with and without storing r: what the y store costs inside the library kernel
            L = h.L; import ctypes as C
            out = C.c_float()
            def timed(mode):
                h.check(L.its_spmv_timed(h.h, capi.MAT_A, C.c_void_p(xd), C.c_void_p(yd), capi.FMT_CSR_EXACT, mode, 5, C.byref(out)), "t")
                h.check(L.its_spmv_timed(h.h, capi.MAT_A, C.c_void_p(xd), C.c_void_p(yd), capi.FMT_CSR_EXACT, mode, REPS, C.byref(out)), "t")
                return out.value
            line += "   residual: stored %.4f  norm-only %.4f" % (timed(2), timed(3))
        print(line, flush=True)
