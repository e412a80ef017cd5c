#!/usr/bin/env python3
"""Timing of the explicit-column ELL kernel (csrc/its_ellg.hip) on the 256^3 7-pt Laplacian with the dictionary forms
switched off, next to the dictionary kernel, the CRS stream kernels, and on the same matrix with randomly permuted rows.
usage: python tools/ellg_ab.py [N=256]      env: REPS, PERM=0 to skip the permuted matrix, GRIDS="512,768"
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = int(os.environ.get("REPS", "40"))
rp, col, val = matgen.laplace3d_7pt(N)
n, nnz = len(rp) - 1, len(col)
by = 12 * nnz + 4 * (n + 1) + 16 * n
h = capi.Handle(0)
xd = h.to_dev(np.random.default_rng(0).uniform(-1, 1, n)); yd = h.dev_alloc(8 * n)


def run(tag):
    name, streamed = h.spmv_kernel_info(capi.MAT_A, True)
    h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=5)
    ms = h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=REPS)
    msd = h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=REPS, with_dot=True)
    print("%-44s %-20s plain %.4f ms  CG form %.4f ms = %.0f GB/s algorithmic (%.3f of 8 TB/s), %.0f GB/s streamed; streamed %.3f GB"
          % (tag, name, ms, msd, by / msd / 1e6, by / msd / 1e6 / 8000, streamed / msd / 1e6, streamed / 1e9), flush=True)


h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
run("dictionary ELL (default)")
h.set_option("spmv_ell", 0)
run("CRS stream kernels, offset dictionary")
h.set_option("spmv_ell", 1)
h.set_option("spmv_ell_dict", 0)
h.upload_crs(rp, col, val, capi.MAT_A, capi.FMT_CSR_EXACT)
print("ell_info:", h.ell_info(), flush=True)
for g in [int(v) for v in os.environ.get("GRIDS", "512").split(",")]:
    h.set_option("spmv_ell_grid", g)
    run("explicit-column ELL, grid %d" % g)
h.set_option("spmv_ell_grid", 512)
h.set_option("spmv_ell_map", 0)
run("explicit-column ELL, contiguous eighths")
h.set_option("spmv_ell_map", 1)
if os.environ.get("PERM", "1") != "0":
    perm = np.random.default_rng(7).permutation(n)
    ln = np.diff(rp)[perm]
    rq = np.concatenate([[0], np.cumsum(ln)]).astype(np.int32)
    starts = rp[:-1][perm]
    idx = (np.repeat(starts - rq[:-1], ln) + np.arange(nnz)).astype(np.int64)
    cq, vq = col[idx], val[idx]
    del idx
    h.upload_crs(rq, cq, vq, capi.MAT_A, capi.FMT_CSR_EXACT)
    print("ell_info (rows permuted):", h.ell_info(), flush=True)
    run("rows randomly permuted, explicit-column ELL")
    h.set_option("spmv_ell", 0)
    os.environ["ITS_NO_OFFSET_DICT"] = "1"
    run("rows randomly permuted, CRS stream kernels")
    h.set_option("spmv_ell", 1)
h.close()
