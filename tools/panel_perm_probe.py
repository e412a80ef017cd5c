#!/usr/bin/env python3
"""Why is the panel-blocked residual pass slower on the multicolour-permuted power-law matrix?  Times the plain product on
the natural and the permuted matrix and prints per-tile statistics of both images (computed here with numpy)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterative_solvers_amd import capi, matgen

N = int(os.environ.get("N", "10000000"))
rp, col, val = matgen.powerlaw(N)
n = len(rp) - 1
perm, nc = capi.multicolor_order(rp, col)
rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
h = capi.Handle(0)
for name, (a, b, c) in (("natural", (rp, col, val)), ("multicolour", (rpp, cp, vp))):
    ln = np.diff(a)
    short = ln <= 128
    rows = np.repeat(np.arange(n), ln)
    keep = short[rows]
    tile = rows[keep] >> 12
    panel = b[keep] >> 16
    P = (n + 65535) >> 16
    piece = tile.astype(np.int64) * P + panel
    cnt = np.bincount(piece, minlength=((n + 4095) >> 12) * P)
    chunks = (cnt + 255) // 256
    per_tile = np.bincount(tile, minlength=(n + 4095) >> 12)
    print(name, "entries in image", int(keep.sum()), "chunks", int(chunks.sum()), "fill %.3f" % (keep.sum() / (chunks.sum() * 256.0)),
          "tile entries min/mean/max", per_tile.min(), int(per_tile.mean()), per_tile.max(),
          "largest piece", cnt.max(), "pieces > 256:", int((cnt > 256).sum()), flush=True)
    h.upload_crs(a, b, c, capi.MAT_A, capi.FMT_CSR_EXACT)
    xd = h.to_dev(np.random.default_rng(0).uniform(-1, 1, n)); yd = h.dev_alloc(8 * n)
    for pan in (1, 0):
        h.set_option("spmv_panel", pan)
        h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=3)
        print("   panel", pan, h.spmv_kernel_info(capi.MAT_A)[0], "plain %.4f ms" % h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=20),
              "residual norm-only %.4f ms" % h.spmv_timed(xd, yd, capi.FMT_CSR_EXACT, reps=20, with_dot=3), flush=True)
    h.set_option("spmv_panel", 1)
    h.dev_free(xd); h.dev_free(yd)
