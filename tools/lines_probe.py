#!/usr/bin/env python3
"""Line-pipelined sweep: trip time on rectangular 2-D grids (ny lines of nx rows) — separates the effect of the line length
(the stride between the lanes' rows) from that of the number of tiles.  usage: lines_probe.py <ny nx | NZxNYxNX> ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
TRIPS = int(os.environ.get("TRIPS", "10"))
a = sys.argv[1:]
grids = []
while a:
    if "x" in a[0]:                                          # NZxNYxNX: a 3-D grid (7-point)
        grids.append(tuple(int(v) for v in a.pop(0).split("x")))
    else:
        grids.append((1, int(a.pop(0)), int(a.pop(0))))
for nz, ny, nx in grids:
    if nz == 1:
        offs = [(0, -1, 0), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 1, 0)]
        rp, col, val = matgen._stencil_crs((1, ny, nx), offs, [-1.0, -1.0, 4.0, -1.0, -1.0])
    else:
        offs = [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0)]
        rp, col, val = matgen._stencil_crs((nz, ny, nx), offs, [-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0])
    n = len(rp) - 1
    for lines in (1, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_LEVEL)
        sol.handle.set_option("sweep_lines", 2 * lines)
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        h = sol.handle
        for it in range(3): h.iterate(it)
        h.synchronize(); t0 = time.perf_counter()
        for it in range(3, 3 + TRIPS): h.iterate(it)
        h.synchronize()
        ms = (time.perf_counter() - t0) / TRIPS * 1e3
        info = h.sweep_info()
        steps = nx // 4 + ny + nz - 1 if lines else 0
        print("%d lines of %d rows, sweep_lines %d: %.4f ms/trip%s" % (ny * nz, nx, lines, ms, (", %d tiles, %.2f us per critical step" % (info["workgroups"], ms * 1e3 / steps)) if lines else ""), flush=True)
        h.close()
