#!/usr/bin/env python3
"""Per colour of the multicolour ordering of the power-law matrix (host only): rows, entries of L, longest row of L,
and for that row how its entries spread over the colours it reads (what a pipelined walk could do early).
usage: tools/c4_colour_stats.py [n_rows]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rp, col, val = matgen.powerlaw(n)
perm, nc = capi.multicolor_order(rp, col)
rp, col, val = capi.permute_crs(rp, col, val, perm)
(Lp, Lc, Lv), (Up, Uc, Uv) = capi.split_lu(rp, col, val)
Llen = np.diff(Lp)
# colour boundaries: a row starts a new colour when it is the first whose L-row reaches back no further than ... use ranges as the library does
maxcol = np.full(n, -1, dtype=np.int64)
nz = Llen > 0
last = Lp[1:][nz] - 1
maxcol[nz] = Lc[last]
starts = [0]; start = 0
for i in range(n):
    if maxcol[i] >= start:
        start = i; starts.append(i)
starts.append(n)
starts = sorted(set(starts))
print("rows %d entries %d L entries %d colours(ranges) %d (multicolor_order says %d)" % (n, rp[-1], Lp[-1], len(starts) - 1, nc))
bounds = np.array(starts)
for c in range(len(starts) - 1):
    lo, hi = starts[c], starts[c + 1]
    ll = Llen[lo:hi]
    ent = int(Lp[hi] - Lp[lo])
    if ent == 0:
        print("colour %2d rows %9d L entries %9d" % (c, hi - lo, 0)); continue
    k = lo + int(np.argmax(ll))
    cols = Lc[Lp[k]:Lp[k + 1]]
    seg = np.searchsorted(bounds, cols, side="right") - 1
    hist = np.bincount(seg, minlength=c)
    nlong = int((ll > 128).sum()); elong = int(ll[ll > 128].sum())
    print("colour %2d rows %9d L entries %9d longest %5d rows>128 %6d (entries %9d) >2048 %4d  longest row's entries per colour read: %s" %
          (c, hi - lo, ent, ll.max(), nlong, elong, int((ll > 2048).sum()), hist.tolist()))
