"""The OpenMP set-up passes of csrc/its_host.cpp produce exactly what a plain serial restatement produces
("identical plans"): colouring = sequential greedy in the documented vertex order, permutation,
L / U split and diagonal extraction, with 1 thread and with many.  CPU only: these entry points need no GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from iterative_solvers_amd import capi, matgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MATS = {
    "lap2d_24": lambda: matgen.laplace2d_5pt(24),
    "lap3d_9": lambda: matgen.laplace3d_7pt(9),
    "cd27_7": lambda: matgen.convdiff3d_27pt(7),
    "pl_4000": lambda: matgen.powerlaw(4000),            # rows of >= 512 entries: the hubs-first order matters
    "pl_700": lambda: matgen.powerlaw(700, seed=7),
}


def greedy_serial(rp, col, hub_row=512):
    """smallest colour no earlier neighbour (graph of A + A^T) holds; order: rows of >= hub_row entries first, then
    the others, each group ascending; new order = stable sort by colour"""
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    nb = [set() for _ in range(n)]
    for i, j in zip(rows.tolist(), col.tolist()):
        if i != j:
            nb[i].add(j); nb[j].add(i)
    ln = np.diff(rp)
    order = [i for i in range(n) if ln[i] >= hub_row] + [i for i in range(n) if ln[i] < hub_row]
    colour = np.full(n, -1)
    for i in order:
        used = {colour[j] for j in nb[i] if colour[j] >= 0}
        c = 0
        while c in used:
            c += 1
        colour[i] = c
    return np.argsort(colour, kind="stable"), int(colour.max()) + 1


@pytest.mark.parametrize("name", sorted(MATS))
def test_parallel_colouring_equals_the_serial_greedy(name):
    rp, col, val = MATS[name]()
    perm, nc = capi.multicolor_order(rp, col)
    ref_perm, ref_nc = greedy_serial(rp, col)
    assert nc == ref_nc
    assert np.array_equal(perm, ref_perm)
    if name == "pl_4000":
        assert np.diff(rp).max() >= 512


@pytest.mark.parametrize("name", sorted(MATS))
def test_permute_split_and_diagonal_equal_numpy(name):
    rp, col, val = MATS[name]()
    n = len(rp) - 1
    perm, _ = capi.multicolor_order(rp, col)
    rpp, cp, vp = capi.permute_crs(rp, col, val, perm)
    old2new = np.empty(n, np.int64); old2new[perm] = np.arange(n)
    rows = np.repeat(np.arange(n), np.diff(rp))
    I, J = old2new[rows], old2new[col]
    order = np.lexsort((J, I))                                  # stable: row, then column
    assert np.array_equal(cp, J[order]) and np.array_equal(vp, val[order])
    assert np.array_equal(rpp, np.concatenate([[0], np.cumsum(np.bincount(I, minlength=n))]))
    # strict triangles keep the CRS order of their entries; D = the diagonal entry (0 where a row has none)
    (lrp, lc, lv), (urp, uc, uv) = capi.split_lu(rpp, cp, vp)
    rws = np.repeat(np.arange(n), np.diff(rpp))
    lo, up, dg = cp < rws, cp > rws, cp == rws
    assert np.array_equal(lc, cp[lo]) and np.array_equal(lv, vp[lo]) and np.array_equal(uc, cp[up]) and np.array_equal(uv, vp[up])
    assert np.array_equal(lrp, np.concatenate([[0], np.cumsum(np.bincount(rws[lo], minlength=n))]))
    assert np.array_equal(urp, np.concatenate([[0], np.cumsum(np.bincount(rws[up], minlength=n))]))
    D = capi.extract_diag(rpp, cp, vp)
    ref = np.zeros(n); ref[rws[dg]] = vp[dg]
    assert np.array_equal(D, ref)


def test_thread_count_does_not_change_the_result():
    """the same passes with OMP_NUM_THREADS = 1 and 8 in fresh processes: identical bytes"""
    code = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "from iterative_solvers_amd import capi, matgen\n"
        "h = hashlib.sha256()\n"
        "for mk in (lambda: matgen.powerlaw(4000), lambda: matgen.convdiff3d_27pt(7)):\n"
        "    rp, col, val = mk()\n"
        "    perm, nc = capi.multicolor_order(rp, col)\n"
        "    rpp, cp, vp = capi.permute_crs(rp, col, val, perm)\n"
        "    (lrp, lc, lv), (urp, uc, uv) = capi.split_lu(rpp, cp, vp)\n"
        "    D = capi.extract_diag(rpp, cp, vp)\n"
        "    for a in (perm, rpp, cp, vp, lrp, lc, lv, urp, uc, uv, D): h.update(np.ascontiguousarray(a).tobytes())\n"
        "    h.update(str(nc).encode())\n"
        "print(h.hexdigest())\n" % ROOT)
    digests = []
    for nt in ("1", "8"):
        env = dict(os.environ, OMP_NUM_THREADS=nt)
        digests.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True, timeout=300).stdout.strip())
    assert digests[0] == digests[1] and len(digests[0]) == 64
