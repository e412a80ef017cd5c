"""Shared helpers for the parity tests."""
import numpy as np

from iterative_solvers_amd import matgen


def random_crs(n, max_len, seed, n_cols=None, empty_every=0, long_rows=()):
    """Ragged CRS with unsorted, possibly repeated columns (the reference never sorts inside a row)."""
    rng = np.random.default_rng(seed)
    n_cols = n if n_cols is None else n_cols
    lens = rng.integers(0, max_len + 1, size=n)
    if empty_every:
        lens[::empty_every] = 0
    for r, l in long_rows:
        lens[r] = l
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=rp[1:])
    nnz = int(rp[-1])
    col = rng.integers(0, n_cols, size=nnz).astype(np.int32)
    val = rng.uniform(-1, 1, size=nnz)
    return rp.astype(np.int32), col, val


MATRICES = {
    "lap2d_16": lambda: matgen.laplace2d_5pt(16),
    "lap2d_64": lambda: matgen.laplace2d_5pt(64),
    "lap3d_8": lambda: matgen.laplace3d_7pt(8),
    "lap3d_16": lambda: matgen.laplace3d_7pt(16),
    "cd27_6": lambda: matgen.convdiff3d_27pt(6),
    "cd27_12": lambda: matgen.convdiff3d_27pt(12),
    "pl_200": lambda: matgen.powerlaw(200),
    "pl_5000": lambda: matgen.powerlaw(5000),
}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.max(np.abs(a - b)) if a.size else 0.0
    s = np.max(np.abs(b)) if b.size else 1.0
    return d / (s if s > 0 else 1.0)
