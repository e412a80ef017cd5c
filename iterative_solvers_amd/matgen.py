"""Deterministic synthetic matrices for the solver hot path.

Every generator returns a CRS triple ``(row_ptr int32[n+1], col int32[nnz], val float64[nnz])``
whose within-row entry order is exactly the order the reference's ``read_mtx``
(/root/reference/io_funcs.hpp:249-270: stable sort by row of the file order) followed by
``convert_to_crs`` (/root/reference/utility_funcs.hpp:326-377) would produce for the ``.mtx``
file written by :func:`write_mtx` — i.e. ascending column inside each row, rows ascending.

The reference ships no matrices; the Laplacians are the standard Dirichlet stencils, the
convection-diffusion and power-law matrices are build-defined.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "laplace2d_5pt", "laplace3d_7pt", "convdiff3d_27pt", "powerlaw", "write_mtx",
    "crs_from_coo", "tridiag",
]


def _stencil_crs(dims, offsets, weights):
    """Assemble a stencil matrix on a regular grid with Dirichlet truncation.

    dims    : (nz, ny, nx) (use 1 for unused leading dims)
    offsets : list of (dz, dy, dx), must be listed in ascending linear-offset order
    weights : list of float, same length
    """
    nz, ny, nx = dims
    n = nz * ny * nx
    z, y, x = np.meshgrid(np.arange(nz, dtype=np.int32), np.arange(ny, dtype=np.int32),
                          np.arange(nx, dtype=np.int32), indexing="ij")
    z = z.ravel(); y = y.ravel(); x = x.ravel()
    k = len(offsets)
    valid = np.empty((n, k), dtype=bool)
    for s, (dz, dy, dx) in enumerate(offsets):
        v = np.ones(n, dtype=bool)
        if dz: v &= (z + dz >= 0) & (z + dz < nz)
        if dy: v &= (y + dy >= 0) & (y + dy < ny)
        if dx: v &= (x + dx >= 0) & (x + dx < nx)
        valid[:, s] = v
    del z, y, x
    counts = valid.sum(axis=1, dtype=np.int64)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    nnz = int(row_ptr[-1])
    assert nnz < 2**31, "CRS uses int32 offsets like the reference (sparse_matrix.hpp:20-27)"
    lin = np.array([(dz * ny + dy) * nx + dx for dz, dy, dx in offsets], dtype=np.int64)
    assert np.all(np.diff(lin) > 0), "offsets must ascend so columns ascend inside a row"
    col = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float64)
    # position of stencil leg s inside its row = number of valid legs before it
    pos = np.zeros(n, dtype=np.int64)
    rows = np.arange(n, dtype=np.int64)
    w = np.asarray(weights, dtype=np.float64)
    for s in range(k):
        m = valid[:, s]
        dst = row_ptr[:-1][m] + pos[m]
        col[dst] = (rows[m] + lin[s]).astype(np.int32)
        val[dst] = w[s]
        pos += m
    return row_ptr.astype(np.int32), col, val


def laplace2d_5pt(N: int):
    """2-D 5-point Laplacian on an N×N grid: diagonal 4, face neighbours −1 (C1)."""
    offs = [(0, -1, 0), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 1, 0)]
    return _stencil_crs((1, N, N), offs, [-1.0, -1.0, 4.0, -1.0, -1.0])


def laplace3d_7pt(N: int, nz: int | None = None):
    """3-D 7-point Laplacian, grid index r = (z·N + y)·N + x: diagonal 6, faces −1 (C2/C5).

    ``nz`` lets a caller build a z-slab of ``nz`` planes of N×N (used only by tests; the
    multi-GPU path partitions the assembled global matrix).
    """
    nz = N if nz is None else nz
    offs = [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0)]
    return _stencil_crs((nz, N, N), offs, [-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0])


def convdiff3d_27pt(N: int):
    """Nonsymmetric 27-point convection-diffusion on N³ (C3, build-defined).

    All 26 neighbours −1, centre 26, plus an upwind skew on the x and y face neighbours:
    +x −1+0.5, −x −1−0.5, +y −1+0.25, −y −1−0.25 (row sums unchanged, matrix nonsymmetric).
    """
    offs, w = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                offs.append((dz, dy, dx))
                if (dz, dy, dx) == (0, 0, 0):
                    w.append(26.0)
                elif (dz, dy, dx) == (0, 0, 1):
                    w.append(-1.0 + 0.5)
                elif (dz, dy, dx) == (0, 0, -1):
                    w.append(-1.0 - 0.5)
                elif (dz, dy, dx) == (0, 1, 0):
                    w.append(-1.0 + 0.25)
                elif (dz, dy, dx) == (0, -1, 0):
                    w.append(-1.0 - 0.25)
                else:
                    w.append(-1.0)
    return _stencil_crs((N, N, N), offs, w)


def tridiag(n: int, lo=-1.0, di=2.0, up=-1.0):
    offs = [(0, 0, -1), (0, 0, 0), (0, 0, 1)]
    return _stencil_crs((1, 1, n), offs, [lo, di, up])


# --------------------------------------------------------------------------- power law
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on uint64 arrays (own integer RNG so CPU/GPU boxes agree)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _u01(h: np.ndarray) -> np.ndarray:
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def powerlaw(n: int, seed: int = 12345, max_len: int = 8192):
    """Irregular-row matrix (C4, build-defined).

    Row length ℓ_i = clamp(⌊3·u^(−2/3)⌋, 1, max_len) (Pareto α = 1.5, mean ≈ 9); the diagonal
    is always present; the other columns are counter-hashed uniform in [0, n), de-duplicated
    and sorted; off-diagonals uniform in (−1, 0); diagonal = 1 + Σ|off-diag| (strictly
    diagonally dominant, so Jacobi and Gauss-Seidel converge).
    """
    seed64 = np.uint64(seed)
    rows = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        u = _u01(_splitmix64(rows * np.uint64(0xD1B54A32D192ED03) + seed64))
    ell = np.floor(3.0 * u ** (-2.0 / 3.0))
    ell = np.clip(ell, 1, min(max_len, n)).astype(np.int64)
    # candidate off-diagonal columns: (ell-1) per row, hashed from (row, k)
    noff = ell - 1
    tot = int(noff.sum())
    r_idx = np.repeat(np.arange(n, dtype=np.int64), noff)
    start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(noff, out=start[1:])
    k_idx = np.arange(tot, dtype=np.int64) - start[r_idx]
    with np.errstate(over="ignore"):
        key = (r_idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
               + k_idx.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F) + seed64)
    h = _splitmix64(key)
    c = (h % np.uint64(n)).astype(np.int64)
    v = -_u01(_splitmix64(h))  # (−1, 0)
    # append diagonals, then sort by (row, col) and drop duplicates / diagonal collisions
    keep = c != r_idx
    r_idx, c, v = r_idx[keep], c[keep], v[keep]
    lin = r_idx * n + c
    order = np.argsort(lin, kind="stable")
    lin, v = lin[order], v[order]
    first = np.ones(lin.shape[0], dtype=bool)
    first[1:] = lin[1:] != lin[:-1]
    lin, v = lin[first], v[first]
    r_idx = lin // n
    c = lin - r_idx * n
    offsum = np.bincount(r_idx, weights=np.abs(v), minlength=n)
    diag = 1.0 + offsum
    # merge the diagonal into sorted position
    rr = np.concatenate([r_idx, np.arange(n, dtype=np.int64)])
    cc = np.concatenate([c, np.arange(n, dtype=np.int64)])
    vv = np.concatenate([v, diag])
    order = np.argsort(rr * n + cc, kind="stable")
    rr, cc, vv = rr[order], cc[order], vv[order]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rr, minlength=n), out=row_ptr[1:])
    assert row_ptr[-1] < 2**31
    return row_ptr.astype(np.int32), cc.astype(np.int32), vv.astype(np.float64)


# --------------------------------------------------------------------------- COO / .mtx
def crs_from_coo(n: int, I, J, V):
    """Stable row sort + CRS, the order read_mtx/convert_to_crs give (io_funcs.hpp:249-270)."""
    I = np.asarray(I, dtype=np.int64); J = np.asarray(J, dtype=np.int32)
    V = np.asarray(V, dtype=np.float64)
    order = np.argsort(I, kind="stable")
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(I, minlength=n), out=row_ptr[1:])
    return row_ptr.astype(np.int32), J[order].copy(), V[order].copy()


def write_mtx(path: str, row_ptr, col, val, n_cols: int | None = None) -> None:
    """Write ``%%MatrixMarket matrix coordinate real general`` with 17 significant digits.

    The banner must start with exactly ``%%MatrixMarket`` (/root/reference/mmio.h:20).
    """
    n = len(row_ptr) - 1
    n_cols = n if n_cols is None else n_cols
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(row_ptr))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n_cols} {len(col)}\n")
        for r, c, v in zip(rows, col, val):
            f.write(f"{r + 1} {int(c) + 1} {float(v):.17g}\n")


def laplace3d_7pt_slab(N: int, z0: int, z1: int, NZ: int):
    """Rows of planes [z0, z1) of the N×N×NZ 7-point Laplacian with GLOBAL int64 column indices.

    Used by the row-block (z-slab) partition of the multi-GPU path: rank r of G owns planes
    [r·NZ/G, (r+1)·NZ/G).  Built from a slab with one ghost plane on each interior side; the
    ghost planes' own rows are dropped, so every kept row equals the global matrix's row.
    """
    g0 = max(z0 - 1, 0); g1 = min(z1 + 1, NZ)
    rp, col, val = laplace3d_7pt(N, nz=g1 - g0)
    plane = N * N
    r0 = (z0 - g0) * plane; r1 = r0 + (z1 - z0) * plane
    a, b = int(rp[r0]), int(rp[r1])
    row_ptr = (rp[r0:r1 + 1].astype(np.int64) - a).astype(np.int32)
    col_g = col[a:b].astype(np.int64) + g0 * plane
    return row_ptr, col_g, val[a:b].copy()
