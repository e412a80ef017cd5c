"""Row-block partition of the solver hot path over N GPUs (the reference is a
single process — "TODO: MPI preprocessing will go here", main.cpp:94 — so this layer is new).

Host-side planning only (numpy, runs once):

* ``plan_partition``   local column renumbering (owned ``[0, n_loc)`` + halo ``[n_loc, n_loc+h)``),
                       per-neighbour receive ranges of the halo and send index lists;
* ``upload_slab``      hands the local CRS + exchange plan to ``libits_hip.so``;
* ``init_comm``        distributes the RCCL unique ids over the control plane.

All per-iteration traffic (halo x-entries, dot/norm all-reduces) moves over RCCL/xGMI inside the
library.  ``comm`` is any object with ``rank``, ``world`` and ``allgather_bytes(b) -> list[bytes]``
(``Rendezvous`` in production, a gloo adapter in the CPU tests).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi


@dataclass
class PartitionPlan:
    rank: int
    world: int
    n_global: int
    row_begin: int
    n_loc: int
    n_halo: int
    col_local: np.ndarray          # int32 local column of every nonzero
    halo_global: np.ndarray        # int64 global ids of the halo columns, ascending
    nbr: np.ndarray                # int32 neighbour ranks, ascending
    recv_off: np.ndarray           # int32 [n_nbr+1] offsets into the halo region, per neighbour
    send_off: np.ndarray           # int32 [n_nbr+1] offsets into send_idx, per neighbour
    send_idx: np.ndarray           # int32 local row indices to pack, grouped by neighbour


def _allgather_arrays(comm, arr: np.ndarray, dtype) -> list[np.ndarray]:
    blobs = comm.allgather_bytes(np.ascontiguousarray(arr, dtype=dtype).tobytes())
    return [np.frombuffer(b, dtype=dtype) for b in blobs]


def plan_partition(comm, n_global: int, row_begin: int, n_loc: int, col_global: np.ndarray) -> PartitionPlan:
    """Collective: every rank calls it with its own contiguous row block [row_begin, row_begin+n_loc)."""
    rank, world = comm.rank, comm.world
    begins = np.concatenate(_allgather_arrays(comm, np.array([row_begin, n_loc]), np.int64)).reshape(world, 2)
    starts = begins[:, 0]
    ends = begins[:, 0] + begins[:, 1]
    order = np.argsort(starts)
    if np.any(starts[order][1:] != ends[order][:-1]) or starts[order][0] != 0 or ends[order][-1] != n_global:
        raise ValueError("row blocks must tile [0, n_global) without gaps or overlap")
    col_global = np.asarray(col_global, dtype=np.int64)
    if col_global.size and (col_global.min() < 0 or col_global.max() >= n_global):
        raise ValueError("global column index out of range")
    owned = (col_global >= row_begin) & (col_global < row_begin + n_loc)
    halo_global = np.unique(col_global[~owned])                      # ascending => grouped by owner
    n_halo = len(halo_global)
    col_local = np.empty(len(col_global), dtype=np.int32)
    col_local[owned] = (col_global[owned] - row_begin).astype(np.int32)
    col_local[~owned] = (n_loc + np.searchsorted(halo_global, col_global[~owned])).astype(np.int32)
    # owner of every halo column
    owner = np.searchsorted(ends, halo_global, side="right").astype(np.int32) if n_halo else np.zeros(0, np.int32)
    if n_halo and np.any((halo_global < starts[owner]) | (halo_global >= ends[owner])):
        raise ValueError("ranks must own ascending row blocks (rank r before rank r+1)")
    # everybody learns everybody's wish list; each rank keeps the part it owns
    wishes = _allgather_arrays(comm, halo_global, np.int64)
    send_lists = {}
    for r in range(world):
        if r == rank:
            continue
        w = wishes[r]
        mine = w[(w >= row_begin) & (w < row_begin + n_loc)]
        if len(mine):
            send_lists[r] = (mine - row_begin).astype(np.int32)       # ascending = the receiver's halo order
    recv_nbr = np.unique(owner) if n_halo else np.zeros(0, np.int32)
    nbr = np.array(sorted(set(send_lists) | set(int(x) for x in recv_nbr)), dtype=np.int32)
    recv_off = np.zeros(len(nbr) + 1, dtype=np.int32)
    send_off = np.zeros(len(nbr) + 1, dtype=np.int32)
    chunks = []
    for k, r in enumerate(nbr):
        recv_off[k + 1] = recv_off[k] + int(np.count_nonzero(owner == r))
        s = send_lists.get(int(r), np.zeros(0, np.int32))
        chunks.append(s)
        send_off[k + 1] = send_off[k] + len(s)
    send_idx = np.concatenate(chunks) if chunks else np.zeros(0, np.int32)
    assert recv_off[-1] == n_halo
    return PartitionPlan(rank, world, n_global, row_begin, n_loc, n_halo, col_local, halo_global, nbr,
                         recv_off, send_off, send_idx.astype(np.int32))


def exchange_halo_host(comm, plan: PartitionPlan, x_owned: np.ndarray) -> np.ndarray:
    """Reference (host) halo exchange through the control plane: returns x extended by the halo.
    Used by the CPU tests and by the host-staged transport; production uses RCCL in the library."""
    packed = x_owned[plan.send_idx] if len(plan.send_idx) else np.zeros(0)
    meta = np.concatenate([[len(plan.nbr)], plan.nbr, plan.send_off]).astype(np.int64)
    blobs = comm.allgather_bytes(meta.tobytes() + np.ascontiguousarray(packed, dtype=np.float64).tobytes())
    x_ext = np.concatenate([x_owned, np.zeros(plan.n_halo)])
    for k, r in enumerate(plan.nbr):
        b = blobs[int(r)]
        nn = int(np.frombuffer(b[:8], dtype=np.int64)[0])
        hdr = np.frombuffer(b[8:8 * (2 + 2 * nn)], dtype=np.int64)
        their_nbr, their_off = hdr[:nn], hdr[nn:]
        data = np.frombuffer(b[8 * (2 + 2 * nn):], dtype=np.float64)
        j = int(np.nonzero(their_nbr == plan.rank)[0][0])
        seg = data[their_off[j]:their_off[j + 1]]
        x_ext[plan.n_loc + plan.recv_off[k]: plan.n_loc + plan.recv_off[k + 1]] = seg
    return x_ext


# --------------------------------------------------------------------------- library glue
def init_comm(handle: capi.Handle, rank: int, world: int, rdzv) -> None:
    """RCCL communicators (one for halo send/recv, one for all-reduces) from ids made on rank 0."""
    L = handle.L
    ids = C.create_string_buffer(256)
    if rank == 0:
        handle.check(L.its_comm_unique_id(ids), "its_comm_unique_id")
        handle.check(L.its_comm_unique_id(C.byref(ids, 128)), "its_comm_unique_id")
    blob = rdzv.broadcast(ids.raw if rank == 0 else None)
    buf = C.create_string_buffer(blob, 256)
    # RCCL 2.27 prints a version banner on STDOUT at communicator creation; callers such as bench.py own
    # stdout (one JSON line), so file descriptor 1 points at stderr for the duration of the call
    import os
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        rc = L.its_comm_init(handle.h, rank, world, buf)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    handle.check(rc, "its_comm_init")


_XCB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double),
                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int)
_RCB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)


def init_comm_host(handle: capi.Handle, rank: int, world: int, rdzv) -> None:
    """HOST test transport (its_comm_init_host): halo entries and scalars cross ranks through the TCP
    rendezvous, so several ranks can share one GPU.  Everything else (pack kernel, interior/boundary
    SpMV, fused updates) runs exactly as with RCCL.  Used by tests and by `bench.py` rehearsals
    (ITS_BENCH_TRANSPORT=host); never the production data path."""

    def exchange(user, send, send_off, recv, recv_off, nbr, n_nbr):
        try:
            nb = [nbr[k] for k in range(n_nbr)]
            so = [send_off[k] for k in range(n_nbr + 1)]
            data = np.ctypeslib.as_array(send, (max(so[-1], 1),))[:so[-1]] if n_nbr else np.zeros(0)
            meta = np.array([n_nbr] + nb + so, dtype=np.int64)
            blobs = rdzv.allgather_bytes(meta.tobytes() + np.ascontiguousarray(data).tobytes())
            n_recv = recv_off[n_nbr] if n_nbr else 0
            out = np.ctypeslib.as_array(recv, (max(n_recv, 1),))
            for k in range(n_nbr):
                b = blobs[nb[k]]
                nn = int(np.frombuffer(b[:8], dtype=np.int64)[0])
                hdr = np.frombuffer(b[8:8 * (2 + 2 * nn)], dtype=np.int64)
                their_nbr, their_off = hdr[:nn], hdr[nn:]
                payload = np.frombuffer(b[8 * (2 + 2 * nn):], dtype=np.float64)
                j = int(np.nonzero(their_nbr == rank)[0][0])
                seg = payload[their_off[j]:their_off[j + 1]]
                assert len(seg) == recv_off[k + 1] - recv_off[k]
                out[recv_off[k]:recv_off[k + 1]] = seg
            return 0
        except Exception:  # noqa: BLE001
            import traceback; traceback.print_exc()
            return 1

    def allreduce(user, vals, count, op):
        try:
            buf = np.ctypeslib.as_array(vals, (count,))
            parts = [np.frombuffer(b, dtype=np.float64) for b in rdzv.allgather_bytes(buf.tobytes())]
            buf[:] = np.max(parts, axis=0) if op == 1 else np.sum(parts, axis=0)   # rank order: deterministic
            return 0
        except Exception:  # noqa: BLE001
            import traceback; traceback.print_exc()
            return 1

    xcb, rcb = _XCB(exchange), _RCB(allreduce)
    handle._host_cbs = (xcb, rcb)             # keep the trampolines alive as long as the handle
    handle.check(handle.L.its_comm_init_host(handle.h, rank, world, C.cast(xcb, C.c_void_p), C.cast(rcb, C.c_void_p), None),
                 "its_comm_init_host")


def upload_slab(handle: capi.Handle, comm, n_global: int, row_begin: int, row_ptr, col_global, val,
                fmt=capi.FMT_CSR_EXACT) -> PartitionPlan:
    """Plan the partition (collective) and give the local matrix + exchange plan to the library."""
    n_loc = len(row_ptr) - 1
    plan = plan_partition(comm, n_global, row_begin, n_loc, col_global)
    handle.upload_crs(row_ptr, plan.col_local, val, capi.MAT_A, fmt, n_cols=n_loc + plan.n_halo)
    nbr = np.ascontiguousarray(plan.nbr, dtype=np.int32)
    so = np.ascontiguousarray(plan.send_off, dtype=np.int32)
    ro = np.ascontiguousarray(plan.recv_off, dtype=np.int32)
    si = np.ascontiguousarray(plan.send_idx, dtype=np.int32)
    handle.check(handle.L.its_dist_setup(handle.h, len(nbr), nbr.ctypes.data_as(C.c_void_p),
                                         so.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p),
                                         ro.ctypes.data_as(C.c_void_p)), "its_dist_setup")
    return plan


# --------------------------------------------------------------------------- multicolour sweeps across ranks
@dataclass
class ColoredPlan:
    """A row partition in GLOBAL multicolour order ("GS across GPUs: colours are global").

    The global system is B = Q A Q^T with Q the colour-major ordering of one global colouring (rows colour by colour,
    ascending original index inside a colour).  Rank r owns the B-rows whose ORIGINAL row lies in its contiguous
    block — for every colour one contiguous run of B-indices — and numbers them locally in ascending B-index, i.e.
    colour by colour.  The natural-order Gauss-Seidel sweep of the reference on B (what the oracle runs) is then
    "for every colour: all ranks sweep their rows of that colour, then exchange that colour's boundary values"."""
    base: PartitionPlan
    perm_new2old: np.ndarray       # Q: B-index -> original row
    owned_global: np.ndarray       # int64 ascending B-indices of the owned rows
    n_colors: int
    color_ptr: np.ndarray          # int32 [n_colors+1] cuts of the owned rows
    send_color_off: np.ndarray     # int32 [n_nbr, n_colors+1]
    recv_color_off: np.ndarray     # int32 [n_nbr, n_colors+1]
    row_ptr: np.ndarray            # local CRS of the owned B-rows, columns local (owned, then halo)
    col_local: np.ndarray
    val: np.ndarray
    L: tuple                       # (row_ptr, col_local, val): entries whose column precedes the row in B order
    U: tuple                       # entries whose column follows it
    diag: np.ndarray


def plan_owned(comm, n_global: int, owned_global: np.ndarray, col_global: np.ndarray) -> PartitionPlan:
    """plan_partition for an arbitrary ownership: every rank passes the ascending global indices it owns (together
    they must tile [0, n_global)); local numbering = position in that list, halo = the other columns, ascending."""
    rank, world = comm.rank, comm.world
    owned_global = np.ascontiguousarray(owned_global, dtype=np.int64)
    lists = _allgather_arrays(comm, owned_global, np.int64)
    owner_of = np.full(n_global, -1, dtype=np.int32)
    for r, l in enumerate(lists):
        if len(l) and (np.any(np.diff(l) <= 0) or l[0] < 0 or l[-1] >= n_global):
            raise ValueError("owned indices must be ascending and inside [0, n_global)")
        if np.any(owner_of[l] != -1):
            raise ValueError("a row is owned twice")
        owner_of[l] = r
    if np.any(owner_of < 0):
        raise ValueError("owned index lists must tile [0, n_global)")
    n_loc = len(owned_global)
    col_global = np.asarray(col_global, dtype=np.int64)
    owned = owner_of[col_global] == rank
    halo_global = np.unique(col_global[~owned])
    n_halo = len(halo_global)
    col_local = np.empty(len(col_global), dtype=np.int32)
    col_local[owned] = np.searchsorted(owned_global, col_global[owned]).astype(np.int32)
    col_local[~owned] = (n_loc + np.searchsorted(halo_global, col_global[~owned])).astype(np.int32)
    # the halo is laid out neighbour by neighbour (its_dist_setup's recv ranges), ascending global index inside
    owner = owner_of[halo_global] if n_halo else np.zeros(0, np.int32)
    order = np.lexsort((halo_global, owner)) if n_halo else np.zeros(0, np.int64)
    pos = np.empty(n_halo, dtype=np.int64); pos[order] = np.arange(n_halo)
    col_local[~owned] = (n_loc + pos[np.searchsorted(halo_global, col_global[~owned])]).astype(np.int32)
    halo_sorted = halo_global[order]; owner_sorted = owner[order] if n_halo else owner
    wishes = _allgather_arrays(comm, np.concatenate([[n_halo], halo_sorted, owner_sorted]).astype(np.int64), np.int64)
    send_lists = {}
    for r in range(world):
        if r == rank:
            continue
        nh = int(wishes[r][0]); w = wishes[r][1:1 + nh]; wo = wishes[r][1 + nh:1 + 2 * nh]
        mine = w[wo == rank]                                            # in the receiver's halo order
        if len(mine):
            send_lists[r] = np.searchsorted(owned_global, mine).astype(np.int32)
    recv_nbr = np.unique(owner_sorted) if n_halo else np.zeros(0, np.int32)
    nbr = np.array(sorted(set(send_lists) | set(int(x) for x in recv_nbr)), dtype=np.int32)
    recv_off = np.zeros(len(nbr) + 1, dtype=np.int32)
    send_off = np.zeros(len(nbr) + 1, dtype=np.int32)
    chunks = []
    for k, r in enumerate(nbr):
        recv_off[k + 1] = recv_off[k] + int(np.count_nonzero(owner_sorted == r))
        sl = send_lists.get(int(r), np.zeros(0, np.int32))
        chunks.append(sl)
        send_off[k + 1] = send_off[k] + len(sl)
    send_idx = np.concatenate(chunks).astype(np.int32) if chunks else np.zeros(0, np.int32)
    return PartitionPlan(rank, world, n_global, int(owned_global[0]) if n_loc else 0, n_loc, n_halo, col_local, halo_sorted,
                         nbr, recv_off, send_off, send_idx)


def plan_colored(comm, row_ptr_g, col_g, val_g, block_begin: int, block_end: int) -> ColoredPlan:
    """Collective.  Every rank holds the GLOBAL matrix in original order (a set-up convenience of this reference
    implementation: the colouring is global) and its contiguous block [block_begin, block_end) of original rows."""
    n = len(row_ptr_g) - 1
    perm, n_colors = capi.multicolor_order(row_ptr_g, col_g)            # Q, rows colour by colour
    rpB, colB, valB = capi.permute_crs(row_ptr_g, col_g, val_g, perm)
    perm = np.asarray(perm, dtype=np.int64)
    # colour of a B-index: colours are maximal ascending runs of perm (ascending original index inside a colour)
    starts = np.concatenate([[0], np.nonzero(np.diff(perm) < 0)[0] + 1, [n]]).astype(np.int64)
    if len(starts) - 1 != n_colors:
        raise ValueError("colour runs of the ordering do not match its colour count")
    owned_global = np.nonzero((perm >= block_begin) & (perm < block_end))[0].astype(np.int64)
    a = np.concatenate([np.arange(rpB[q], rpB[q + 1]) for q in owned_global]) if len(owned_global) else np.zeros(0, np.int64)
    lens = (rpB[owned_global + 1] - rpB[owned_global]).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cg = colB[a].astype(np.int64); v = valB[a].astype(np.float64)
    base = plan_owned(comm, n, owned_global, cg)
    rows_g = np.repeat(owned_global, lens)
    color_of = lambda q: np.searchsorted(starts, q, side="right") - 1    # noqa: E731
    color_ptr = np.concatenate([[0], np.cumsum(np.bincount(color_of(owned_global), minlength=n_colors))]).astype(np.int32)
    sco = np.zeros((len(base.nbr), n_colors + 1), dtype=np.int32)
    rco = np.zeros((len(base.nbr), n_colors + 1), dtype=np.int32)
    for k in range(len(base.nbr)):
        s_glob = owned_global[base.send_idx[base.send_off[k]:base.send_off[k + 1]]]
        r_glob = base.halo_global[base.recv_off[k]:base.recv_off[k + 1]]
        sco[k] = base.send_off[k] + np.concatenate([[0], np.cumsum(np.bincount(color_of(s_glob), minlength=n_colors))])
        rco[k] = base.recv_off[k] + np.concatenate([[0], np.cumsum(np.bincount(color_of(r_glob), minlength=n_colors))])
    lower, upper, dg = cg < rows_g, cg > rows_g, cg == rows_g

    def sub(mask):
        cnt = np.bincount(np.repeat(np.arange(len(owned_global)), lens)[mask], minlength=len(owned_global))
        return (np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32), base.col_local[mask].astype(np.int32), v[mask])

    diag = np.zeros(len(owned_global))
    diag[np.repeat(np.arange(len(owned_global)), lens)[dg]] = v[dg]
    return ColoredPlan(base, perm.astype(np.int32), owned_global, n_colors, color_ptr, sco, rco, rp, base.col_local, v,
                       sub(lower), sub(upper), diag)


def upload_colored(handle: capi.Handle, cp: ColoredPlan, fmt=capi.FMT_CSR_EXACT, with_LU=True) -> None:
    """Local matrix (+ crs_L / crs_U), exchange plan and colour structure of a ColoredPlan into the library."""
    b = cp.base
    nc = b.n_loc + b.n_halo
    handle.upload_crs(cp.row_ptr, cp.col_local, cp.val, capi.MAT_A, fmt, n_cols=nc)
    if with_LU:
        handle.upload_crs(*cp.L, capi.MAT_L, capi.FMT_CSR_EXACT, n_cols=nc)
        handle.upload_crs(*cp.U, capi.MAT_U, capi.FMT_CSR_EXACT, n_cols=nc)
    arr = lambda x: np.ascontiguousarray(x, dtype=np.int32)          # noqa: E731
    nbr, so, ro, si = arr(b.nbr), arr(b.send_off), arr(b.recv_off), arr(b.send_idx)
    handle.check(handle.L.its_dist_setup(handle.h, len(nbr), nbr.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p),
                                         si.ctypes.data_as(C.c_void_p), ro.ctypes.data_as(C.c_void_p)), "its_dist_setup")
    cptr, sco, rco = arr(cp.color_ptr), arr(cp.send_color_off), arr(cp.recv_color_off)
    handle.check(handle.L.its_dist_set_colors(handle.h, cp.n_colors, cptr.ctypes.data_as(C.c_void_p),
                                              sco.ctypes.data_as(C.c_void_p), rco.ctypes.data_as(C.c_void_p)), "its_dist_set_colors")


def exchange_color_host(comm, cp: ColoredPlan, x_ext: np.ndarray, color: int) -> None:
    """Host reference of the per-colour exchange (its_dist_exchange): publishes the owned boundary values of one
    colour (all colours: color < 0) to the neighbours' halos, in place.  CPU tests only."""
    b = cp.base
    k_n = len(b.nbr)
    if color < 0:
        s0, s1, r0, r1 = b.send_off[:-1], b.send_off[1:], b.recv_off[:-1], b.recv_off[1:]
    else:
        s0, s1 = cp.send_color_off[:, color], cp.send_color_off[:, color + 1]
        r0, r1 = cp.recv_color_off[:, color], cp.recv_color_off[:, color + 1]
    segs = [x_ext[b.send_idx[s0[k]:s1[k]]] for k in range(k_n)]
    so = np.concatenate([[0], np.cumsum([len(sg) for sg in segs])]).astype(np.int64)
    meta = np.concatenate([[k_n], b.nbr, so]).astype(np.int64)
    data = np.concatenate(segs) if segs else np.zeros(0)
    blobs = comm.allgather_bytes(meta.tobytes() + np.ascontiguousarray(data, dtype=np.float64).tobytes())
    for k, r in enumerate(b.nbr):
        blob = blobs[int(r)]
        nn = int(np.frombuffer(blob[:8], dtype=np.int64)[0])
        hdr = np.frombuffer(blob[8:8 * (2 + 2 * nn)], dtype=np.int64)
        their_nbr, their_off = hdr[:nn], hdr[nn:]
        payload = np.frombuffer(blob[8 * (2 + 2 * nn):], dtype=np.float64)
        j = int(np.nonzero(their_nbr == b.rank)[0][0])
        seg = payload[their_off[j]:their_off[j + 1]]
        assert len(seg) == r1[k] - r0[k]
        x_ext[b.n_loc + r0[k]: b.n_loc + r1[k]] = seg
