"""GPU: the multi-GPU code path on ONE device.
 * HOST transport, 2 and 3 processes sharing GPU 0: partition plan, pack kernel, interior/boundary
   split SpMV with accumulated partials, halo unpack and scalar all-reduce all execute on the GPU;
   only the wire is replaced (TCP instead of RCCL).  Result must equal the single-process oracle.
 * RCCL transport with world = 1: communicator creation, all-reduce on the compute stream and the
   split launch sequence run through RCCL itself; bit-identical to the non-distributed path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as orc
from iterative_solvers_amd import capi, matgen

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _global_problem(kind):
    if kind == "lap":
        return matgen.laplace3d_7pt(12, nz=24)
    return matgen.powerlaw(6000)


@pytest.mark.parametrize("world,kind,solver", [(2, "lap", "cg"), (3, "lap", "cg"), (3, "pl", "cg"),
                                                (2, "pl", "jacobi"), (3, "lap", "gmres")])
def test_host_transport_multi_process(tmp_path, world, kind, solver):
    out = str(tmp_path / "res")
    port = 30000 + os.getpid() % 3000 + world
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ITS_RDZV_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), kind, solver, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=110)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [json.load(open(f"{out}.{r}.json")) for r in range(world)]
    rp, col, val = _global_problem(kind)
    n = len(rp) - 1
    orc.set_threads(1)
    K = {"cg": orc.CG, "jacobi": orc.JACOBI, "gmres": orc.GMRES}[solver]
    ref = orc.solve(rp, col, val, np.ones(n), np.ones(n), K, jacobi_mode=1, restart_len=10, fixed_iters=30, max_iters=30)
    x = np.concatenate([np.array(r["x"]) for r in res])
    assert sum(r["n_loc"] for r in res) == n
    for r in res:                                  # every rank saw the same global scalars
        assert r["hist"] == res[0]["hist"]
        assert abs(r["r0_two"] - ref["r0_two"]) <= 1e-13 * ref["r0_two"] and r["r0_inf"] == ref["r0_inf"]
    exp = ref["all_values"]
    k = min(len(exp), 30)
    got = np.array(res[0]["hist"][:k])
    assert np.all(np.abs(got - exp[:k]) <= 1e-10 * np.maximum(exp[:k], 1e-2 * ref["r0_two"]))
    assert np.max(np.abs(x - ref["x_star"])) <= 1e-10 * max(1.0, np.max(np.abs(ref["x_star"])))
    if world == 3 and kind == "lap":
        assert res[1]["nbr"] == [0, 2] and res[0]["nbr"] == [1]
        assert res[1]["n_halo"] == 2 * 144


@pytest.mark.parametrize("world,kind,solver", [(2, "lap", "gs"), (3, "lap", "gs"), (3, "pl", "gs"), (2, "lap", "gmres_pgs"),
                                                (3, "pl", "gmres_pgs")])
def test_host_transport_triangular_sweeps_across_ranks(tmp_path, world, kind, solver):
    """Gauss-Seidel and GS-preconditioned GMRES row-partitioned over 2 / 3 processes (HOST transport, one GPU): the
    sweep runs in GLOBAL multicolour order, one exchange of a colour's boundary values after each colour
    (its_dist_set_colors).  Reference: the oracle's natural-order sweep on the same colour-permuted global matrix
    B = Q A Q^T — identical arithmetic per row, so Gauss-Seidel iterates must agree to rounding of nothing at all
    (bit-exact x), GMRES within the usual reduction-order tolerance."""
    out = str(tmp_path / "res")
    port = 33000 + os.getpid() % 3000 + world
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ITS_RDZV_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), kind, solver, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=110)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [json.load(open(f"{out}.{r}.json")) for r in range(world)]
    rp, col, val = _global_problem(kind)
    n = len(rp) - 1
    perm, nc = capi.multicolor_order(rp, col)
    rpB, colB, valB = capi.permute_crs(rp, col, val, perm)
    orc.set_threads(1)
    if solver == "gs":
        ref = orc.solve(rpB, colB, valB, np.ones(n), np.ones(n), orc.GAUSS_SEIDEL, fixed_iters=30, max_iters=30)
    else:
        ref = orc.solve(rpB, colB, valB, np.ones(n), np.ones(n), orc.GMRES, precond=orc.PRECOND["gs"], restart_len=10,
                        fixed_iters=30, max_iters=30)
    xB = np.empty(n)
    for r in res:
        xB[np.array(r["owned"], dtype=np.int64)] = np.array(r["x"])
        assert r["hist"] == res[0]["hist"]
    exp = ref["all_values"]
    k = min(len(exp), 30)
    got = np.array(res[0]["hist"][:k])
    assert np.all(np.abs(got - exp[:k]) <= 1e-10 * np.maximum(exp[:k], 1e-2 * ref["r0_two"]))
    if solver == "gs":
        assert np.array_equal(xB, ref["x_star"])          # no reduction touches a Gauss-Seidel iterate
    else:
        assert np.max(np.abs(xB - ref["x_star"])) <= 1e-10 * max(1.0, np.max(np.abs(ref["x_star"])))


def test_rccl_world_one_matches_single_gpu():
    import ctypes as C
    rp, col, val = matgen.laplace3d_7pt(16)
    n = len(rp) - 1
    hists = []
    for use_rccl in (False, True):
        h = capi.Handle(0)
        try:
            if use_rccl:
                ids = C.create_string_buffer(256)
                h.check(h.L.its_comm_unique_id(ids), "its_comm_unique_id")
                h.check(h.L.its_comm_unique_id(C.byref(ids, 128)), "its_comm_unique_id")
                h.check(h.L.its_comm_init(h.h, 0, 1, ids), "its_comm_init")
            h.upload_crs(rp, col, val)
            if use_rccl:
                h.check(h.L.its_dist_setup(h.h, 0, None, None, None, None), "its_dist_setup")
            h.upload_vectors(np.full(n, 6.0), np.ones(n), np.ones(n))
            r0 = h.init_residual()
            h.solver_setup(capi.CG)
            hists.append((r0, [h.iterate(i) for i in range(40)], h.download_x(n, 40)))
        finally:
            h.close()
    assert hists[0][0] == hists[1][0]
    assert hists[0][1] == hists[1][1]                      # bit-identical: same reduction tree
    assert np.array_equal(hists[0][2], hists[1][2])


def _periodic_z_laplacian(N, diag=7.0, NZ=None):
    """7-pt stencil on an N x N x NZ grid (NZ = N by default), periodic in z (SPD with diag > 6): the global matrix and the
    one-rank local form whose wrap-around columns are halo entries [n, n + 2 N^2): first the copy of plane NZ-1, then of
    plane 0."""
    NZ = N if NZ is None else NZ
    n = N * N * NZ
    idx = np.arange(n).reshape(NZ, N, N)                     # [z, y, x]
    rows, cols_g, cols_l, vals = [], [], [], []
    P = N * N
    for dz, dy, dx in ((-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0)):
        z, y, x = np.meshgrid(np.arange(NZ), np.arange(N), np.arange(N), indexing="ij")
        zz, yy, xx = z + dz, y + dy, x + dx
        ok = (yy >= 0) & (yy < N) & (xx >= 0) & (xx < N)
        wrap_lo, wrap_hi = zz < 0, zz >= NZ
        zg = zz % NZ
        g = idx[zg[ok], yy[ok], xx[ok]]
        l = g.copy()
        plane_pos = (yy * N + xx)[ok]
        l[wrap_lo[ok]] = n + plane_pos[wrap_lo[ok]]          # copy of plane NZ-1
        l[wrap_hi[ok]] = n + P + plane_pos[wrap_hi[ok]]      # copy of plane 0
        rows.append(idx[ok]); cols_g.append(g); cols_l.append(l)
        vals.append(np.full(g.size, diag if (dz, dy, dx) == (0, 0, 0) else -1.0))
    rows = np.concatenate(rows); cg_ = np.concatenate(cols_g); cl_ = np.concatenate(cols_l); v = np.concatenate(vals)
    def to_crs(c):
        order = np.lexsort((c, rows))
        rp = np.zeros(n + 1, dtype=np.int32); np.add.at(rp, rows + 1, 1); rp = np.cumsum(rp).astype(np.int32)
        return rp, c[order].astype(np.int32), v[order]
    send_idx = np.concatenate([idx[NZ - 1].ravel(), idx[0].ravel()]).astype(np.int32)
    return n, to_crs(cg_), to_crs(cl_), send_idx


@pytest.mark.parametrize("two_entries", [False, True])
@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("N,NZ", [(12, 12), (64, 8)])
def test_rccl_send_recv_loopback_on_one_gpu(overlap, two_entries, N, NZ):
    """The RCCL halo exchange (pack kernel -> grouped ncclSend/ncclRecv on the side stream -> boundary SpMV after
    the halo event) run for real on ONE GPU: a one-rank world whose only neighbour is itself, carrying the
    wrap-around planes of a z-periodic Laplacian.  Checked against the same system as one global matrix.
    (64, 8): the small analogue of a rank's slab of BASELINE configs[4] — 512 x 512 x 64 of the 512^3 grid: few wide
    planes; a plane is 8 row blocks of the ELL kernel, so the period work map is on, as it is at full size)."""
    import ctypes as C
    n, (rpg, colg, valg), (rpl, coll, vall), send_idx = _periodic_z_laplacian(N, NZ=NZ)
    P = N * N
    hists = []
    for loop in (False, True):
        h = capi.Handle(0)
        try:
            if loop:
                h.set_option("dist_overlap", overlap)        # 0: exchange, then one product; 1: beside the interior product
                ids = C.create_string_buffer(256)
                h.check(h.L.its_comm_unique_id(ids), "its_comm_unique_id")
                h.check(h.L.its_comm_unique_id(C.byref(ids, 128)), "its_comm_unique_id")
                h.check(h.L.its_comm_init(h.h, 0, 1, ids), "its_comm_init")
                h.upload_crs(rpl, coll, vall, n_cols=n + 2 * P)
                # one neighbour entry carrying both planes (two runs of rows: packed through the send buffer), or two
                # entries of one plane each (an ascending run each: sent straight out of the vector)
                nbr = np.zeros(2 if two_entries else 1, np.int32)
                off = np.array([0, P, 2 * P] if two_entries else [0, 2 * P], np.int32)
                h.check(h.L.its_dist_setup(h.h, len(nbr), nbr.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                           send_idx.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p)), "its_dist_setup")
            else:
                h.upload_crs(rpg, colg, valg)
            h.upload_vectors(np.full(n, 7.0), np.ones(n), np.zeros(n))
            r0 = h.init_residual()
            h.solver_setup(capi.CG)
            hist = [h.iterate(i) for i in range(25)]
            hists.append((r0, hist, h.download_x(n, 25)))
        finally:
            h.close()
    assert abs(hists[0][0][1] - hists[1][0][1]) <= 1e-13 * hists[0][0][1]
    a, b = np.array(hists[0][1]), np.array(hists[1][1])
    assert np.all(np.abs(a - b) <= 1e-11 * np.maximum(a, 1e-3 * a[0]))
    assert np.max(np.abs(hists[0][2] - hists[1][2])) <= 1e-11 * np.max(np.abs(hists[0][2]))


@pytest.mark.parametrize("overlap", [0, 1])
def test_rccl_colored_sweep_exchange_loopback_on_one_gpu(overlap):
    """The RCCL branch of the per-colour exchange (its_dist_exchange: grouped ncclSend / ncclRecv of one colour's segment
    of the send list) on ONE GPU: a one-rank world that is its own neighbour, Gauss-Seidel in global
    multicolour (red-black) order on a z-periodic Laplacian whose wrap-around columns are halo entries.  After every
    colour the halo copies are refreshed through RCCL, so the sweep is the natural-order sweep on the colour-permuted
    global matrix B = Q A Q^T: the iterate must equal the oracle's on B bit for bit, histories to the usual bar."""
    import ctypes as C
    N, NZ = 10, 8                                            # even NZ: parity colouring is proper across the z wrap
    n, (rpg, colg, valg), _, _ = _periodic_z_laplacian(N, NZ=NZ)
    z, y, x = np.meshgrid(np.arange(NZ), np.arange(N), np.arange(N), indexing="ij")
    color = ((x + y + z) % 2).ravel()
    perm = np.concatenate([np.nonzero(color == 0)[0], np.nonzero(color == 1)[0]]).astype(np.int32)   # B-index -> original row
    rpB, colB, valB = capi.permute_crs(rpg, colg, valg, perm)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    n_red = int(np.count_nonzero(color == 0))
    rowsB = np.repeat(np.arange(n), np.diff(rpB))
    # wrap-around entries: original row and column in planes 0 / NZ-1 of different z
    zo = perm // (N * N)
    wrap = (np.abs(zo[rowsB] - zo[colB]) == NZ - 1)
    S = np.unique(colB[wrap]).astype(np.int64)               # B-rows whose values travel, ascending = colour by colour
    pos = np.full(n, -1, np.int64); pos[S] = np.arange(len(S))
    coll = colB.astype(np.int64).copy(); coll[wrap] = n + pos[colB[wrap]]
    lower, upper, dg = colB < rowsB, colB > rowsB, colB == rowsB

    def sub(mask):
        cnt = np.bincount(rowsB[mask], minlength=n)
        return np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32), coll[mask].astype(np.int32), valB[mask]
    diag = np.zeros(n); diag[rowsB[dg]] = valB[dg]
    orc.set_threads(1)
    trips = 20
    ref = orc.solve(rpB, colB, valB, np.ones(n), np.ones(n), orc.GAUSS_SEIDEL, fixed_iters=trips, max_iters=trips)
    h = capi.Handle(0)
    try:
        h.set_option("dist_overlap", overlap)
        ids = C.create_string_buffer(256)
        h.check(h.L.its_comm_unique_id(ids), "its_comm_unique_id")
        h.check(h.L.its_comm_unique_id(C.byref(ids, 128)), "its_comm_unique_id")
        h.check(h.L.its_comm_init(h.h, 0, 1, ids), "its_comm_init")
        nc = n + len(S)
        h.upload_crs(rpB, coll.astype(np.int32), valB, capi.MAT_A, capi.FMT_CSR_EXACT, n_cols=nc)
        h.upload_crs(*sub(lower), capi.MAT_L, capi.FMT_CSR_EXACT, n_cols=nc)
        h.upload_crs(*sub(upper), capi.MAT_U, capi.FMT_CSR_EXACT, n_cols=nc)
        arr = lambda v: np.ascontiguousarray(v, dtype=np.int32)          # noqa: E731
        nbr, off, si = arr([0]), arr([0, len(S)]), arr(S)
        h.check(h.L.its_dist_setup(h.h, 1, nbr.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                   si.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p)), "its_dist_setup")
        s_red = int(np.count_nonzero(S < n_red))
        cptr, sco = arr([0, n_red, n]), arr([[0, s_red, len(S)]])
        h.check(h.L.its_dist_set_colors(h.h, 2, cptr.ctypes.data_as(C.c_void_p), sco.ctypes.data_as(C.c_void_p),
                                        sco.ctypes.data_as(C.c_void_p)), "its_dist_set_colors")
        h.upload_vectors(diag, np.ones(n), np.ones(n))
        h.init_residual()
        h.solver_setup(capi.GAUSS_SEIDEL, gs_order=capi.GS_MULTICOLOR)
        hist = np.array([h.iterate(i) for i in range(trips)])
        xB = h.download_x(n, trips)
    finally:
        h.close()
    exp = ref["all_values"][:trips]
    assert np.all(np.abs(hist[:len(exp)] - exp) <= 1e-10 * np.maximum(exp, 1e-2 * ref["r0_two"]))
    assert np.array_equal(xB, ref["x_star"])                 # no reduction touches a Gauss-Seidel iterate


def test_bench_global_grid_host_transport():
    """`bench.py --gpus 2 --global-grid 32`: the ranks share ONE 32^3 grid (the configs[4] mode: 512^3 over 8 ranks),
    16 planes of 32^2 each, through the HOST rehearsal transport on the one-GPU box."""
    env = dict(os.environ, ITS_BENCH_TRANSPORT="host", ITS_BENCH_SPAWN_TIMEOUT="100")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--global-grid", "32", "--no-cpu"], env=env, capture_output=True, text=True, timeout=115)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert out["n_gpus"] == 2 and out["config"]["rows_per_gpu"] == 32 * 32 * 16 and out["config"]["grid_mode"] == "global-grid"
    assert "32^3 grid shared by 2 z-slabs of 16 planes" in out["config"]["workload"] and out["scaling"] == "strong"
    assert out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"] and out["regions"] == 5


def test_bench_spawns_its_own_ranks_host_transport():
    """`python bench.py --gpus 2` with no launcher: the parent starts two fresh rank processes itself (it never
    touches the GPU) and relays rank 0's single JSON line.  HOST transport puts both ranks on GPU 0, so the whole
    N > 1 flow — rendezvous, partition plan, halo exchange, all-reduced partials, max-over-ranks timing — runs on
    the one-GPU box; without ITS_BENCH_TRANSPORT the same command is the RCCL path of the scaling bench."""
    env = dict(os.environ, ITS_BENCH_TRANSPORT="host", ITS_BENCH_SPAWN_TIMEOUT="100")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "4",
                        "--warmup", "2", "--grid", "32", "--no-cpu"], env=env, capture_output=True, text=True, timeout=115)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["metric"] == "cg_iters_per_sec" and out["value"] > 0 and out["jacobi"]["value"] > 0


def test_comm_init_timeout_leaves_the_handle_refusing_a_second_try():
    """A rank whose peer never arrives: its_comm_init gives up after ITS_COMM_TIMEOUT_S with ITS_ECOMM and a message that
    names rank and device; the handle then refuses another its_comm_init and its_dist_setup (what a caller may
    do after that error is documented in include/its_hip.h — exit and restart with fresh processes).  Run in a child
    process: the helper thread is still inside the RCCL bootstrap when the process ends."""
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from iterative_solvers_amd import capi
h = capi.Handle(0)
ids = C.create_string_buffer(256)
h.check(h.L.its_comm_unique_id(ids), "id"); h.check(h.L.its_comm_unique_id(C.byref(ids, 128)), "id")
rc = h.L.its_comm_init(h.h, 0, 2, ids)                      # world of 2, rank 1 never joins
msg = h.L.its_last_error(h.h).decode()
rc2 = h.L.its_comm_init(h.h, 0, 2, ids)
msg2 = h.L.its_last_error(h.h).decode()
rc3 = h.L.its_dist_setup(h.h, 0, None, None, None, None)
print("ITSRC", rc, rc2, rc3); print("ITSMSG1", msg); print("ITSMSG2", msg2)
sys.stdout.flush()
import os; os._exit(0)
''' % os.path.dirname(HERE)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ITS_COMM_TIMEOUT_S="3"), capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, p.stdout + p.stderr
    rcs = [ln for ln in p.stdout.splitlines() if ln.startswith("ITSRC")][0].split()[1:]
    assert int(rcs[0]) < 0 and int(rcs[1]) < 0 and int(rcs[2]) < 0, p.stdout
    assert "did not complete within 3 s" in p.stdout and "fresh process" in p.stdout
    assert "timed out" in [ln for ln in p.stdout.splitlines() if ln.startswith("ITSMSG2")][0]
