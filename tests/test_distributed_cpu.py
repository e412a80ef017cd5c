"""CPU, world_size 2 and 3 over gloo: the row-block partition plan (iterative_solvers_amd/distributed.py)
— local column renumbering, halo receive ranges, send lists — reproduces the global SpMV and a
global CG when each rank multiplies its own slab (oracle kernels as the local arithmetic) and halo
entries / dot products cross ranks.  Also the TCP rendezvous used by bench.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, kind, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle_py as orc
    from iterative_solvers_amd import distributed as D, matgen
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc.set_threads(1)

    class Gloo:
        def __init__(self):
            self.rank, self.world = rank, world

        def allgather_bytes(self, b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out

    comm = Gloo()
    try:
        if kind == "lap":
            N, NZ = 6, 12
            rpg, cg, vg = matgen.laplace3d_7pt(N, nz=NZ)
            cuts = [0, 5, 12] if world == 2 else [0, 3, 8, 12]
            z0, z1 = cuts[rank], cuts[rank + 1]
            rp, colg, val = matgen.laplace3d_7pt_slab(N, z0, z1, NZ)
            row_begin = z0 * N * N
        else:
            rpg, cg, vg = matgen.powerlaw(900)
            ng = len(rpg) - 1
            cuts = [0, 400, 900] if world == 2 else [0, 250, 600, 900]
            r0, r1 = cuts[rank], cuts[rank + 1]
            a, b = rpg[r0], rpg[r1]
            rp = (rpg[r0:r1 + 1] - a).astype(np.int32); colg = cg[a:b].astype(np.int64); val = vg[a:b].copy()
            row_begin = r0
        n_global = len(rpg) - 1
        n_loc = len(rp) - 1
        plan = D.plan_partition(comm, n_global, row_begin, n_loc, colg)
        assert plan.col_local.max(initial=0) < n_loc + plan.n_halo
        # distributed SpMV == global SpMV (bit-exact: same per-row order)
        xg = np.random.default_rng(7).uniform(-1, 1, n_global)
        x_ext = D.exchange_halo_host(comm, plan, xg[row_begin:row_begin + n_loc].copy())
        assert np.array_equal(x_ext[n_loc:], xg[plan.halo_global])
        y = orc.spmv(rp, plan.col_local, val, x_ext)
        yg = orc.spmv(rpg, cg, vg, xg)
        assert np.array_equal(y, yg[row_begin:row_begin + n_loc])

        # distributed CG (cg.hpp:40-281) == global CG up to the reduction order
        def gsum(v):
            parts = [np.frombuffer(b, dtype=np.float64)[0] for b in comm.allgather_bytes(np.array([v]).tobytes())]
            return float(sum(parts))

        b_loc = np.ones(n_loc); x = np.ones(n_loc)
        r = b_loc - orc.spmv(rp, plan.col_local, val, D.exchange_halo_host(comm, plan, x))
        p = r.copy(); rr = gsum(orc.dot(r, r)); hist = []
        for _ in range(25):
            qv = orc.spmv(rp, plan.col_local, val, D.exchange_halo_host(comm, plan, p))
            alpha = rr / gsum(orc.dot(qv, p))
            x = orc.sum_vectors(x, p, alpha); r = orc.subtract_vectors(r, qv, alpha)
            rr_new = gsum(orc.dot(r, r)); hist.append(np.sqrt(rr_new))
            p = orc.sum_vectors(r, p, rr_new / rr); rr = rr_new
        ref = orc.solve(rpg, cg, vg, np.ones(n_global), np.ones(n_global), orc.CG, fixed_iters=25, max_iters=25)
        exp = ref["all_values"][:25] if len(ref["all_values"]) >= 25 else ref["history"]
        k = min(len(exp), 25)
        assert np.allclose(hist[:k], exp[:k], rtol=1e-9, atol=1e-12 * ref["r0_two"])
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "lap"), (3, "lap"), (2, "pl"), (3, "pl")])
def test_partition_plan_gloo(world, kind):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world * 7 + (3 if kind == "pl" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(s == "ok" for _, s in res), res


def _colored_worker(rank, world, port, kind, q):
    """Gauss-Seidel across ranks in global multicolour order (distributed.plan_colored): every colour is swept by all
    ranks, then its boundary values are exchanged.  Must reproduce the reference's natural-order sweep on the
    colour-permuted global matrix (what the oracle runs) — the arithmetic per row is the same."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle_py as orc
    from iterative_solvers_amd import capi, distributed as D, matgen
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc.set_threads(1)

    class Gloo:
        def __init__(self):
            self.rank, self.world = rank, world

        def allgather_bytes(self, b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out

    comm = Gloo()
    try:
        rpg, cg, vg = matgen.laplace3d_7pt(6, nz=12) if kind == "lap" else matgen.powerlaw(900)
        n = len(rpg) - 1
        cuts = np.linspace(0, n, world + 1).astype(int)
        cp = D.plan_colored(comm, rpg, cg, vg, int(cuts[rank]), int(cuts[rank + 1]))
        b_ = cp.base
        n_loc = b_.n_loc
        # every rank's rows, L, U and diagonal are those of B = Q A Q^T
        perm, nc = capi.multicolor_order(rpg, cg)
        rpB, colB, valB = capi.permute_crs(rpg, cg, vg, perm)
        assert cp.n_colors == nc and cp.color_ptr[-1] == n_loc
        ref = orc.solve(rpB, colB, valB, np.ones(n), np.ones(n), orc.GAUSS_SEIDEL, fixed_iters=6, max_iters=6)
        Lp, Lc, Lv = cp.L; Up, Uc, Uv = cp.U
        x = np.zeros(n_loc + b_.n_halo)                                  # the reference's GS starts from Solver::x = 0
        rhs_b = np.ones(n_loc)
        for _ in range(6):
            tmp = rhs_b - orc.spmv(Up, Uc, Uv, x)                        # gauss_seidel.hpp:144,177 (halo of x is current)
            for c in range(cp.n_colors):
                for i in range(cp.color_ptr[c], cp.color_ptr[c + 1]):   # kernels.hpp:460-472, rows of one colour
                    s_ = 0.0
                    for k in range(Lp[i], Lp[i + 1]):
                        s_ += Lv[k] * x[Lc[k]]
                    x[i] = (tmp[i] - s_) / cp.diag[i]
                D.exchange_color_host(comm, cp, x, c)
        want = ref["x_star"][cp.owned_global]
        assert np.allclose(x[:n_loc], want, rtol=1e-12, atol=1e-13), float(np.max(np.abs(x[:n_loc] - want)))
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "lap"), (3, "lap"), (3, "pl")])
def test_colored_gauss_seidel_partition_gloo(world, kind):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world * 11 + (5 if kind == "pl" else 0)
    procs = [ctx.Process(target=_colored_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(s == "ok" for _, s in res), res


def _rdzv_worker(rank, world, key, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=key, RANK=str(rank), WORLD_SIZE=str(world))
    from iterative_solvers_amd.rendezvous import Rendezvous
    r = Rendezvous(rank, world, timeout=60)
    got = r.broadcast(b"id" * 64 if rank == 0 else None)
    r.barrier()
    m = r.allreduce_max(float(rank))
    ag = r.allgather_bytes(bytes([rank]) * (rank + 1))
    r.close()
    q.put((rank, got == b"id" * 64 and m == world - 1 and ag == [bytes([k]) * (k + 1) for k in range(world)]))


def test_tcp_rendezvous_three_ranks():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = str(40000 + os.getpid() % 5000)
    procs = [ctx.Process(target=_rdzv_worker, args=(r, 3, key, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(ok for _, ok in res), res
