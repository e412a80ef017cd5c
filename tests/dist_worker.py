"""One rank of the multi-process GPU test (tests/test_gpu_distributed.py).  All ranks share GPU 0 and
use the library's HOST test transport: halo entries and scalars cross ranks through the TCP
rendezvous, everything else (pack, interior/boundary split SpMV, fused updates) runs on the GPU
exactly as with RCCL.  Writes <out>.<rank>.json."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from iterative_solvers_amd import capi, distributed as D, matgen  # noqa: E402
from iterative_solvers_amd.rendezvous import Rendezvous  # noqa: E402

def main():
    kind, solver, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rdzv = Rendezvous()
    rank, world = rdzv.rank, rdzv.world

    if kind == "lap":
        N, NZ = 12, 24
        cuts = np.linspace(0, NZ, world + 1).astype(int)
        cuts[1:-1] += np.arange(1, world) % 2                          # uneven slabs
        z0, z1 = int(cuts[rank]), int(cuts[rank + 1])
        rp, colg, val = matgen.laplace3d_7pt_slab(N, z0, z1, NZ)
        row_begin, n_global = z0 * N * N, NZ * N * N
        diag = np.full(len(rp) - 1, 6.0)
    else:
        rpg, cg, vg = matgen.powerlaw(6000)
        n_global = 6000
        cuts = np.linspace(0, n_global, world + 1).astype(int)
        r0, r1 = int(cuts[rank]), int(cuts[rank + 1])
        a, b = rpg[r0], rpg[r1]
        rp = (rpg[r0:r1 + 1] - a).astype(np.int32); colg = cg[a:b].astype(np.int64); val = vg[a:b].copy()
        row_begin = r0
        rows = np.repeat(np.arange(r0, r1), np.diff(rp))
        diag = np.zeros(r1 - r0); m = rows == colg; diag[rows[m] - r0] = val[m]
    n_loc = len(rp) - 1
    h = capi.Handle(0)
    D.init_comm_host(h, rank, world, rdzv)
    owned = None
    if solver in ("gs", "gmres_pgs"):
        # triangular sweeps across ranks: global multicolour order (distributed.plan_colored); every rank builds the
        # global matrix here (test sizes) and owns the rows of its contiguous block of ORIGINAL rows
        rpg, cg, vg = matgen.laplace3d_7pt(12, nz=24) if kind == "lap" else matgen.powerlaw(6000)
        cp = D.plan_colored(rdzv, rpg, cg, vg, row_begin, row_begin + n_loc)
        D.upload_colored(h, cp)
        plan = cp.base
        diag = cp.diag
        owned = cp.owned_global
    else:
        plan = D.upload_slab(h, rdzv, n_global, row_begin, rp, colg, val, capi.FMT_CSR_EXACT)
    for kv in filter(None, os.environ.get("ITS_TEST_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        h.set_option(k, int(v))
    h.upload_vectors(diag, np.ones(n_loc), np.ones(n_loc))
    r0_inf, r0_two = h.init_residual()
    kindmap = {"cg": capi.CG, "jacobi": capi.JACOBI, "gmres": capi.GMRES, "gs": capi.GAUSS_SEIDEL, "gmres_pgs": capi.GMRES}
    h.solver_setup(kindmap[solver], restart_len=10, jacobi_mode=capi.JACOBI_REF,
                   precond=capi.PRECOND_GS if solver == "gmres_pgs" else capi.PRECOND_NONE,
                   gs_order=capi.GS_MULTICOLOR if owned is not None else capi.GS_LEVEL)
    trips = 30
    hist = []
    for it in range(trips):
        hist.append(h.iterate(it))
        if solver in ("gmres", "gmres_pgs") and (it + 1) % 10 == 0 and it + 1 < trips:
            h.gmres_restart(it)
    x = h.download_x(n_loc, trips)
    json.dump({"rank": rank, "row_begin": row_begin, "n_loc": n_loc, "n_halo": plan.n_halo, "nbr": plan.nbr.tolist(),
               "r0_inf": r0_inf, "r0_two": r0_two, "hist": hist, "x": x.tolist(),
               "owned": owned.tolist() if owned is not None else None}, open(f"{out}.{rank}.json", "w"))
    rdzv.barrier()
    h.close(); rdzv.close()


if __name__ == "__main__":
    main()
