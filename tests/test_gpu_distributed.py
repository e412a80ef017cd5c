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
