"""GPU: the communication-reducing Krylov variants (csrc/its_krylov_cr.hip) — pipelined CG
("cg_variant" 1) and CGS-2 Arnoldi ("gmres_ortho" 1).  Opt-in and outside the parity contract (other recurrences, so
other residual histories): what is asserted is that they solve the same system — the converged x within 1e-8 of the
reference algorithm's, trip count within 5 % at TOL = 1e-10 — on one GPU and row-partitioned over the HOST transport."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import MATRICES
from iterative_solvers_amd import capi, solver as S

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _solve(name, solver, options, restart=50, max_iters=2000):
    rp, col, val = MATRICES[name]()
    n = len(rp) - 1
    args = S.Args(solver, S.LoopParams(max_iters=max_iters, tol=1e-10, gmres_restart_len=restart), S.Flags(), "", "none")
    sol = S.Solver(solver, device=0)
    try:
        for k, v in options.items():
            sol.handle.set_option(k, v)
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        S.solve(args, sol)
        return args.loop_params.iter_count, args.flags.convergence_flag, sol.x_star.copy(), \
            args.normed_residuals[:args.loop_params.residual_count].copy()
    finally:
        sol.handle.close()


@pytest.mark.parametrize("name", ["lap2d_64", "lap3d_16", "lap2d_16"])
def test_pipelined_cg_converges_to_the_same_solution(name):
    it0, ok0, x0, h0 = _solve(name, "conjugate-gradient", {})
    it1, ok1, x1, h1 = _solve(name, "conjugate-gradient", {"cg_variant": 1})
    assert ok0 and ok1, (ok0, ok1, it0, it1)
    assert abs(it1 - it0) <= max(1, round(0.05 * it0)), (it0, it1)
    assert np.max(np.abs(x1 - x0)) <= 1e-8 * max(1.0, np.max(np.abs(x0)))
    # the recurrences are equivalent in exact arithmetic: early histories agree closely
    k = min(10, len(h0), len(h1))
    assert np.all(np.abs(h1[:k] - h0[:k]) <= 1e-9 * h0[0])


@pytest.mark.parametrize("name,restart", [("lap3d_16", 50), ("cd27_12", 10), ("cd27_12", 50), ("lap2d_64", 100)])
def test_cgs2_gmres_converges_to_the_same_solution(name, restart):
    it0, ok0, x0, h0 = _solve(name, "gmres", {}, restart=restart)
    it1, ok1, x1, h1 = _solve(name, "gmres", {"gmres_ortho": 1}, restart=restart)
    assert ok0 and ok1, (ok0, ok1, it0, it1)
    assert abs(it1 - it0) <= max(1, round(0.05 * it0)), (it0, it1)
    assert np.max(np.abs(x1 - x0)) <= 1e-8 * max(1.0, np.max(np.abs(x0)))
    k = min(10, len(h0), len(h1))
    assert np.all(np.abs(h1[:k] - h0[:k]) <= 1e-9 * h0[0])


@pytest.mark.parametrize("solver,opt", [("cg", "cg_variant=1"), ("gmres", "gmres_ortho=1")])
def test_variants_row_partitioned_over_host_transport(tmp_path, solver, opt):
    """Two processes on one GPU: the single reduction point of a pipelined-CG trip and the batched CGS-2 reductions
    cross ranks as all-reduces; the 30 recorded values must equal the one-process run of the same variant up to the
    reduction order."""
    world = 2
    out = str(tmp_path / "res")
    port = 35000 + os.getpid() % 3000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ITS_RDZV_DIR=str(tmp_path), ITS_TEST_OPTIONS=opt)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), "lap", solver, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=110)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [json.load(open(f"{out}.{r}.json")) for r in range(world)]
    from iterative_solvers_amd import matgen
    rp, col, val = matgen.laplace3d_7pt(12, nz=24)
    n = len(rp) - 1
    h = capi.Handle(0)
    try:
        k, v = opt.split("=")
        h.set_option(k, int(v))
        h.upload_crs(rp, col, val)
        h.upload_vectors(np.full(n, 6.0), np.ones(n), np.ones(n))
        h.init_residual()
        h.solver_setup(capi.CG if solver == "cg" else capi.GMRES, restart_len=10)
        hist = []
        for it in range(30):
            hist.append(h.iterate(it))
            if solver == "gmres" and (it + 1) % 10 == 0 and it + 1 < 30:
                h.gmres_restart(it)
        x = h.download_x(n, 30)
    finally:
        h.close()
    got = np.array(res[0]["hist"]); exp = np.array(hist)
    assert res[0]["hist"] == res[1]["hist"]
    assert np.all(np.abs(got - exp) <= 1e-9 * np.maximum(exp, 1e-3 * exp[0]))
    xg = np.concatenate([np.array(r["x"]) for r in res])
    assert np.max(np.abs(xg - x)) <= 1e-9 * max(1.0, np.max(np.abs(x)))
