import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from iterative_solvers_amd import capi, matgen, solver as S
rp, col, val = matgen.convdiff3d_27pt(128)
n = len(rp) - 1
for ahead in (1, 0, 1, 0):
    args = S.Args("gmres", S.LoopParams(max_iters=10**9, tol=0.0, gmres_restart_len=50))
    sol = S.Solver("gmres", device=0)
    sol.handle.set_option("gmres_ahead", ahead)
    S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
    h = sol.handle
    it = 0
    def cycle(trips):
        global it
        for _ in range(trips):
            h.iterate(it); it += 1
            if it % 50 == 0: h.gmres_restart(it)
    cycle(50)
    h.synchronize(); t0 = time.perf_counter(); cycle(150); h.synchronize()
    print("gmres_ahead %d: %.4f ms/trip" % (ahead, (time.perf_counter() - t0) / 150 * 1e3), flush=True)
    h.close()
