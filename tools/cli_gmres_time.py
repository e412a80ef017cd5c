#!/usr/bin/env python3
"""Wall time of `its_solve -gm` (dense Givens formulation of the reference) with the incremental and the full products."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from helpers import mtx_path
exe = os.path.join(ROOT, "iterative_solvers_amd", "host", "its_solve")
for m, restart in (("lap2d_64", 100), ("lap2d_256", 100), ("lap2d_256", 250)):
    for full in (0, 1):
        env = dict(os.environ, ITS_GMRES_RESTART_LEN=str(restart), ITS_TOL="1e-30", ITS_MAX_ITERS="500")
        if full: env["ITS_GMRES_DENSE_FULL"] = "1"
        t0 = time.time()
        subprocess.run([exe, mtx_path(m), "-gm", "-scale", "none"], env=env, capture_output=True, text=True, check=True)
        print("%s restart %d, 500 trips, %s products: %.2f s" % (m, restart, "full" if full else "incremental", time.time() - t0), flush=True)
