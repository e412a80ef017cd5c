#!/usr/bin/env python3
"""Wall time of `its_solve` on BASELINE.json configs[0] (2-D 5-pt Laplacian 256x256 .mtx), every solver, default settings."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from helpers import mtx_path
exe = os.path.join(ROOT, "iterative_solvers_amd", "host", "its_solve")
m = sys.argv[1] if len(sys.argv) > 1 else "lap2d_256"
if m.startswith("gen:"):                                   # gen:lap3d:64 | gen:cd27:40 | gen:pl:200000 | gen:lap2d:1000
    from iterative_solvers_amd import matgen
    import tempfile
    kind, size = m.split(":")[1], int(m.split(":")[2])
    gen = {"lap3d": matgen.laplace3d_7pt, "cd27": matgen.convdiff3d_27pt, "pl": matgen.powerlaw, "lap2d": matgen.laplace2d_5pt}[kind]
    path = os.path.join(tempfile.gettempdir(), "cli_time_%s_%d.mtx" % (kind, size))
    if not os.path.exists(path):
        matgen.write_mtx(path, *gen(size))
    mtx_path = lambda _m: path
for flags in (["-j"], ["-gs"], ["-cg"], ["-gm"], ["-gm", "-p", "gs"]):
    t0 = time.time()
    out = subprocess.run([exe, mtx_path(m)] + flags, capture_output=True, text=True, env=dict(os.environ, ITS_MAX_ITERS=os.environ.get("ITS_MAX_ITERS", "5000")))
    dt = time.time() - t0
    it = re.search(r"(converged in: \d+ iterations|did not converge[^\n]*)", out.stdout)
    tl = [l for l in out.stdout.splitlines() if "time" in l.lower()][:6]
    print("%-12s %.2f s  rc=%d  %s" % (" ".join(flags), dt, out.returncode, it.group(1) if it else ""), flush=True)
    for l in tl: print("      ", l)
