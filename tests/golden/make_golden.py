#!/usr/bin/env python3
"""Regenerates the golden fixtures of tests/golden/ from the UNMODIFIED compiled reference
(oracle/_ref/iterative_solvers_cpu_m*, built by `make -C oracle ref` from /root/reference).

Runs only in the build container (the reference does not travel to the GPU box).  Output:
  * <name>.mtx                     small Matrix Market inputs (written by iterative_solvers_amd.matgen)
  * reference_histories.json       for each (matrix, solver, scale, restart, precond): the printed
                                   residual history (%.16g text -> float), the convergence line,
                                   the stopping value and the printed final residual.
  * reference_xstar.npz            for the same keys: Solver::x_star as the reference holds it after save_x_star
                                   (dumped by oracle/_ref/ref_xstar_m*, our driver around the reference's own
                                   translation unit — its stdout is asserted equal to the unmodified binary's);
                                   systems of more than 4096 rows keep every 16th entry plus sum and 2-norm.
Long histories (stalled Jacobi / GS runs that hit MAX_ITERS=5000) keep the first 300 lines and the
line count.  OMP_NUM_THREADS=1: deterministic reduction order.

Matrices in BIG are the sizes SURVEY.md §8c lists (2-D 64^2, 3-D 16^3, 27-pt 12^3) and BASELINE.json configs[0]
(2-D 256^2); their .mtx files are written to a temporary directory, not committed: iterative_solvers_amd.matgen
regenerates them bit for bit wherever a test needs them.
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import matgen  # noqa: E402

MATS = {
    "lap2d_16": lambda: matgen.laplace2d_5pt(16),
    "lap3d_8": lambda: matgen.laplace3d_7pt(8),
    "cd27_6": lambda: matgen.convdiff3d_27pt(6),
    "pl_200": lambda: matgen.powerlaw(200),
}
BIG = {
    "lap2d_64": lambda: matgen.laplace2d_5pt(64),
    "lap3d_16": lambda: matgen.laplace3d_7pt(16),
    "cd27_12": lambda: matgen.convdiff3d_27pt(12),
    "lap2d_256": lambda: matgen.laplace2d_5pt(256),
}
CASES = []
for m in BIG:
    for s in ("-cg", "-j", "-gs", "-gm"):
        if m == "cd27_12" and s == "-cg":
            continue                                   # nonsymmetric: CG is not defined on it
        if m == "lap2d_256" and s in ("-gs", "-gm"):
            continue                                   # minutes of serial sweeps / dense Givens products: not needed
        CASES.append((m, s, "none", 100, ""))
    CASES.append((m, "-cg" if m != "cd27_12" else "-gm", "diag", 100, ""))
for m in MATS:
    for s in ("-cg", "-gs", "-j", "-gm"):
        for sc in ("none", "diag", "max"):
            CASES.append((m, s, sc, 100, ""))
    CASES.append((m, "-gm", "none", 10, ""))
    CASES.append((m, "-gm", "diag", 10, "gs"))
    CASES.append((m, "-gm", "none", 50, ""))

LINE = re.compile(r"\|\|A\*x_(\d+) - b\|\|_infty = (\S+)")


def main():
    import tempfile
    import numpy as np
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = {}
    xs = {}
    tmp = tempfile.mkdtemp(prefix="its_golden_")
    paths = {}
    for name, gen in MATS.items():
        paths[name] = os.path.join(HERE, name + ".mtx")
        matgen.write_mtx(paths[name], *gen())
    for name, gen in BIG.items():
        paths[name] = os.path.join(tmp, name + ".mtx")
        matgen.write_mtx(paths[name], *gen())
    for (m, s, sc, restart, pre) in CASES:
        exe = os.path.join(ROOT, "oracle", "_ref", f"iterative_solvers_cpu_m{restart}")
        cmd = [exe, paths[m], s]
        cmd += ["-p", pre] if pre else ["-scale", sc]     # the reference CLI takes at most one option pair
        txt = subprocess.run(cmd, env=env, capture_output=True, text=True, check=True).stdout
        hist = [float(v) for _, v in LINE.findall(txt)]
        key = f"{m}|{s}|{sc if not pre else 'diag'}|m{restart}|p{pre}"
        # x_star through the driver (restart lengths 100 and 10 are built); its history must be the binary's
        xexe = os.path.join(ROOT, "oracle", "_ref", f"ref_xstar_m{restart}")
        if os.path.exists(xexe):
            xfile = os.path.join(tmp, "xstar.bin")
            xtxt = subprocess.run([xexe] + cmd[1:], env=dict(env, ITS_XSTAR_OUT=xfile), capture_output=True, text=True, check=True).stdout
            assert LINE.findall(xtxt) == LINE.findall(txt), key
            x = np.fromfile(xfile)
            if len(x) <= 4096:
                xs[key] = x
            else:
                with np.errstate(all="ignore"):
                    xs[key] = np.concatenate([x[::16], [np.sum(x), np.sqrt(np.sum(x * x))]])
        conv = re.search(r"solver converged in: (\d+) iterations", txt)
        stop = re.search(r"is: (\S+)\n", txt)
        fin = re.search(r"\|\|A\*x_star - b\|\|_infty = (\S+)\.\n", txt)
        out[key] = {"n_lines": len(hist), "history": hist[:300], "converged": bool(conv),
                    "iters": int(conv.group(1)) if conv else 5000, "stopping": float(stop.group(1)),
                    "final": float(fin.group(1)) if fin else None}
    with open(os.path.join(HERE, "reference_histories.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(HERE, "reference_xstar.npz"), **xs)
    print(f"{len(out)} cases written, {len(xs)} with x_star")


if __name__ == "__main__":
    main()
