#!/usr/bin/env python3
"""Regenerates the golden fixtures of tests/golden/ from the UNMODIFIED compiled reference
(oracle/_ref/iterative_solvers_cpu_m*, built by `make -C oracle ref` from /root/reference).

Runs only in the build container (the reference does not travel to the GPU box).  Output:
  * <name>.mtx                     small Matrix Market inputs (written by iterative_solvers_amd.matgen)
  * reference_histories.json       for each (matrix, solver, scale, restart, precond): the printed
                                   residual history (%.16g text -> float), the convergence line,
                                   the stopping value and the printed final residual.
Long histories (stalled Jacobi / GS runs that hit MAX_ITERS=5000) keep the first 300 lines and the
line count.  OMP_NUM_THREADS=1: deterministic reduction order.
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
CASES = []
for m in MATS:
    for s in ("-cg", "-gs", "-j", "-gm"):
        for sc in ("none", "diag", "max"):
            CASES.append((m, s, sc, 100, ""))
    CASES.append((m, "-gm", "none", 10, ""))
    CASES.append((m, "-gm", "diag", 10, "gs"))
    CASES.append((m, "-gm", "none", 50, ""))

LINE = re.compile(r"\|\|A\*x_(\d+) - b\|\|_infty = (\S+)")


def main():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = {}
    for name, gen in MATS.items():
        matgen.write_mtx(os.path.join(HERE, name + ".mtx"), *gen())
    for (m, s, sc, restart, pre) in CASES:
        exe = os.path.join(ROOT, "oracle", "_ref", f"iterative_solvers_cpu_m{restart}")
        cmd = [exe, os.path.join(HERE, m + ".mtx"), s]
        cmd += ["-p", pre] if pre else ["-scale", sc]     # the reference CLI takes at most one option pair
        txt = subprocess.run(cmd, env=env, capture_output=True, text=True, check=True).stdout
        hist = [float(v) for _, v in LINE.findall(txt)]
        conv = re.search(r"solver converged in: (\d+) iterations", txt)
        stop = re.search(r"is: (\S+)\n", txt)
        fin = re.search(r"\|\|A\*x_star - b\|\|_infty = (\S+)\.\n", txt)
        key = f"{m}|{s}|{sc if not pre else 'diag'}|m{restart}|p{pre}"
        out[key] = {"n_lines": len(hist), "history": hist[:300], "converged": bool(conv),
                    "iters": int(conv.group(1)) if conv else 5000, "stopping": float(stop.group(1)),
                    "final": float(fin.group(1)) if fin else None}
    with open(os.path.join(HERE, "reference_histories.json"), "w") as f:
        json.dump(out, f)
    print(f"{len(out)} cases written")


if __name__ == "__main__":
    main()
