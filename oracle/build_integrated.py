#!/usr/bin/env python3
"""Compiles the binding of INTEGRATION.md INTO THE REFERENCE'S OWN SOURCES and links it against libits_hip.so
(container only: needs /root/reference).

What it does: copies the reference's headers, main.cpp and mmio.* into a temporary directory, inserts the
`#ifdef USE_MI355X` bodies below BY LINE NUMBER into solver.hpp (six methods + members) and utility_funcs.hpp
(record_residual_norm), compiles main.cpp with -DUSE_MI355X -Iinclude and links -lits_hip.  main.cpp,
solver_harness.hpp, io_funcs.hpp, structs.hpp, the methods/ headers and mmio stay byte for byte what the reference ships:
the do..while loop that runs is the reference's own (solver_harness.hpp:40-116), its CLI, its Matrix Market reader, its
printing.  No reference text is committed here: this file holds only the inserted lines, the line numbers they go to, and
a short pattern per anchor that must match (a changed reference fails loudly instead of being patched in the wrong place).

  python oracle/build_integrated.py                   # compile + link into a temp dir, report, clean up
                                                      # (tools/check_integration.py is the same call)
  python oracle/build_integrated.py --build-into oracle/_ref     # keep binaries its_integrated_m{10,50,100} there
                                                      # (`make -C oracle integrated`; git-ignored, they travel to the GPU box
                                                      # like oracle/_ref's others; tests/test_gpu_integration.py runs them
                                                      # against the goldens of the unmodified reference)
TEST INFRASTRUCTURE, like everything under oracle/: the product's own host is iterative_solvers_amd/host/.
The binaries are compiled with -DTOL=1e-10 (the compiled-in tolerance of the reference, config.mk), because the golden
histories below 1e-10 * ||r0|| are rounding noise (tests/test_gpu_host_cli.py explains); everything else as oracle/Makefile.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

BINDING_HPP = r'''// its_binding.hpp — what a maintainer adds next to solver.hpp (INTEGRATION.md): the C ABI, the process's one handle
// (main.cpp:362 creates one Solver), the reference's own error convention (printf + exit(1), e.g. solver.hpp:545-548)
#pragma once
#ifdef USE_MI355X
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
extern "C" {
#include "its_hip.h"
}
static its_handle *g_its_handle = nullptr;
#define ITS_CALL(x) do { if ((x) != 0) { printf("ERROR: %s\n", g_its_handle ? its_last_error(g_its_handle) : "libits_hip"); exit(1); } } while (0)
static inline int its_solver_kind(const std::string &s)
{ return s == "jacobi" ? ITS_JACOBI : s == "gauss-seidel" ? ITS_GAUSS_SEIDEL : s == "gmres" ? ITS_GMRES : ITS_CG; }
static inline int its_precond_kind(const std::string &s)
{ return s == "gauss-seidel" ? ITS_PRECOND_GS : s == "jacobi" ? ITS_PRECOND_JACOBI : ITS_PRECOND_NONE; }
#endif
'''

# (file, line, where, anchor pattern the ORIGINAL line must match, text)   where: "before" | "after"
# Line numbers are those of the reference at 2024-12-18 (SURVEY.md header).
PATCHES = [
    # ---------------------------------------------------------------- solver.hpp
    ("solver.hpp", 9, "after", r'#include "methods/cg\.hpp"', '#include "its_binding.hpp"\n'),
    ("solver.hpp", 86, "before", r"^\s*Solver\(std::string _solver_type",
     "#ifdef USE_MI355X\n    its_handle *handle = nullptr;      // device-resident solver state (include/its_hip.h)\n"
     "    double r0_inf = 0.0, r0_two = 0.0;\n#endif\n"),
    # init_structs (solver.hpp:381-401): the host structs and timers are still initialised (postprocessing prints the
    # timers), then the device side is set up
    ("solver.hpp", 401, "before", r"^    \}\s*$",
     "#ifdef USE_MI355X\n"
     "        ITS_CALL(its_set_option(handle, \"gmres_givens\", 1));     // the reference's dense Givens formulation, gmres.hpp:427-658\n"
     "        ITS_CALL(its_solver_setup(handle, its_solver_kind(solver_type), GMRES_RESTART_LEN,\n"
     "                                  its_precond_kind(preconditioner_type), ITS_JACOBI_SHIPPED, ITS_GS_LEVEL));\n"
     "#endif\n"),
    # iterate (solver.hpp:403-550)
    ("solver.hpp", 410, "after", r"^\s*\)\{\s*$",
     "#ifdef USE_MI355X\n        double rn = 0.0;                                       // the ABI is fp64 (WORKING_PRECISION = double)\n"
     "        ITS_CALL(its_iterate(handle, iter_count, &rn));\n        *residual_norm = (VT)rn;\n#else\n"),
    ("solver.hpp", 550, "before", r"^    \}\s*$", "#endif\n"),
    # restart_gmres (solver.hpp:552-656)
    ("solver.hpp", 558, "after", r"^\s*\)\{\s*$", "#ifdef USE_MI355X\n        ITS_CALL(its_gmres_restart(handle, iter_count));\n#else\n"),
    ("solver.hpp", 656, "before", r"^    \}\s*$", "#endif\n"),
    # save_x_star (solver.hpp:658-726)
    ("solver.hpp", 662, "after", r"^\s*\)\{\s*$",
     "#ifdef USE_MI355X\n        if constexpr (std::is_same<VT, double>::value) ITS_CALL(its_download_x(handle, this->x_star, vec_size, iter_count));\n#else\n"),
    ("solver.hpp", 726, "before", r"^    \}\s*$", "#endif\n"),
    # exchange_arrays (solver.hpp:772-836): the swaps happen on the device inside its_iterate
    ("solver.hpp", 772, "after", r"void exchange_arrays\(int vec_size\)\{", "#ifndef USE_MI355X\n"),
    ("solver.hpp", 836, "before", r"^    \}\s*$", "#endif\n"),
    # init_residual (solver.hpp:838-920): the host still computes r (main.cpp:319 takes the stopping value from it, with
    # the reference's own norm routines); then matrix and vectors go to the device — everything main.cpp:206-276 has
    # prepared by now
    ("solver.hpp", 920, "before", r"^    \}\s*$",
     "#ifdef USE_MI355X\n"
     "        if constexpr (std::is_same<VT, double>::value) {\n"
     "            if (!handle) {\n"
     "                if (its_create(&handle, 0) != 0) { printf(\"ERROR: no HIP device / libits_hip failed to start.\\n\"); exit(1); }\n"
     "                g_its_handle = handle;\n"
     "            }\n"
     "            CRSMtxData<VT> *A = sparse_mat->crs_mat;\n"
     "            ITS_CALL(its_upload_crs(handle, ITS_MAT_A, A->n_rows, A->n_cols, A->nnz, A->row_ptr, A->col, A->val, ITS_FMT_CSR_EXACT));\n"
     "            if (solver_type == \"gauss-seidel\" || preconditioner_type == \"gauss-seidel\") {\n"
     "                CRSMtxData<VT> *L = sparse_mat->crs_L, *U = sparse_mat->crs_U;\n"
     "                ITS_CALL(its_upload_crs(handle, ITS_MAT_L, L->n_rows, L->n_cols, L->nnz, L->row_ptr, L->col, L->val, ITS_FMT_CSR_EXACT));\n"
     "                ITS_CALL(its_upload_crs(handle, ITS_MAT_U, U->n_rows, U->n_cols, U->nnz, U->row_ptr, U->col, U->val, ITS_FMT_CSR_EXACT));\n"
     "            }\n"
     "            ITS_CALL(its_upload_vectors(handle, this->D, this->b, this->x_old, vec_size));\n"
     "            ITS_CALL(its_init_residual(handle, &r0_inf, &r0_two));\n"
     "        } else { printf(\"ERROR: the MI355X path computes in fp64: set WORKING_PRECISION = double.\\n\"); exit(1); }\n"
     "#endif\n"),
    # ---------------------------------------------------------------- utility_funcs.hpp: record_residual_norm keeps its
    # bookkeeping half (:488-502) and takes the value its_iterate returned; the post-loop call (solver_harness.hpp:122-149
    # passes x_star for both x and x_new) asks the device for the reference's "final" value
    ("utility_funcs.hpp", 12, "after", r"#include <vector>", '#include "its_binding.hpp"\n'),
    ("utility_funcs.hpp", 407, "after", r"^\)\{\s*$",
     "#ifdef USE_MI355X\n"
     "    if (x == x_new) { double v = (double)*residual_norm; ITS_CALL(its_final_residual(g_its_handle, &v)); *residual_norm = (VT)v; }\n"
     "#else\n"),
    ("utility_funcs.hpp", 485, "before", r"^\s*$", "#endif\n"),
]


def patched_tree(dst):
    for name in os.listdir(REF):
        p = os.path.join(REF, name)
        if os.path.isfile(p) and (name.endswith((".hpp", ".h", ".cpp"))):
            shutil.copy(p, os.path.join(dst, name))
    shutil.copytree(os.path.join(REF, "methods"), os.path.join(dst, "methods"))
    open(os.path.join(dst, "its_binding.hpp"), "w").write(BINDING_HPP)
    by_file = {}
    for f, line, where, pat, text in PATCHES:
        by_file.setdefault(f, []).append((line, where, pat, text))
    for f, items in by_file.items():
        path = os.path.join(dst, f)
        lines = open(path).read().split("\n")
        for line, where, pat, text in items:
            got = lines[line - 1]
            if not re.search(pat, got):
                raise SystemExit(f"build_integrated: {f}:{line} is {got!r}, expected /{pat}/ — the reference changed, update the line numbers")
        for line, where, pat, text in sorted(items, key=lambda t: -t[0]):          # bottom up: earlier numbers stay valid
            at = line - 1 if where == "before" else line
            lines[at:at] = text.rstrip("\n").split("\n")
        open(path, "w").write("\n".join(lines))


def build(dst, out_dir, restarts, tol, opt="-O3"):
    lib_dir = os.path.join(ROOT, "iterative_solvers_amd")
    if not os.path.exists(os.path.join(lib_dir, "libits_hip.so")):
        raise SystemExit("build_integrated: build the library first (make -C iterative_solvers_amd/csrc)")
    outs = []
    for m in restarts:
        flags = ["-DUSE_MI355X", "-DVECTOR_LENGTH=8", "-DMAX_ITERS=5000", f"-DTOL={tol}", f"-DGMRES_RESTART_LEN={m}",
                 "-DWORKING_PRECISION=double", "-w", opt, "-fopenmp", "-mavx2", "-mfma", "-std=c++17", "-I" + dst,
                 "-I" + os.path.join(ROOT, "include")]
        exe = os.path.join(out_dir, f"its_integrated_m{m}")
        subprocess.check_call(["g++"] + flags + ["-c", os.path.join(dst, "mmio.cpp"), "-o", os.path.join(dst, "mmio.o")])
        subprocess.check_call(["g++"] + flags + ["-c", os.path.join(dst, "main.cpp"), "-o", os.path.join(dst, f"main_m{m}.o")])
        subprocess.check_call(["g++"] + flags + [os.path.join(dst, "mmio.o"), os.path.join(dst, f"main_m{m}.o"), "-o", exe,
                                                 "-L" + lib_dir, "-lits_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,$ORIGIN/../../iterative_solvers_amd"])
        outs.append(exe)
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-into", default=None, help="directory that keeps the binaries (oracle/_ref)")
    ap.add_argument("--restarts", default=None, help="comma-separated GMRES_RESTART_LEN values (default 100; 10,50,100 with --build-into)")
    ap.add_argument("--tol", default="1e-10")
    ap.add_argument("--opt", default="-O3", help="optimisation flag (-O0 for a quick does-it-compile-and-link check)")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        print("build_integrated: /root/reference is not here: nothing to check (container-only step)")
        return 0
    restarts = [int(v) for v in (a.restarts or ("10,50,100" if a.build_into else "100")).split(",")]
    with tempfile.TemporaryDirectory() as td:
        patched_tree(td)
        out_dir = a.build_into or td
        os.makedirs(out_dir, exist_ok=True)
        outs = build(td, out_dir, restarts, a.tol, a.opt)
        for exe in outs:
            # the binary must need libits_hip.so and must not have pulled in any CPU solver path for iterate
            needed = subprocess.check_output(["readelf", "-d", exe], text=True)
            assert "libits_hip.so" in needed, "the binding did not link against libits_hip.so"
            syms = subprocess.check_output(["nm", "-C", "--undefined-only", exe], text=True)
            for fn in ("its_create", "its_upload_crs", "its_upload_vectors", "its_init_residual", "its_solver_setup", "its_iterate",
                       "its_gmres_restart", "its_download_x", "its_final_residual"):
                assert re.search(rf"\bU {fn}\b", syms), f"{fn} is not called by the integrated binary"
            print(f"build_integrated: {os.path.relpath(exe, ROOT) if a.build_into else os.path.basename(exe)}: compiled from the reference's "
                  f"main.cpp + solver_harness.hpp with the USE_MI355X binding, linked against libits_hip.so "
                  f"({len(PATCHES)} insertions into solver.hpp / utility_funcs.hpp)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
