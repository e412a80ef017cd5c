// io_funcs.hpp — CLI, Matrix Market input and the summary text, byte-compatible with the
// reference (io_funcs.hpp:22-123, 125-289, 291-332, 361-458) so its output can be diffed.
#pragma once
#include <algorithm>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <numeric>
#include "structs.hpp"
#include "mtx_fast.hpp"

inline void assign_cli_inputs(argType *args, int argc, char *argv[])
{
    if (argc < 3 || argc > 5) {
        printf("ERROR: assign_cli_inputs: Please only select a .mtx file name and solver type:\n-j (Jacobi)\n-gs (Gauss-Seidel)\n-gm ([Preconditioned] GMRES)\n-cg ([Preconditioned] Conjugate Gradient)\n");
        exit(1);
    }
    args->matrix_file_name = argv[1];
    const std::string st = argv[2];
    if (st == "-j") args->solver_type = "jacobi";
    else if (st == "-gs") args->solver_type = "gauss-seidel";
    else if (st == "-cg") args->solver_type = "conjugate-gradient";
    else if (st == "-gm") args->solver_type = "gmres";
    else {
        printf("ERROR: assign_cli_inputs: Please choose an available solver type:\n-j (Jacobi)\n-gs (Gauss-Seidel)\n-[p]gm ([Preconditioned] GMRES)\n-[p]cg ([Preconditioned] Conjugate Gradient)\n");
        exit(1);
    }
    for (int i = 3; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg == "-p" && i + 1 < argc) {
            const std::string pt = argv[++i];
            if (pt == "j") args->preconditioner_type = "jacobi";
            else if (pt == "gs") args->preconditioner_type = "gauss-seidel";
            else { printf("ERROR: assign_cli_inputs: Please choose an available preconditioner type:\n-j (Jacobi)\n-gs (Gauss-Seidel)\n"); exit(1); }
        } else if (arg == "-scale" && i + 1 < argc) {
            const std::string sc = argv[++i];
            if (sc == "max" || sc == "diag" || sc == "none") args->scale_type = sc;
            else { printf("ERROR: assign_cli_inputs: Please choose an available matrix scaling type:\nmax (Max row/col element)\ndiag (Diagonal)\nnone\n"); exit(1); }
        } else {
            std::cout << "ERROR: assign_cli_inputs: Arguement \"" << arg << "\" not recongnized." << std::endl;
        }
    }
}

// coordinate real|integer|pattern, general|symmetric; 1-based -> 0-based; symmetric entries are
// mirrored right behind their original; stable sort by row keeps the file order inside a row
inline void read_mtx(const std::string &file, COOMtxData *coo)
{
    // parallel parser / binary cache first (mtx_fast.hpp: the same COO, faster); ITS_MTX_READER=fscanf
    // forces the loop below, which is also where every malformed file ends up and gets its error message
    const char *rd = getenv("ITS_MTX_READER");
    if (!(rd && !strcmp(rd, "fscanf")) && mtx_fast::read(file, coo)) return;
    FILE *f = fopen(file.c_str(), "r");
    if (!f) { printf("Unable to open file"); exit(1); }
    char line[1100], banner[64], obj[64], fmt[64], field[64], symm[64];
    if (!fgets(line, sizeof line, f) || sscanf(line, "%63s %63s %63s %63s %63s", banner, obj, fmt, field, symm) != 5 ||
        strcmp(banner, "%%MatrixMarket") != 0) {
        printf("mm_read_unsymetric: Could not process Matrix Market banner  in file [%s]\n", file.c_str());
        exit(0);
    }
    auto lower = [](char *s) { for (; *s; ++s) *s = (char)tolower(*s); };
    lower(obj); lower(fmt); lower(field); lower(symm);
    const bool pattern = !strcmp(field, "pattern"), symmetric = !strcmp(symm, "symmetric");
    if (strcmp(obj, "matrix") || strcmp(fmt, "coordinate") || !(pattern || !strcmp(field, "real") || !strcmp(field, "integer")) ||
        !(symmetric || !strcmp(symm, "general"))) {
        printf("The matrix market file provided is not supported.\n");
        exit(0);
    }
    int M = 0, N = 0, nz = 0;
    do { if (!fgets(line, sizeof line, f)) { printf("Error in file reading\n"); exit(1); } } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", &M, &N, &nz) != 3) { printf("Error in file reading\n"); exit(1); }
    if (M != N) { printf("Matrix not square. Currently only square matrices are supported\n"); exit(1); }
    std::vector<int> ri, ci; std::vector<double> vv;
    for (int k = 0; k < nz; ++k) {
        int r, c; double v = 0.01;
        const int got = pattern ? fscanf(f, "%d %d\n", &r, &c) : fscanf(f, "%d %d %lg\n", &r, &c, &v);
        if (got != (pattern ? 2 : 3)) { printf("Error in file reading\n"); exit(1); }
        --r; --c;
        ri.push_back(r); ci.push_back(c); vv.push_back(v);
        if (symmetric && r != c) { ri.push_back(c); ci.push_back(r); vv.push_back(v); }
    }
    fclose(f);
    std::vector<int> perm(ri.size());
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return ri[a] < ri[b]; });
    coo->n_rows = M; coo->n_cols = N; coo->nnz = (long)ri.size();
    coo->I.resize(ri.size()); coo->J.resize(ri.size()); coo->values.resize(ri.size());
    for (size_t k = 0; k < ri.size(); ++k) { coo->I[k] = ri[perm[k]]; coo->J[k] = ci[perm[k]]; coo->values[k] = vv[perm[k]]; }
}

inline void summary_output(argType *args)
{
    const LoopParams &lp = args->loop_params;
    if (args->flags.print_residuals)
        for (int i = 0; i < lp.residual_count; ++i)
            std::cout << "||A*x_" << i * lp.residual_check_len << " - b||_infty = " << std::setprecision(16)
                      << args->normed_residuals[i] << std::endl;
    if (args->flags.convergence_flag)
        std::cout << "\n" << args->solver_type << " solver converged in: " << lp.iter_count << " iterations." << std::endl;
    else
        std::cout << "\n" << args->solver_type << " solver did not converge after " << lp.max_iters << " iterations." << std::endl;
    std::cout << "With the stopping criteria \"tol * || b-A*x_0 ||_infty\" is: " << lp.stopping_criteria << std::endl;
    std::cout << "The residual of the final iteration is: ||A*x_star - b||_infty = " << std::scientific
              << args->normed_residuals[lp.residual_count] << ".\n";
}

inline void print_timers(argType *args)
{
    const double total = args->timers.total_wtime.get_wtime(), pre = args->timers.preprocessing_wtime.get_wtime();
    const double harness = args->timers.solver_harness_wtime.get_wtime(), solver = args->timers.solver_wtime.get_wtime();
    const int R = 30, Lw = 25;
    const char *name = args->solver_type == "jacobi" ? "| | Jacobi Solver time: " : args->solver_type == "gauss-seidel" ? "| | GS Solver time: "
                     : args->solver_type == "gmres" ? "| | GMRES Solver time: " : "| | CG Solver time: ";
    std::cout << "+---------------------------------------------------------+" << std::endl;
    std::cout << std::left << std::setw(Lw) << "Total elapsed time: " << std::right << std::setw(R) << total << "[s]" << std::endl;
    std::cout << std::left << std::setw(Lw) << "| Preprocessing time: " << std::right << std::setw(R) << pre << "[s]" << std::endl;
    std::cout << std::left << std::setw(Lw) << "| Solver Harness time: " << std::right << std::setw(R) << harness << "[s]" << std::endl;
    std::cout << std::left << std::setw(Lw) << name << std::right << std::setw(R) << solver << "[s]" << std::endl;
    std::cout << std::left << std::setw(Lw) << "| | Other: " << std::right << std::setw(R) << harness - solver << "[s]" << std::endl;
    std::cout << "+---------------------------------------------------------+" << std::endl;
}

inline void postprocessing(argType *args)
{
    args->timers.total_wtime.end_stopwatch();
    if (args->flags.print_summary) summary_output(args);
    print_timers(args);
}
