// structs.hpp — Flags / LoopParams / argType with the reference's field names (structs.hpp:17-119).
#pragma once
#include <string>
#include <vector>
#include "stopwatch.hpp"

struct Flags {
    bool print_iters = false;
    bool print_summary = true;
    bool print_residuals = true;
    bool convergence_flag = false;
    bool random_data = false;
};

struct LoopParams {
    int iter_count = 0;
    int residual_count = 0;
    int residual_check_len = 1;
    int max_iters = MAX_ITERS;
    double stopping_criteria = 0.0;
    double tol = TOL;
    double init_b = 1.0;
    double init_x = 1.0;
    int gmres_restart_len = GMRES_RESTART_LEN;

    // tol * ||r0||_2 for GMRES, tol * ||r0||_inf otherwise (reference structs.hpp:41-68)
    double compute_stopping_criteria(const std::string &solver_type, double r0_inf, double r0_two) const
    {
        return tol * (solver_type == "gmres" ? r0_two : r0_inf);
    }
};

struct COOMtxData {
    long n_rows = 0, n_cols = 0, nnz = 0;
    std::vector<int> I, J;
    std::vector<double> values;
};

struct CRSMtxData {
    int n_rows = 0, n_cols = 0, nnz = 0;
    std::vector<int> row_ptr, col;
    std::vector<double> val;
};

struct argType {
    COOMtxData coo_mat;
    CRSMtxData crs_mat, crs_L, crs_U;
    Timers timers;
    std::vector<double> normed_residuals;
    LoopParams loop_params;
    Flags flags;
    std::string solver_type, preconditioner_type, scale_type = "diag", matrix_file_name;
    int vec_size = 0;
};
