// solver.hpp — Solver<VT> with the reference's method names (solver.hpp:151,381,403,552,658,772,
// 838), every one a thin call into the C ABI of libits_hip.so (include/its_hip.h).  All vectors
// live in HBM; the host object only keeps x_star.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "its_hip.h"
#include "structs.hpp"
#include "utility_funcs.hpp"

template <typename VT>
class Solver {
    std::string solver_type, preconditioner_type;
    its_handle *h = nullptr;

    void check(int rc, const char *what) const
    {
        if (rc != ITS_OK) { printf("ERROR: %s: %s (%d)\n", what, its_last_error(h), rc); exit(1); }   // reference convention
    }
    int kind() const
    {
        return solver_type == "jacobi" ? ITS_JACOBI : solver_type == "gauss-seidel" ? ITS_GAUSS_SEIDEL
             : solver_type == "gmres" ? ITS_GMRES : ITS_CG;
    }

public:
    std::vector<VT> x_star, D, b, x_old;
    int gmres_restart_length = GMRES_RESTART_LEN;

    Solver(std::string st, std::string pt) : solver_type(std::move(st)), preconditioner_type(std::move(pt))
    {
        static_assert(sizeof(VT) == sizeof(double), "the gfx950 kernels compute in fp64");
        if (its_create(&h, DEVICE) != ITS_OK) { printf("ERROR: Solver: no HIP device / libits_hip failed to start.\n"); exit(1); }
    }
    ~Solver() { if (h) its_destroy(h); }

    // allocate_cpu_solver_structs + the CRS hand-over (reference main.cpp:206-229; GS: gauss_seidel.hpp:249-630)
    void allocate_cpu_solver_structs(argType *args)
    {
        std::cout << "Allocating space for Solver structs" << std::endl;
        const CRSMtxData &A = args->crs_mat;
        check(its_upload_crs(h, ITS_MAT_A, A.n_rows, A.n_cols, A.nnz, A.row_ptr.data(), A.col.data(), A.val.data(), SPMV_FORMAT), "its_upload_crs");
        if (solver_type == "gauss-seidel" || preconditioner_type == "gauss-seidel") {
            COOMtxData L, U;
            split_L_U(args->coo_mat, L, U);
            convert_to_crs(L, args->crs_L); convert_to_crs(U, args->crs_U);
            check(its_upload_crs(h, ITS_MAT_L, A.n_rows, A.n_cols, args->crs_L.nnz, args->crs_L.row_ptr.data(), args->crs_L.col.data(), args->crs_L.val.data(), ITS_FMT_CSR_EXACT), "its_upload_crs(L)");
            check(its_upload_crs(h, ITS_MAT_U, A.n_rows, A.n_cols, args->crs_U.nnz, args->crs_U.row_ptr.data(), args->crs_U.col.data(), args->crs_U.val.data(), ITS_FMT_CSR_EXACT), "its_upload_crs(U)");
        }
        x_star.assign(A.n_cols, 0.0);
    }
    void upload_vectors() { check(its_upload_vectors(h, D.data(), b.data(), x_old.data(), (int)b.size()), "its_upload_vectors"); }

    void init_residual(double *r0_inf, double *r0_two) { check(its_init_residual(h, r0_inf, r0_two), "init_residual"); }
    void init_structs()
    {
        const int pre = preconditioner_type == "gauss-seidel" ? ITS_PRECOND_GS : preconditioner_type == "jacobi" ? ITS_PRECOND_JACOBI : ITS_PRECOND_NONE;
        check(its_set_option(h, "gmres_givens", GMRES_GIVENS), "its_set_option");
        check(its_solver_setup(h, kind(), gmres_restart_length, pre, JACOBI_MODE, GS_ORDER), "init_structs");
    }
    // iterate + the value record_residual_norm computes + exchange_arrays (solver_harness.hpp:43-90)
    void iterate(int iter_count, VT *residual_norm) { check(its_iterate(h, iter_count, residual_norm), "iterate"); }
    void exchange_arrays(int) {}                        // pointer swaps happen on the device inside iterate
    void restart_gmres(int iter_count) { check(its_gmres_restart(h, iter_count), "restart_gmres"); }
    void save_x_star(int iter_count) { check(its_download_x(h, x_star.data(), (int)x_star.size(), iter_count), "save_x_star"); }
    void final_residual(VT *r) { check(its_final_residual(h, r), "record_residual_norm"); }
};
