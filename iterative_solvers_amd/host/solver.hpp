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
    static constexpr int GS_ORDER_AUTO = 2, GS_AUTO_LEVELS = 600, GS_AUTO_ROWS = 1000000;
    std::vector<VT> x_star, D, b, x_old;
    int gmres_restart_length = GMRES_RESTART_LEN;
    int gs_order = GS_ORDER;                   // config.mk GS_ORDER (run-time override: ITS_GS_ORDER)
    std::vector<int> perm;                     // multicolour order, new -> old (empty = natural order)

    Solver(std::string st, std::string pt) : solver_type(std::move(st)), preconditioner_type(std::move(pt))
    {
        static_assert(sizeof(VT) == sizeof(double), "the gfx950 kernels compute in fp64");
        if (its_create(&h, DEVICE) != ITS_OK) { printf("ERROR: Solver: no HIP device / libits_hip failed to start.\n"); exit(1); }
    }
    ~Solver() { if (h) its_destroy(h); }

    // GS_ORDER = multicolor: the reference's natural-order sweep (gauss_seidel.hpp:41-247) runs on P A P^T with
    // rows ordered colour by colour (its_hip.h, "Gauss-Seidel ordering helpers"); b and x0 are permuted alike
    // and x_star comes back in the original numbering.
    void reorder_multicolor(argType *args)
    {
        if (solver_type == "gauss-seidel" && gs_order == GS_ORDER_AUTO) {
            // GS_ORDER = auto: the reference's natural order unless it has more than GS_AUTO_LEVELS dependency levels on a
            // matrix of GS_AUTO_ROWS rows or more (256^3: 1.5 ms per trip with the line-pipelined sweep, 3.6 ms with a launch per
            // level, against 0.45 multicolour)
            const CRSMtxData &A0 = args->crs_mat;
            const int levels = A0.n_rows >= GS_AUTO_ROWS ? natural_order_levels(A0) : 0;
            gs_order = levels > GS_AUTO_LEVELS ? ITS_GS_MULTICOLOR : ITS_GS_LEVEL;
            if (gs_order == ITS_GS_MULTICOLOR)
                fprintf(stderr, "its_solve: GS_ORDER = auto: the natural order of this matrix has %d dependency levels: sweeping colour by colour "
                                "(a different iteration than the reference's; GS_ORDER = level keeps its order)\n", levels);
        }
        if (solver_type != "gauss-seidel" || gs_order != ITS_GS_MULTICOLOR) return;
        CRSMtxData &A = args->crs_mat;
        int n_colors = 0;
        perm.assign(A.n_rows, 0);
        check(its_multicolor_order(A.n_rows, A.row_ptr.data(), A.col.data(), perm.data(), &n_colors), "its_multicolor_order");
        CRSMtxData B = A;
        check(its_permute_crs(A.n_rows, A.row_ptr.data(), A.col.data(), A.val.data(), perm.data(),
                              B.row_ptr.data(), B.col.data(), B.val.data()), "its_permute_crs");
        A = B;
        COOMtxData &C = args->coo_mat;          // keep the COO image (split_L_U, extract_diag) in step
        for (int i = 0; i < A.n_rows; ++i)
            for (int j = A.row_ptr[i]; j < A.row_ptr[i + 1]; ++j) { C.I[j] = i; C.J[j] = A.col[j]; C.values[j] = A.val[j]; }
        std::cout << "Multicolour Gauss-Seidel ordering: " << n_colors << " colours" << std::endl;
    }
    // dependency levels of the natural-order forward substitution: level(i) = 1 + max level(j) over a_ij != 0, j < i
    static int natural_order_levels(const CRSMtxData &A)
    {
        std::vector<int> lev((size_t)A.n_rows, 0);
        int nl = 0;
        for (int i = 0; i < A.n_rows; ++i) {
            int l = 0;
            for (int j = A.row_ptr[i]; j < A.row_ptr[i + 1]; ++j) if (A.col[j] < i) l = std::max(l, lev[A.col[j]] + 1);
            lev[i] = l; nl = std::max(nl, l + 1);
        }
        return nl;
    }
    // after init_structs: a one-line hint (stderr) when the reference's order costs a launch per level on a large matrix
    void hint_sweep_order() const
    {
        if (solver_type != "gauss-seidel" || gs_order != ITS_GS_LEVEL) return;
        int lines_wg = 0, levels = 0;
        if (its_sweep_info(h, &lines_wg, nullptr, nullptr, &levels) != ITS_OK) return;
        if (lines_wg == 0 && levels > GS_AUTO_LEVELS && (int)D.size() >= GS_AUTO_ROWS)
            fprintf(stderr, "its_solve: hint: the natural-order sweep of this matrix is %d launches per trip (one per dependency level); "
                            "ITS_GS_ORDER=multicolor (config.mk GS_ORDER = multicolor, or auto) sweeps colour by colour, several times faster "
                            "per trip on large 3-D stencils — a different iteration than the reference's\n", levels);
    }
    void permute_vector(std::vector<VT> &v) const
    {
        if (perm.empty()) return;
        std::vector<VT> o(v.size());
        for (size_t i = 0; i < v.size(); ++i) o[i] = v[perm[i]];
        v.swap(o);
    }

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
        check(its_solver_setup(h, kind(), gmres_restart_length, pre, JACOBI_MODE, gs_order == GS_ORDER_AUTO ? (int)ITS_GS_LEVEL : gs_order), "init_structs");
    }
    // iterate + the value record_residual_norm computes + exchange_arrays (solver_harness.hpp:43-90)
    void iterate(int iter_count, VT *residual_norm) { check(its_iterate(h, iter_count, residual_norm), "iterate"); }
    void exchange_arrays(int) {}                        // pointer swaps happen on the device inside iterate
    void restart_gmres(int iter_count) { check(its_gmres_restart(h, iter_count), "restart_gmres"); }
    void save_x_star(int iter_count)
    {
        check(its_download_x(h, x_star.data(), (int)x_star.size(), iter_count), "save_x_star");
        if (!perm.empty()) {
            std::vector<VT> o(x_star.size());
            for (size_t i = 0; i < x_star.size(); ++i) o[perm[i]] = x_star[i];
            x_star.swap(o);
        }
    }
    void final_residual(VT *r) { check(its_final_residual(h, r), "record_residual_norm"); }
};
