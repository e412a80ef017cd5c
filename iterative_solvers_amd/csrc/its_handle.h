// its_handle.h — the library's per-GPU state (opaque `its_handle` of include/its_hip.h), shared by
// its_api.hip (single-GPU state machine) and its_dist.hip (multi-GPU layer).
#pragma once
#include "../../include/its_hip.h"
#include "its_kernels.h"

#include <algorithm>
#include <string>
#include <vector>

using namespace its;

// ---------------------------------------------------------------------------------------
struct DistPlan;   // its_dist.hip (multi-GPU); opaque here

struct its_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_region_ms = 0.f;
    // in-loop timing of the dominant kernel (SpMV) for bench.py's roofline: event pairs around the
    // launch inside its_iterate_n
    std::vector<hipEvent_t> prof_ev;
    int prof_n = 0;
    bool prof_on = false;
    float prof_spmv_ms = 0.f;
    std::string err;

    // A, crs_L, crs_U as uploaded, and ITS_MAT_DU: the rows of A from the diagonal on (D + U), built by its_solver_setup
    // for Gauss-Seidel (option "gs_prefix", below) — never uploaded by a caller
    CsrDev mat[4];
    bool have[4] = {false, false, false, false};
    int fmt = ITS_FMT_CSR_EXACT;          // format used by the solver loops for A
    // ITS_FMT_MIXED uploads (its_ap.h): storage mode (ITS_AP_*: the reference's AP_VALUE_TYPE) and its two magnitude
    // thresholds (AP_THRESHOLD_1, AP_THRESHOLD_2 of config.mk:59-62); its_set_mixed_threshold = ITS_AP_DP_SP with threshold 1
    int mixed_mode = 0;
    double mixed_threshold = 1.0, mixed_threshold2 = 0.0;

    int n = 0;                            // owned rows
    int n_ext = 0;                        // owned + halo (== n on one GPU)
    // Solver<VT> vectors (solver.hpp:23-34).  x_a/x_b ping-pong as x_old/x_new.
    double *D = nullptr, *b = nullptr, *r = nullptr, *tmp = nullptr;
    double *x_a = nullptr, *x_b = nullptr, *x = nullptr;   // x: GS / GMRES iterate
    // cgArgs (cg.hpp:16-38): p and q=A p; r is shared
    double *p = nullptr;
    // gmresArgs (gmres.hpp:16-42)
    double *V = nullptr, *hcol_dev = nullptr, *y_dev = nullptr;
    // GMRES: the next trip's w = A v_{k+1} (+ preconditioner) queued behind this trip's Gram-Schmidt sweep, so that it runs
    // while the host waits for the Hessenberg column and turns it (option "gmres_ahead", default 1).  gm_w2 alternates
    // with tmp; gm_ahead_iter = the trip the product is for, -1 = none
    double *gm_w2 = nullptr, *gm_ahead_w = nullptr;
    int gm_ahead_iter = -1, gm_ahead_restart = -1, gmres_ahead = 1;
    int m = 0, restart_count = 0;
    double beta = 0.0;
    std::vector<double> Rm, g, cs, sn, hcol;   // host: R (col-major (m+1) x m), g, Givens c/s
    // dense least-squares state of the reference formulation (gmres_givens == 1)
    std::vector<double> Hd, Qd, Jd, Qc, Ht, Rd;
    int gmres_givens = 0;                 // 0: O(m) rotation updates, 1: the reference's dense products
    bool dense_full = false;              // dense formulation: the rest of this cycle runs the full products (a non-finite entry was seen)
    // communication-reducing variants (its_krylov_cr.hip), opt-in: "cg_variant" 1 = pipelined CG, "gmres_ortho" 1 = CGS-2
    int cg_variant = 0, gmres_ortho = 0;
    // persistent modified Gram-Schmidt (its_mgs.hip): option "gmres_persist" (1), for vectors of at least
    // "gmres_persist_min_n" doubles (default 1: every length) that fit the registers of one resident grid;
    // mgs_grid == 0: per-step kernels
    int gmres_persist = 1, gmres_persist_min_n = 1;
    unsigned long long *mgs_slots = nullptr;
    int mgs_grid = 0, mgs_fallbacks = 0, mgs_parity = 0;
    int mgs_inject_kind = 1;              // option "gmres_persist_inject_kind": who gives up (its_mgs.hip)
    int mgs_inject_fail = 0;              // option "gmres_persist_inject_fail" (tests): launches that give up on purpose
    double *pcg_w = nullptr, *pcg_s = nullptr, *pcg_z = nullptr;     // pipelined CG: w = A r, s = A p, z = A s
    double *pcg_pair[2] = {nullptr, nullptr};                        // (r,r) and (w,r) partials, double-buffered by trip parity
    int pcg_trip = 0;
    double *gram_part = nullptr, *gram_h = nullptr;                  // CGS-2: batched-dot partials, h of both passes
    // reductions / scalars
    double *part[4] = {nullptr, nullptr, nullptr, nullptr};
    double *sc = nullptr;                 // SC_COUNT device scalars
    double *res_dev = nullptr;            // per-trip residuals for its_iterate_n
    int res_cap = 0;
    // pinned host buffer.  Layout (doubles): [0, PIN_RES) scalar read-backs, PIN_RES the per-trip residual word,
    // then three regions of m + 2 each, sized from the GMRES restart length in its_solver_setup:
    // pin_copy (H column, copy path), pin_y (y staging of gmres_get_x), pin_h (H column written by the kernels)
    double *pinned = nullptr;
    size_t pinned_cap = 0;
    size_t pin_copy = 0, pin_y = 0, pin_h = 0;
    int rr_slot = 0;

    int solver = -1, precond = 0, jacobi_mode = 0, gs_order = 0;
    // Jacobi / Gauss-Seidel trips read the matrix once: the pass that records the residual of the new iterate
    // also produces the next trip's product of it (EPI_JACOBI_FUSED / EPI_GS_FUSED).  `ahead` = that product
    // (Jacobi: the next iterate in x_b; GS: b - U x in tmp) is already there.  Option "fused_residual" = 0
    // keeps the two separate launches (same bits).
    int fused_residual = 1;
    int dist_overlap = 0;                        // option "dist_overlap": halo exchange beside the interior SpMV (its_dist.hip)
    bool ahead = false;
    // GS, fused pass on an ELL kernel, first sweep level without entries in L (first colour): that pass also stores the
    // level's next iterate (b - U x) / D into x_b, the sweep continues there from level 1 and x / x_b swap (option
    // "gs_first_level_fused", same bits).  ahead_pre = x_b holds it (only meaningful while `ahead`).
    bool ahead_pre = false;
    int gs_first_level_fused = 1;
    // Gauss-Seidel, option "gs_prefix" (1): with ascending columns a row of A is [its entries of L | diagonal | its entries
    // of U], so the in-order chain of the recorded residual b - A x_new begins with exactly the chain the sweep of that
    // row has just run over L.  The sweep leaves its chain sums in tmp (in place of the right-hand side it has consumed),
    // and the pass that records the residual and produces the next b - U x walks D + U only (mat[ITS_MAT_DU]), its full
    // chain starting from those sums: the same operations in the same order — the same bits — with the entries of L
    // read once per trip instead of twice.  gs_prefix_ready: its_solver_setup has verified on the device that crs_L,
    // the diagonal and crs_U are A's rows entry for entry (bitwise, in order) and built the image.
    int gs_prefix = 1;
    bool gs_prefix_ready = false;
    bool setup_done = false, residual_done = false;
    double last_residual = 0.0;
    int watch_window_ms = 50;                    // option "watch_window_ms": how long its_iterate watches before it synchronizes
    int watch_misses = 0;                        // wait_pinned_word: times the watched word did not arrive within its window
    // triangular sweep schedule (levels of L; colours when the matrix was colour-permuted)
    std::vector<int> level_ptr;           // host offsets into level_rows
    std::vector<int> level_nshort;            // rows of each level with <= SWEEP_SHORT entries in L (listed first)
    // levels whose rows are one contiguous range and large (the colours of a multicolour ordering) are swept by
    // the exact stream kernels (EPI_SWEEP) over their own row-block descriptors of L
    int4 *sweep_desc = nullptr;                  // device, all such levels back to back
    std::vector<int> level_desc_first, level_desc_count;   // per level; count 0 = level-list kernel
    std::vector<SweepEll> level_ell;             // per level: its own ELL image (val == nullptr: none), option "sweep_ell_levels"
    int sweep_ell_levels = 1;
    std::vector<int> level_empty_lo;             // per level: first row of a contiguous level whose rows have no entries in L (-1: not such a level)
    int sweep_stream = 1;                        // option "sweep_stream": 0 = level-list kernel for every level
    // host image of crs_L's row structure, kept at upload so that the schedule needs no copy back from the device:
    // row_ptr and, per row, the largest owned column (-1: none)
    std::vector<int> L_rp, L_maxcol;
    bool L_has_halo_cols = false;                // crs_L has columns >= its row count (legal only in a row-partitioned run)
    // level-ordered copy of L for scattered levels of short rows (natural-order sweeps of stencils), its_kernels.hip
    int *sw_col = nullptr; double *sw_val = nullptr, *sw_D = nullptr;
    int sw_w = 0;                         // 0: not built
    int sweep_ell = 1;                    // option "sweep_ell"
    // runs of consecutive small levels of that copy go out as one launch of one workgroup (option "sweep_run")
    int *level_ptr_dev = nullptr;         // device copy of level_ptr (only with the level-ordered copy)
    std::vector<int> level_run_end;       // per level: one past the last level of the run that starts here (0: no run starts here)
    int sweep_run = 1;
    int *level_rows = nullptr;            // device
    int n_levels = 0;
    // natural-order sweep of a stencil matrix as one launch pipelined over grid lines (its_lines.hip; option "sweep_lines"):
    // built by its_solver_setup when crs_L's offsets are {-1, -a[, -b]} and every row passes the device check
    LinesPlan lines;
    bool lines_ready = false, lines_failed = false;
    int sweep_lines = 1;
    // multicolour permutation (gs_order == ITS_GS_MULTICOLOR): new index -> old index
    std::vector<int> mc_perm, mc_iperm;

    DistPlan *dist = nullptr;
};

constexpr int ITS_MAT_DU = 3;             // its_handle::mat slot of the D + U image (internal)
constexpr size_t PIN_LINES = 9;           // "the line-pipelined sweep gave up" word inside its_handle::pinned
constexpr size_t PIN_RES = 8;             // index of the residual word inside its_handle::pinned
constexpr size_t PIN_REGIONS = 16;        // first index of the restart-length-sized regions

#define HIPCHK(h, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return ITS_EHIP;                                                                \
        }                                                                                   \
    } while (0)
#define FAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

template <class T> static inline int dalloc(its_handle *h, T **p, size_t count)
{
    void *q = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T) + 256;
    HIPCHK(h, hipMalloc(&q, bytes));
    // ITS_DEBUG_POISON_MALLOC=1 (test aid): every allocation starts as all-ones (NaN as a double, -1 as an index) instead of
    // whatever the allocator hands out — fresh memory is zero in a new process and something else later
    static const bool poison = getenv("ITS_DEBUG_POISON_MALLOC") && atoi(getenv("ITS_DEBUG_POISON_MALLOC")) != 0;
    if (poison) { HIPCHK(h, hipMemset(q, 0xFF, bytes)); HIPCHK(h, hipDeviceSynchronize()); }     // (the fill must not overtake what the caller queues next)
    *p = (T *)q;
    return ITS_OK;
}
template <class T> static inline void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

