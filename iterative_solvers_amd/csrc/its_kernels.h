// its_kernels.h — launch interface between the C ABI (its_api.hip) and the gfx950 kernels.
// Every launcher enqueues on the given stream and returns immediately (no sync, no malloc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "its_device.h"

namespace its {

// ---------------------------------------------------------------------------------------
// CSR "stream" plan: rows are cut into row blocks of at most ITS_BLOCK rows and at most
// SPMV_NNZ_MAX nonzeros (a single longer row forms a block of its own and is walked in
// several LDS chunks).  blk_desc[nb] = {first row, rows | longest row << 16, first nnz, end nnz}.
// ---------------------------------------------------------------------------------------
constexpr int SPMV_NNZ_MAX = 1536;        // (1536+8)*(8+4) B + row ptrs ≈ 19.6 KB LDS per buffer
// second plan for the two-workgroups-per-CU kernel of the CG form (spmv_stream4_kernel): the staging registers
// hold 2048 entries per trip, and with 256 full rows per block (7-pt: 1792 entries) it runs 3 % faster; the
// four-per-CU kernel needs the smaller LDS footprint of SPMV_NNZ_MAX
constexpr int SPMV_NNZ_BIG = 2040;
constexpr int SPMV_SOLO_LEN = 256;        // rows longer than this leave the irregular-row plan for the wave-per-row launch
// column-panel blocked image of large irregular matrices (its_panel.hip)
constexpr int PAN_TR = 4096;              // rows per tile: their running sums live in LDS (32 KB)
constexpr int PAN_CH = 256;               // entries per chunk (one per lane); a chunk never spans two panels
constexpr int PAN_LCUT = 128;             // rows longer than this stay on the irregular-row kernel
constexpr int PAN_COLBITS = 19;           // key = row in tile << 19 | column in panel
constexpr uint32_t PAN_COLMASK = (1u << PAN_COLBITS) - 1u;
constexpr int IRR_ROW_LEN = 32;           // longest row above which the exact SpMV splits gather and arithmetic
// col/val are allocated with this many zero elements after nnz: the stream kernel's prefetch
// loads are unconditional (2 slices x 1024 entries past an aligned block start)
constexpr int CSR_PAD = 2048 + 16;

// sliced unsorted ELL image of matrices with short regular rows (its_ell.hip): slices of ELL_S rows, a workgroup
// trip covers ELL_RB rows
constexpr int ELL_S = 128;
constexpr int ELL_RB = 4 * ELL_S;
constexpr int ELL_MAX_W = 8;              // longest row an ELL image is built for (8 nibbles in the index word)
constexpr int ELL_MAX_OFF = 15;           // dictionary entries a nibble can name (0xF = no entry)
constexpr int ELLW_MAX_W = 32;            // longest row of the byte-index form
constexpr int ELLW_MAX_OFF = 255;         // its dictionary (0xFF = no entry)

enum SpmvEpilogue {
    EPI_STORE = 0,        // y = A x                                  (kernels.hpp:380-409)
    EPI_RESIDUAL = 1,     // y = b - A x, + sum r^2 and max|r|        (kernels.hpp:670-726)
    EPI_JACOBI_SHIPPED = 2,  // out = (b - (x - D*(A x)))/D           (jacobi.hpp:191,41-62)
    EPI_JACOBI_REF = 3,   // out = (b - sum_{j!=i} a_ij x_j)/a_ii     (jacobi.hpp:65-93)
    EPI_GS_RHS = 4,       // y = b - U x                              (gauss_seidel.hpp:144,177)
    // one pass over A, two in-order chains per row: the recorded residual norm of x (full chain, as
    // EPI_RESIDUAL) AND the next trip's product of the same x (restricted chain).  Bit-identical to the two
    // separate launches they replace; the trips below use them to read the matrix once per trip.
    EPI_JACOBI_FUSED = 5, // out = EPI_JACOBI_REF of x, + sum (b - A x)^2 / max
    EPI_GS_FUSED = 6,     // y = b - (strictly upper part of A) x, + sum (b - A x)^2 / max
    // one dependency level of the triangular sweep whose rows are a contiguous range (a colour of the
    // multicolour ordering): x_i = (b_i - sum_j L_ij x_j) / D_i, y aliases x   (kernels.hpp:460-472)
    EPI_SWEEP = 7
};

struct CsrDev {
    int n_rows = 0, n_cols = 0;
    int64_t nnz = 0;
    int *row_ptr = nullptr, *col = nullptr;   // col/val padded by CSR_PAD elements
    double *val = nullptr;
    // column-offset dictionary (banded / stencil-class matrices): when the whole matrix has at most 256
    // distinct values of col - row, cidx[k] is the 1-byte dictionary index of entry k (padded like col)
    // and the exact SpMV streams it instead of the 4-byte column: 9 instead of 12 bytes per nonzero
    unsigned char *cidx = nullptr;
    int *off_dict = nullptr;                  // [256]
    int n_off = 0;                            // 0 = not eligible
    // value dictionary (opt-in, option "spmv_valdict"): at most 256 distinct fp64 bit patterns in val
    unsigned char *vidx = nullptr;
    double *val_dict = nullptr;               // [256]
    int n_vals = 0;                           // 0 = not eligible
    // sliced ELL image (its_ell.hip), built at upload when the matrix is eligible: longest row <= ELL_MAX_W,
    // n_off <= ELL_MAX_OFF, padding below 25 %.  ell_val[(slice*ell_w + k)*ELL_S + row_in_slice], ell_idx[row] =
    // ell_w nibbles (dictionary index of entry k, 0xF = none); ell_pb = row blocks per period of the far band
    // (0 = no period map)
    double *ell_val = nullptr;
    uint32_t *ell_idx = nullptr;
    int ell_w = 0, ell_blocks = 0, ell_pb = 0;
    int ell_kind = 0;                         // 1: nibble index word per row (rows <= 8, <= 15 offsets); 2: byte indices, four to a
                                              // word, slot-major like the values (rows <= 32, <= 255 offsets);
                                              // 3: explicit columns (rows <= 8, any columns; its_ellg.hip): base per (slice, slot)
                                              // + 8/16/32-bit deltas, width per slice
    uint32_t *ellg_desc = nullptr;            // [slices] first index word of the slice / 128 << 2 | width code (0: 8, 1: 16, 2: 32 bits)
    int *ellg_base = nullptr;                 // [slices * ell_w] smallest column of the slot in the slice
    int64_t ellg_idx_words = 0;               // index words of the image
    int ellg_hist[3] = {0, 0, 0};             // slices per width code
    // mixed-precision image (ITS_FMT_MIXED): the entries at or above the threshold as an fp64 CRS, the rest as an
    // fp32 CRS, both in the row order of the original
    int *mix_rp_d = nullptr, *mix_col_d = nullptr, *mix_rp_s = nullptr, *mix_col_s = nullptr;
    double *mix_val_d = nullptr;
    float *mix_val_s = nullptr;
    int64_t mix_nnz_d = 0, mix_nnz_s = 0;
    // third tier (its_ap.h: the "hp" modes of the reference's AP_VALUE_TYPE): binary16 bits of the float-rounded value
    int *mix_rp_h = nullptr, *mix_col_h = nullptr;
    uint16_t *mix_val_h = nullptr;
    int64_t mix_nnz_h = 0;
    // tiered sliced-ELL image of the same split (its_ellm.hip; matrices that hold the narrow ELL image): every (slice, slot)
    // run of 128 values in the narrowest type its entries' tiers allow; tier word and start (in 256-byte units) per slice
    unsigned char *ellm_img = nullptr;
    uint32_t *ellm_tier = nullptr, *ellm_off = nullptr;
    int64_t ellm_bytes = 0;
    // stream plan
    int n_blocks = 0;
    int4 *blk_desc = nullptr;                 // {first row, rows, first nnz, end nnz} per row block
    int n_blocks_big = 0;
    int4 *blk_desc_big = nullptr;             // same with blocks of up to SPMV_NNZ_BIG entries (rows <= 8 entries only)
    // irregular matrices without a panel image: the rows that do not fit a row block (> SPMV_NNZ_MAX entries: a block of
    // their own, walked chunk by chunk by ONE lane between workgroup barriers) are taken out of the plan — blk_desc_short is
    // blk_desc without those blocks — and get a wave each in a second launch (spmv_long_kernel), longest first
    int4 *blk_desc_short = nullptr;
    int n_blocks_short = 0;
    int *irr_long_rows = nullptr;
    int irr_n_long = 0;
    // vector-CSR plan
    int lanes_per_row = 0;
    int max_row_len = 0;                      // longest row: > IRR_ROW_LEN selects the irregular-row exact kernel
    // panel image (its_panel.hip; only for large irregular A): entries of the rows of <= PAN_LCUT entries sorted by
    // (tile, panel, row, position in row) as (key, value); chunk descriptors {first entry, entries | panel << 9} of
    // tile t (rows pan_tile_row[t] .. pan_tile_row[t+1]) at pan_desc[pan_tile_ptr[t] .. pan_tile_ptr[t+1]); the longer rows, longest first, in pan_long_rows
    uint32_t *pan_key = nullptr; double *pan_val = nullptr; int2 *pan_desc = nullptr; int *pan_tile_ptr = nullptr; int *pan_tile_row = nullptr;
    unsigned char *pan_long = nullptr; double *pan_diag = nullptr; int *pan_long_rows = nullptr;
    int pan_tiles = 0, pan_n_long = 0, pan_log2pw = 0, pan_chunks = 0;
    int64_t pan_nnz = 0;
    int64_t n_upper = 0;                      // entries with col > row (what a matching crs_U must hold)
    int64_t upper_col_sum = 0;                // their column sum and a hash of their values: crs_U is compared on all three
    uint64_t upper_val_xor = 0;
    // SELL-64-sigma image
    int sell_chunks = 0, sell_sigma = 0;
    int64_t sell_elems = 0;
    long long *sell_chunk_ptr = nullptr;     // 64-bit: padding can take the stored elements past 2^31
    int *sell_chunk_len = nullptr, *sell_col = nullptr, *sell_perm = nullptr;
    int *sell_grp_start = nullptr;           // [ITS_NPART + 1] chunk-group range of every workgroup slot (work-balanced)
    double *sell_val = nullptr;
};

struct SpmvArgs {
    const double *x = nullptr;     // gathered vector
    double *y = nullptr;           // output vector
    const double *b = nullptr;     // rhs for residual / Jacobi / GS epilogues
    const double *D = nullptr;     // diagonal for EPI_JACOBI_SHIPPED and EPI_SWEEP
    const double *dotv = nullptr;  // if set: part_sum[g] = sum_i (A x)_i * dotv_i   (EPI_STORE only)
    double *part_sum = nullptr;    // ITS_NPART partial sums (dot, or r^2 for EPI_RESIDUAL)
    double *part_max = nullptr;    // ITS_NPART partial max|r| (EPI_RESIDUAL, optional)
    // distributed runs: only rows of this class are processed (0 = all, 1 = interior, 2 = boundary)
    int row_begin = 0, row_end = -1;
    int part_accum = 0;            // add this launch's partial to part_sum/part_max instead of overwriting
    int map_mode = 0;              // work map of the stream kernel (see spmv_stream2_kernel)
    // EPI_GS_FUSED on the ELL kernels: rows [y2_lo, y2_hi) — the sweep's first level when its rows have no entries in L —
    // store (b - U x) / D, i.e. their next iterate, to y2 instead of b - U x to y (y2_lo even; y2_hi even or n_rows)
    double *y2 = nullptr;
    int y2_lo = 0, y2_hi = 0;
    // Gauss-Seidel, "the sweep's chain is the residual's prefix" (its_api.hip trip_gs, option "gs_prefix"): with ascending
    // columns a row of A is [entries of L | diagonal | entries of U], so the in-order chain of b - A x_new over the row
    // starts with exactly the chain the sweep of that row has just run over L (same entries, same order, same x).
    //   acc_out  EPI_SWEEP and the sweep kernels: the row's chain sum is stored here (may alias the sweep's rhs: every row
    //            reads its rhs before it stores)
    //   acc0     EPI_GS_FUSED over the image of D + U: the full chain of a row starts from acc0[row] instead of 0.0
    //            (rows in [y2_lo, y2_hi) with y2 set — the first level, swept by the pass itself — start from 0.0)
    const double *acc0 = nullptr;
    double *acc_out = nullptr;
};

void launch_spmv_exact(const CsrDev &A, const SpmvArgs &a, int epilogue, hipStream_t s);
// same kernel over an explicit list of row-block descriptors (interior / boundary split of the
// multi-GPU path); accumulate != 0 adds to the partials an earlier launch left
// launch tunables of the exact SpMV ("spmv_grid", "spmv_map", "spmv_variant", "spmv_idx", "spmv_valdict", "spmv_irr",
// "spmv_irr_len", "spmv_irr_grid", "spmv_bigplan"); 0 ok, -1 bad value, -2 unknown name
int spmv_tune(const char *name, int value);
void launch_sweep_empty(double *x, const double *rhs, const double *D, int64_t n, hipStream_t s, double *acc_out = nullptr);   // level without entries in L
void launch_spmv_exact_desc(const CsrDev &A, const int4 *desc, int n_desc, const SpmvArgs &a, int epilogue,
                            int accumulate, hipStream_t s);
// true when this launch would walk CsrDev::blk_desc_big (callers that pass descriptor lists of their own must
// derive them from the matching plan: only spmv_stream4_kernel has room for the larger blocks)
bool spmv_exact_takes_big_plan(const CsrDev &A, const SpmvArgs &a, int epilogue);
const char *spmv_exact_kernel_name(const CsrDev &A, const SpmvArgs &a, int epilogue, int64_t *streamed_bytes);
// sliced-ELL form of the exact SpMV (its_ell.hip): whole-matrix launches of eligible matrices
bool spmv_valdict_selected(const CsrDev &A);            // option "spmv_valdict" applies to this matrix
bool spmv_ell_eligible(const CsrDev &A, const SpmvArgs &a, int epilogue);
// ELL image of one contiguous sweep level of crs_L (its_api.hip build_levels, its_ell.hip launch_sweep_ell_view)
struct SweepEll { double *val = nullptr; uint32_t *idx = nullptr; int w = 0, blocks = 0, kind = 0, lo = 0, cnt = 0, pb = 0; };
int ell_choose_period(const int *dict, int n_off);    // row blocks per period of the ELL work map, 0 = none (its_ell.hip)
void launch_sweep_ell_view(const CsrDev &L, const SweepEll &v, double *x, const double *rhs, const double *D, hipStream_t s, double *acc_out = nullptr);
// the wave-per-row launch for a list of long rows (its_panel.hip); partial sums are ADDED to the ones a launch before it wrote
void launch_spmv_long(const CsrDev &A, const int *rows, int n_long, const SpmvArgs &a, int epilogue, hipStream_t s);
bool spmv_panel_eligible(const CsrDev &A, const SpmvArgs &a, int epilogue);     // its_panel.hip
void launch_spmv_panel(const CsrDev &A, const SpmvArgs &a, int epilogue, hipStream_t s);
int panel_tune(const char *name, int value);             // "spmv_panel", "spmv_panel_min_cols", "spmv_panel_log2pw", "spmv_panel_grid"
void launch_spmv_ell(const CsrDev &A, const SpmvArgs &a, int epilogue, hipStream_t s);
// explicit-column form (its_ellg.hip): builder (allocates A.ell_idx; ell_val, ellg_desc, ellg_base are the caller's) and launch
hipError_t ellg_build(CsrDev &A, hipStream_t s, int64_t *idx_words_out, int *hist);
void launch_spmv_ellg(const CsrDev &A, const SpmvArgs &a, int epilogue, int grid, int pb, int nt, hipStream_t s);
int ell_dict_forms_enabled();                            // option "spmv_ell_dict" (its_ell.hip): 0 = uploads build the explicit-column image instead
void launch_ell_build(const CsrDev &A, hipStream_t s, int row_base = 0);   // fills ell_val / ell_idx from the device CRS arrays
int ell_tune(const char *name, int value);               // "spmv_ell", "spmv_ell_grid", "spmv_ell_nt", "spmv_ell_map"
void launch_spmv_vector(const CsrDev &A, const SpmvArgs &a, int epilogue, hipStream_t s);
void launch_sell_fill(const CsrDev &A, hipStream_t s);   // sell_col / sell_val (zeroed) from the device CRS arrays, sell_perm, sell_chunk_ptr
void launch_spmv_sell(const CsrDev &A, const SpmvArgs &a, int epilogue, hipStream_t s);
void launch_spmv_mixed(const CsrDev &A, const SpmvArgs &a, int epilogue, hipStream_t s);   // EPI_STORE (+ dot), EPI_RESIDUAL
// tiered sliced-ELL form of the mixed image (its_ellm.hip)
hipError_t ellm_build(CsrDev &A, int mode, double t1, double t2, hipStream_t s);
bool spmv_ellm_eligible(const CsrDev &A, const SpmvArgs &a, int epilogue);
void launch_spmv_ellm(const CsrDev &A, const SpmvArgs &a, int epilogue, hipStream_t s);

// ---------------------------------------------------------------------------------------
// BLAS-1.  Reductions write ITS_NPART partials; reduce_finalize turns them into one scalar.
// ---------------------------------------------------------------------------------------
enum { FIN_SUM = 0, FIN_SQRT = 1, FIN_MAX = 2 };
void launch_dot_partials(const double *a, const double *b, int64_t n, double *part, hipStream_t s);
void launch_absmax_partials(const double *v, int64_t n, double *part, hipStream_t s);
void launch_reduce_finalize(const double *part, double *out, int mode, hipStream_t s);
void launch_sqrt_scalar(const double *in, double *out, hipStream_t s);   // out[0] = sqrt(in[0])
void launch_axpby(double *out, const double *a, const double *b, int64_t n, double sc, hipStream_t s);
void launch_scale(double *out, const double *v, double sc, int64_t n, hipStream_t s);
void launch_copy(double *out, const double *v, int64_t n, hipStream_t s);
void launch_fill(double *out, double v, int64_t n, hipStream_t s);

// ---------------------------------------------------------------------------------------
// CG (methods/cg.hpp:40-281).  Device scalars live in a small double array `sc`:
// ---------------------------------------------------------------------------------------
enum { SC_RR0 = 0, SC_RR1 = 1, SC_PQ = 2, SC_ALPHA = 3, SC_BETA = 4, SC_RES = 5, SC_TMP = 6, SC_COUNT = 16 };
// CG pass 1 (cg.hpp:162,182,186): alpha = rr_in[0] / sum(part_pq); r -= alpha q; part_rr[block] = partial r.r;
// block 0 stores p.q and alpha in sc.  If pq_is_scalar, part_pq[0] already holds the (all-reduced) sum.
void launch_cg_update_r(double *r, const double *q, int64_t n, const double *part_pq, int pq_is_scalar,
                        const double *rr_in, double *sc, double *part_rr, hipStream_t s);
// CG pass 2 (cg.hpp:181,188,202): beta = rr_new/rr_old[0]; x += sc[SC_ALPHA] p; p = r + beta p; block 0 stores
// rr_new -> rr_new[0] (a different slot than rr_old), sqrt(rr_new) -> res_out[0]
void launch_cg_update_xp(double *x, double *p, const double *r, int64_t n, const double *part_rr,
                         int rr_is_scalar, const double *rr_old, double *rr_new, double *sc,
                         double *res_out, hipStream_t s);

// ---------------------------------------------------------------------------------------
// GMRES modified Gram-Schmidt (methods/gmres.hpp:269-329,359,389) as fused axpy+dot steps.
// ---------------------------------------------------------------------------------------
// w -= h_prev * v_prev (h_prev = sum of part_prev, skipped when v_prev == nullptr);
// part_next = partial sums of w . v_next (v_next == nullptr: of w . w);  block 0 stores
// h_prev to h_out[0] when given.
void launch_mgs_step(double *w, const double *v_prev, const double *part_prev, int prev_is_scalar,
                     const double *v_next, double *part_next, double *h_out, int64_t n, hipStream_t s);
// v_out = w * (1/sqrt(sum part_ww)) ; block 0 stores sqrt(sum) to h_out[0]
void launch_mgs_normalize(double *v_out, const double *w, const double *part_ww, int ww_is_scalar,
                          double *h_out, int64_t n, hipStream_t s);
// The whole sweep — every step, ||w|| and v_{k+1} = w / ||w|| — as one persistent launch with w in registers
// (its_mgs.hip).  mgs_persist_grid: workgroups it would use for vectors of n doubles on this device, 0 = not applicable.
int mgs_persist_grid(int64_t n, int device);
size_t mgs_persist_slot_words(int m, int G);
hipError_t launch_mgs_persist(const double *w_in, const double *V, int64_t ld, int nv, double *v_out, double *h_out,
                              unsigned long long *slots, int parity, int m, int G, int64_t n, hipStream_t s, int inject_fail = 0);
hipError_t mgs_persist_reset(unsigned long long *slots, int m, int G, hipStream_t s);
// x = x0 + sum_{k<m} y[k] * V[k*ldv + i], in order k = 0..m-1 (gmres.hpp:940,971-972)
void launch_gmres_update_x(double *x, const double *x0, const double *V, int64_t ldv,
                           const double *y_dev, int m, int64_t n, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Communication-reducing variants (its_krylov_cr.hip; opt-in, outside the parity contract)
// ---------------------------------------------------------------------------------------
enum { SC_PCG0 = 8, SC_PCG1 = 11 };        // two triples {gamma_old, alpha_old, beta} of pipelined CG, by trip parity
void launch_pcg_dots(const double *r, const double *w, int64_t n, double *pair, hipStream_t s);   // pair: 2 x ITS_NPART
void launch_pcg_update(double *x, double *r, double *w, double *p, double *sv, double *z, const double *q, int64_t n,
                       const double *pair_in, double *pair_out, const double *sc_in, double *sc_out, int first, hipStream_t s);
int gram_blocks();                         // partials per basis vector of launch_gram_dots
void launch_gram_dots(const double *w, const double *V, int64_t ld, int nv, int64_t n, double *part, hipStream_t s);
void launch_gram_finalize(const double *part, int nv, double *h, hipStream_t s);
void launch_gram_axpy(double *w, const double *V, int64_t ld, int nv, const double *h, int64_t n, double *part_ww, hipStream_t s);
void launch_gram_combine(const double *h1, const double *h2, int nv, double *hcol, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Gauss-Seidel / triangular sweeps (kernels.hpp:451-473) by dependency level or colour.
// rows[] lists the rows of one level; x[i] = (rhs[i] - sum_j L_ij x_j) / D[i], in-order fma.
// ---------------------------------------------------------------------------------------
constexpr int SWEEP_STREAM_MIN = 4096;    // a contiguous level of at least this many rows runs on the exact SpMV kernels
constexpr int SWEEP_HUB = 512;            // a row of L this long is started first in its level
constexpr int SWEEP_SHORT = 16;            // rows of L up to this length take one lane, longer ones one wave
// rows[0, n_short) must be the short rows of the level, rows[n_short, n_level_rows) the long ones
void launch_sweep_level(const CsrDev &L, const int *rows, int n_level_rows, int n_short, double *x,
                        const double *rhs, const double *D, hipStream_t s, double *acc_out = nullptr);
// level-ordered copy of L (rows of at most SWEEP_ELL_W entries): slot-major (col, val) + diagonal per list position
constexpr int SWEEP_ELL_W = 16;           // (rows of L of a 27-point stencil have up to 13 entries)
void launch_sweep_ell_build(int n, int W, const int *rows, const CsrDev &L, const double *D, int *sw_col, double *sw_val,
                            double *sw_D, hipStream_t s);
// a run of consecutive levels of at most sweep_run_rows(W) list positions each, in one launch of one workgroup
int sweep_run_rows(int W);
void launch_sweep_ell_run(int n, int W, int l0, int l1, const int *level_ptr_dev, const int *rows, const int *sw_col,
                          const double *sw_val, const double *sw_D, double *x, const double *rhs, hipStream_t s, double *acc_out = nullptr);
void launch_sweep_ell_level(int n, int W, int p0, int p1, const int *rows, const int *sw_col, const double *sw_val,
                            const double *sw_D, double *x, const double *rhs, hipStream_t s, double *acc_out = nullptr);

// ---------------------------------------------------------------------------------------
// Natural-order sweep of a stencil matrix as ONE launch, a wavefront pipelined over grid lines (its_lines.hip).
// a = rows per line, b = rows per plane (0: 2-D), ny x nz lines, tiles of ty x tz lines per workgroup, ch rows per chunk;
// lv = the entries of L in three fixed slots (offset -b, -a, -1), slot-major; mask = presence bits per row.
// ---------------------------------------------------------------------------------------
struct LinesPlan {
    int n = 0, a = 0, b = 0, ny = 0, nz = 0, ty = 0, tz = 0, nby = 0, nbz = 0, ch = 0;
    int nsteps = 0, tslots = 0;                       // steps of a tile's sweep; slots (windows + a neutral one per step) of a tile
    double *img = nullptr;                            // operands in the order the sweep consumes them (its_lines.hip)
    unsigned int *mimg = nullptr;                     // presence bits per window
    double *rimg = nullptr, *ximg = nullptr;          // per sweep: right-hand side packed into, x produced in, the same order
    double *xedge = nullptr;                          // the lines on the tiles' high edges, row-major: what neighbouring tiles poll
    bool masked = true;                               // rows with entries missing inside the grid: presence bits decide (its_lines.hip)
    bool mute_first_tile = false;                     // test hook (option "sweep_lines_fault"): tile 0 publishes nothing
    int *sbase = nullptr, *tabs = nullptr;            // slots of a tile before step S; {active lanes, per-wave prefixes} of step S
    int *counter = nullptr, *fail = nullptr;          // tile counter; sticky "a poll gave up" flag (device)
    unsigned long long *fail_host = nullptr;          // the same, in pinned host memory (what the host checks)
};
size_t lines_image_doubles(const LinesPlan &P);
size_t lines_image_masks(const LinesPlan &P);
int lines_steps(const LinesPlan &P);
void lines_plan_geometry(LinesPlan &P);            // tile shape, steps and slots from a, b, ny, nz
void lines_step_tables(const LinesPlan &P, int *sbase, int *tabs);
size_t lines_edge_doubles(const LinesPlan &P);
void launch_lines_build(const LinesPlan &P, const CsrDev &L, const double *D, int *flags, hipStream_t s);
void launch_sweep_lines(const LinesPlan &P, double *x, const double *rhs, double *acc_out, hipStream_t s);

}  // namespace its
