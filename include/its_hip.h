/* its_hip.h — C ABI of libits_hip.so: the gfx950 (MI355X) implementation of the
 * DanecLacey/iterative_solvers solver hot path.
 *
 * The reference has no FFI: its seam is the set of Solver<VT> methods that preprocessing()
 * and solve_cpu() call with raw pointers.  Each entry point below names the
 * reference call site it replaces (paths relative to the reference root).  All functions
 * return 0 on success or a negative ITS_E* code; nothing throws, nothing calls exit().
 * Threading: one caller thread per handle; a handle is bound to the HIP device that was
 * current when it was created.  Host buffers stay owned by the caller, device memory by the
 * handle.  All values are fp64, all indices int32 (sparse_matrix.hpp:20-27).
 */
#ifndef ITS_HIP_H
#define ITS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct its_handle its_handle;

enum { ITS_OK = 0, ITS_EINVAL = -1, ITS_EHIP = -2, ITS_ESTATE = -3, ITS_ENOMEM = -4,
       ITS_ECOMM = -5, ITS_EUNSUPPORTED = -6 };

/* solver kinds = the strings Solver<VT> dispatches on (solver.hpp:411,441,467,509) */
enum { ITS_JACOBI = 0, ITS_GAUSS_SEIDEL = 1, ITS_CG = 2, ITS_GMRES = 3 };
/* Jacobi flavour: SHIPPED reproduces methods/jacobi.hpp:191 (SpMV arguments swapped),
 * REF is jacobi_iteration_ref_cpu (methods/jacobi.hpp:65-93), the correct sweep. */
enum { ITS_JACOBI_SHIPPED = 0, ITS_JACOBI_REF = 1 };
enum { ITS_PRECOND_NONE = 0, ITS_PRECOND_JACOBI = 1, ITS_PRECOND_GS = 2 };
/* SpMV formats (sparse_matrix.hpp:11-64 is the CRS layout; SELL-C-sigma layout follows
 * CUDA_tests/helpers/utilities.cpp:183-361 with C = 64).
 *   ITS_FMT_CSR_EXACT : one lane per row, in-order fused multiply-add from 0 — bit-identical
 *                       to spmv_crs_cpu (kernels.hpp:380-409); col/val streamed through LDS.
 *   ITS_FMT_CSR_VECTOR: 2^k lanes per row chosen from the row-length histogram; 1e-12 rel.
 *   ITS_FMT_SELL      : SELL-64-sigma; 1e-12 rel.
 *   ITS_FMT_MIXED     : mixed precision storage in the spirit of the reference's USE_AP "ap[dp_sp]" mode (main.cpp:171-184,
 *                       config.mk:56-63: entries split by a magnitude threshold): entries with |a_ij| >= threshold
 *                       (its_set_mixed_threshold) stay fp64, the others are STORED in fp32; x, y and the accumulation
 *                       stay fp64.  Opt-in, not bit-exact: per row |error| <= 2^-24 sum over the fp32 entries of
 *                       |a_ij x_j| (+ the reordering error of the vector variant).  8 instead of 12 bytes per fp32 entry. */
enum { ITS_FMT_CSR_EXACT = 0, ITS_FMT_CSR_VECTOR = 1, ITS_FMT_SELL = 2, ITS_FMT_MIXED = 3 };
/* which matrix of SparseMtxFormat<VT> (sparse_matrix.hpp:352-394) an upload fills */
enum { ITS_MAT_A = 0, ITS_MAT_L = 1, ITS_MAT_U = 2 };
/* Gauss-Seidel sweep order.  The library always executes the reference's forward substitution
 * (kernels.hpp:451-473) level by level (rows of one dependency level in parallel: identical
 * arithmetic per row, hence identical iterates).  ITS_GS_MULTICOLOR is a HOST-side choice: the
 * caller permutes the system with its_multicolor_order / its_permute_crs before uploading (the
 * Python and C++ hosts do), after which levels == colours.  its_solver_setup records the value. */
enum { ITS_GS_LEVEL = 0, ITS_GS_MULTICOLOR = 1 };

/* ---- library / handle --------------------------------------------------------------- */
/* start of run_solver (main.cpp:331).  device < 0 keeps the current device. */
int its_create(its_handle **h, int device);
/* HIP devices visible to this process (0 when there is none or the runtime cannot start): what a one-process-per-GPU
 * launcher maps its local rank onto (no counterpart in the reference: single process, main.cpp:94) */
int its_device_count(void);
/* end of run_solver (main.cpp:492-519) */
int its_destroy(its_handle *h);
const char *its_last_error(const its_handle *h);
const char *its_version(void);

/* ---- matrix / vectors --------------------------------------------------------------- */
/* after convert_to_crs (main.cpp:221-229); again for crs_L / crs_U
 * (methods/gauss_seidel.hpp:433-450).  Builds the row-block plan / SELL image on the host. */
int its_upload_crs(its_handle *h, int which, int n_rows, int n_cols, int64_t nnz,
                   const int *row_ptr, const int *col, const double *val, int format);
/* magnitude threshold of ITS_FMT_MIXED uploads that follow (AP_THRESHOLD_1 of config.mk:59): entries below it are
 * stored in fp32.  Default 1.0; threshold <= 0 keeps everything fp64, +inf stores everything in fp32. */
int its_set_mixed_threshold(its_handle *h, double threshold);
/* entries kept in fp64 / stored in fp32 by the last ITS_FMT_MIXED upload of this matrix (dp_percent / sp_percent of
 * main.cpp:186-190) */
int its_mixed_info(its_handle *h, int which, int64_t *nnz_fp64, int64_t *nnz_fp32);
/* The reference's AP_VALUE_TYPE (config.mk:57: "ap[dp_sp]", "ap[dp_hp]", "ap[sp_hp]", "ap[dp_sp_hp]") with AP_THRESHOLD_1 and
 * AP_THRESHOLD_2 (config.mk:59-62), for the ITS_FMT_MIXED uploads that follow: an entry with |a_ij| >= threshold_1 is stored
 * in the mode's first type, a smaller one in the second; in the three-type mode entries below threshold_2 in fp16.  An entry
 * its type cannot hold (|a_ij| > 65504 for fp16) stays one type up.  fp16 entries are the fp32-rounded value rounded to
 * binary16.  x, y and the accumulation stay fp64.  (partition_precisions itself, main.cpp:171-184, lives in the USpMV
 * library, which is not part of the reference tree: the split follows its parameters — parity unpinned.)
 * Matrices with rows of <= 8 entries and <= 15 distinct col - row are multiplied from a tiered sliced-ELL image in CRS
 * order: BIT-identical to spmv_crs_cpu (kernels.hpp:380-409) applied to the matrix of the stored values. */
enum { ITS_AP_DP_SP = 0, ITS_AP_DP_HP = 1, ITS_AP_SP_HP = 2, ITS_AP_DP_SP_HP = 3 };
int its_set_mixed_mode(its_handle *h, int ap_value_type, double threshold_1, double threshold_2);
/* entries per storage type of the last ITS_FMT_MIXED upload (nnz_by_tier[3]: fp64, fp32, fp16 — dp_percent, sp_percent,
 * hp_percent of main.cpp:186-190), the bytes of matrix VALUES a product streams, and whether it walks the tiered ELL image */
int its_mixed_tiers(its_handle *h, int which, int64_t *nnz_by_tier, int64_t *value_bytes, int *ell_form);
/* after extract_diag + generate_vector + scale_vector (main.cpp:268-276): D, b, x_old. */
int its_upload_vectors(its_handle *h, const double *D, const double *b, const double *x0, int n);
/* Solver::init_residual (solver.hpp:838-855) + both norms that
 * LoopParams::compute_stopping_criteria (structs.hpp:41-68) chooses between. */
int its_init_residual(its_handle *h, double *r0_inf, double *r0_two);
/* allocate_*_structs + Solver::init_structs (solver.hpp:335-378, 381-401). */
int its_solver_setup(its_handle *h, int solver_kind, int restart_len, int precond_kind,
                     int jacobi_mode, int gs_order);

/* Options (set before its_solver_setup):
 *   "gmres_givens" 0 = least squares by O(m) Givens rotation updates on the host (default, fast);
 *                  1 = the reference's formulation product for product (dense (m+1)^2 products
 *                      J, Q <- J Q, R = Q H, g = Q beta e1, methods/gmres.hpp:427-658): O(m^3) host
 *                      work per trip, reproduces the reference's |g[k+1]| also BELOW the attainable
 *                      accuracy, where the rotation form keeps shrinking and the dense form stalls.
 *   "cg_variant"   0 (default) = the reference's CG (methods/cg.hpp:40-281); 1 = pipelined CG (Ghysels-Vanroose): one
 *                  reduction point per trip instead of two, one fused vector pass — for row-partitioned runs, where a
 *                  reduction is an all-reduce.  OUTSIDE the parity contract: another recurrence, so another residual
 *                  history; it converges to the same solution (tests: x within 1e-8, trip count within 5 %).
 *   "gmres_ortho"  0 (default) = modified Gram-Schmidt (methods/gmres.hpp:269-329), one reduction per basis vector;
 *                  1 = classical Gram-Schmidt applied twice (CGS-2) with batched dot products: 3 reductions per trip
 *                  whatever the inner index.  Same standing as cg_variant 1.
 *   "gmres_persist" 1 (default) = the Gram-Schmidt sweep of a GMRES trip (gmres.hpp:269-389: every h_j = w . v_j,
 *                  w -= h_j v_j, ||w||, v_{k+1}) is ONE persistent launch that keeps w in registers and reads each
 *                  basis vector once (its_mgs.hip), when the run is on one GPU, uses modified Gram-Schmidt and the
 *                  vectors have between "gmres_persist_min_n" (default 1) and 12 * 960 * CUs (2.9 M) entries; the grid is
 *                  one workgroup per CU for long vectors and as few as 8 for short ones;
 *                  0 = one launch per basis vector.  Same operations per element; the dot products are summed in
 *                  another fixed order (H agrees to rounding, not bit for bit).  A launch whose workgroups cannot all
 *                  be resident (another process on the GPU) gives up after 20 ms and the trip is repeated the other way.
 *   "fused_residual" 1 (default) = a Jacobi / Gauss-Seidel trip reads A once: the pass that records
 *                  ||b - A x_new|| (utility_funcs.hpp:423-447) also produces the next trip's product of
 *                  x_new (second in-order chain per row; the shipped Jacobi sweep works on the full product, so there
 *                  the one chain serves both); 0 = separate launches.  Same bits either way.
 *   "gs_first_level_fused" 1 (default) = when the first level of the sweep has no entries in L (the first colour of a
 *                  multicolour order) and A runs on the ELL kernels, the one pass over A also stores that level's next
 *                  iterate (b - U x) / D; the sweep starts at the second level; 0 = separate kernel.  Same bits.
 *   "dist_overlap" 0 (default) = multi-GPU halo exchange first, then one product; 1 = exchange on a side
 *                  stream beside the interior row blocks, boundary blocks afterwards.  Same values.
 *   "sweep_stream" 1 (default) = sweep levels that are contiguous row ranges (the colours of a multicolour
 *                  ordering) run on the exact SpMV kernels; 0 = level-list kernels only.  Same bits.
 *   "sweep_run"    1 (default) = consecutive levels of at most 1024 rows of a natural-order sweep of short rows (all
 *                  levels of a 2-D stencil, the first and last ones of a 3-D one) run as ONE launch of one workgroup with
 *                  a barrier between levels instead of a launch per level; 0 = a launch per level.  Same bits.
 *   "sweep_ell_levels" 1 (default) = such a level gets an ELL image of its own rows of crs_L when they are short and
 *                  regular (colours of a stencil: nibble or byte dictionary indices, as for "spmv_ell"), built at
 *                  its_solver_setup; a level whose rows have no entries is one elementwise kernel.  Same bits.
 * Process-wide launch options of the exact SpMV (any time; all lossless, results bit-identical):
 *   "spmv_idx"     1 (default) = stream the 1-byte column-offset dictionary index instead of the 4-byte
 *                  column when the matrix has <= 256 distinct col - row; 0 = always the plain col stream
 *   "spmv_valdict" 1 = also stream values as 1-byte dictionary indices (<= 256 distinct fp64 values);
 *                  default 0
 *   "spmv_ell"     1 (default) = matrices with rows of <= 8 entries and <= 15 distinct col - row (5-/7-point
 *                  stencils, also halo-renumbered) run on a sliced ELL image built at upload (its_ell.hip): 8 W + 4
 *                  bytes per row instead of 9 nnz + 4, no LDS staging; 0 = the CRS stream kernels
 *   "spmv_ell_grid" workgroups (default 512 = 2 per CU), "spmv_ell_nt" 1 (default) = non-temporal store of the result,
 *   "spmv_ell_map" 1 (default) = row blocks dealt to the XCDs by their position inside the far band's period
 *   "spmv_ell_wide" 1 (default) = rows of <= 32 entries with <= 255 distinct col - row (27-point stencils) run on the
 *                  byte-index form of that image (8 W + 4 ceil(W/4) bytes per row); "spmv_ellw_grid" workgroups (768)
 *   "spmv_panel"   1 (default) = irregular matrices (longest row > 32) with at least "spmv_panel_min_cols" columns
 *                  (default 2^21: an x of 16 MB) and ascending columns in every row also get a column-panel blocked
 *                  image at upload (its_panel.hip: panels of 2^"spmv_panel_log2pw" doubles, default 16), so that the x
 *                  gathers of whole-matrix products hit L2; used for the plain, CG, residual, Jacobi and (image of
 *                  crs_U) Gauss-Seidel right-hand-side forms; 0 = the irregular-row kernel.  The two size options are
 *                  read when a matrix is uploaded, as is "spmv_panel_lcut" (128: rows longer than this get a wave each instead of a
 *                  place in the image; 64: 1.05 ms, 128: 0.95, 192: 0.96, 256: 0.99 on the 10 M-row power-law matrix).
 *                  "spmv_panel_grid" workgroups (1024)
 *   "spmv_irr"     1 (default) = two-phase kernel when the longest row exceeds 32 entries
 *   "spmv_irr_split" 1 (default) = rows of such a matrix with more than 256 entries (environment ITS_SPMV_SOLO_LEN, read at upload)
 *                  leave the plan and get a wave each in a second launch, longest first (as the rows outside a panel image
 *                  do); 0 = one lane walks them inside the two-phase kernel
 *   "spmv_variant" 4 (default) | 5 | 2, "spmv_grid", "spmv_irr_grid", "spmv_irr_len", "spmv_map" 0|1:
 *                  tuning knobs kept for A/B measurements (tools/spmv_ab.py) */
int its_set_option(its_handle *h, const char *name, int value);

/* ---- the loop body (solver_harness.hpp:40-116) -------------------------------------- */
/* Solver::iterate + record_residual_norm's value + Solver::exchange_arrays
 * (solver_harness.hpp:43-90).  Blocks until the scalar is on the host.  *residual_norm is what
 * the reference records: ||r||_2 (Jacobi/GS: of b - A x after the sweep; CG: of the
 * recurrence residual) or |g[k+1]| (GMRES, methods/gmres.hpp:658). */
int its_iterate(its_handle *h, int iter_count, double *residual_norm);
/* n_trips iterations back to back with ONE host sync at the end; residuals[n_trips] receives
 * every trip's value.  Same arithmetic as n_trips its_iterate calls (bench / fixed-count use). */
int its_iterate_n(its_handle *h, int iter_count0, int n_trips, double *residuals);
/* Solver::restart_gmres (solver_harness.hpp:98-104, solver.hpp:552-656) */
int its_gmres_restart(its_handle *h, int iter_count);
/* Solver::save_x_star (solver_harness.hpp:120, solver.hpp:658-726); iter_count as passed by
 * the harness (GMRES back-substitutes with iter_count - 1). */
int its_download_x(its_handle *h, double *x_star, int n, int iter_count);
/* post-loop record_residual_norm (solver_harness.hpp:122-149): Jacobi/GS recompute
 * ||b - A x_star||_2, CG returns the norm of Solver::r after the swaps, GMRES the last value */
int its_final_residual(its_handle *h, double *residual_norm);

/* ---- kernels alone (bench + parity tests; kernels.hpp) ------------------------------ */
/* where: 0 = x,y are host pointers (copied), 1 = device pointers.  kernels.hpp:380-409 */
int its_spmv(its_handle *h, int which, const double *x, double *y, int where, int format);
/* Device-pointer BLAS-1 with deterministic fixed-order reductions.  Result to *out (host). */
int its_dot(its_handle *h, const double *a_dev, const double *b_dev, int64_t n, double *out);      /* kernels.hpp:170-183 */
/* its_dot for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  Calls no hipMalloc / hipFree / allocating routine and does not
 * synchronise: the ITS_NPART partial sums go to a static 16 KB device array that every GPU the library runs on holds
 * until the process ends, the result (bit-identical to its_dot's) to *out_dev (device memory of the caller) in
 * stream order on the handle's stream — read it after its_synchronize.  n <= ITS_NOALLOC_MAX_N, else ITS_EINVAL.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two handles, two threads) race. */
#define ITS_NOALLOC_MAX_N ((int64_t)1 << 33)
int its_dot_noalloc(its_handle *h, const double *a_dev, const double *b_dev, int64_t n, double *out_dev);
int its_nrm2(its_handle *h, const double *v_dev, int64_t n, double *out);                          /* kernels.hpp:542-556 */
int its_norm_inf(its_handle *h, const double *v_dev, int64_t n, double *out);                      /* kernels.hpp:558-575 */
int its_axpby(its_handle *h, double *out_dev, const double *a_dev, const double *b_dev,
              int64_t n, double s);          /* out = a + s*b: sum_vectors_cpu / subtract (s<0) kernels.hpp:54-87 */
int its_scale(its_handle *h, double *out_dev, const double *v_dev, double s, int64_t n);           /* kernels.hpp:136-168 */
/* forward solve (D+L) x = rhs, device pointers; kernels.hpp:451-473 (needs ITS_MAT_L + D) */
int its_spltsv(its_handle *h, double *x_dev, const double *rhs_dev);

/* plain device memory helpers for callers without a HIP runtime of their own */
int its_dev_alloc(its_handle *h, void **p, size_t bytes);
int its_dev_free(its_handle *h, void *p);
int its_memcpy_h2d(its_handle *h, void *dst_dev, const void *src, size_t bytes);
int its_memcpy_d2h(its_handle *h, void *dst, const void *src_dev, size_t bytes);
int its_synchronize(its_handle *h);

/* ---- introspection for bench.py ----------------------------------------------------- */
/* average duration (ms) of the last its_spmv_timed / its_iterate_n region, measured with HIP
 * events on the handle's own stream */
/* with_dot == 1 launches the variant the CG loop uses (y = A x fused with the partial sums
 * of y.x, methods/cg.hpp:128,160) so its duration matches the in-loop kernel; 2 and 3 time the residual
 * epilogue (r = x - A x, with x standing in for b) with r stored (2) and norm-only (3); 4..7 time the
 * Jacobi sweep, the GS right-hand side and their fused forms (sweep / rhs + recorded residual) the same way. */
int its_spmv_timed(its_handle *h, int which, const double *x_dev, double *y_dev, int format,
                   int with_dot, int reps, float *avg_ms);
int its_last_region_ms(its_handle *h, float *ms);
/* average duration (ms) of the SpMV launch INSIDE the last its_iterate_n CG region (HIP event
 * pairs around each of the first 256 launches, on the handle's stream) */
int its_last_spmv_ms(its_handle *h, float *ms);
/* which kernel family a whole-matrix exact SpMV (with_dot = 1: the CG form) of this matrix runs on — "spmv_ell_kernel", "spmv_ellw_kernel", "spmv_ellg_kernel",
 * "spmv_stream4_kernel", "spmv_stream2_kernel", "spmv_stream5_kernel", "spmv_irr_kernel" — and the bytes one launch
 * is asked to stream in that kernel's storage form (matrix arrays + x once + y once) */
int its_spmv_kernel_info(its_handle *h, int which, int with_dot, char *name, int cap, int64_t *streamed_bytes);
/* plan facts: lanes per row chosen, SELL occupancy (nnz / stored elements), row-block count */
int its_plan_info(its_handle *h, int which, int format, int *lanes_per_row, double *sell_occupancy,
                  int *n_row_blocks);
/* sliced-ELL image of an uploaded matrix (what sparse_matrix.hpp:11-64's CRS arrays become for the exact SpMV of short
 * rows): kind 0 = none, 1 = nibble dictionary indices (rows <= 8 entries, <= 15 distinct col - row), 2 = byte dictionary
 * indices (rows <= 32, <= 255), 3 = explicit columns as a base per (slice of 128 rows, slot) + 8/16/32-bit deltas (rows
 * <= 8, any columns); width = slots per row; index_words = 32-bit index words stored; slices_by_delta_width[3] = slices
 * of kind 3 that use 8- / 16- / 32-bit deltas (zeros otherwise).  Any pointer may be NULL. */
int its_ell_info(its_handle *h, int which, int *kind, int *width, int64_t *index_words, int *slices_by_delta_width);
/* after its_solver_setup(ITS_GMRES): workgroups of the persistent Gram-Schmidt launch ("gmres_persist"; 0 = the trips
 * run one launch per basis vector) and how many trips had to be repeated that way because the launch gave up */
int its_gmres_info(its_handle *h, int *persist_grid, int *fallbacks);
/* after its_solver_setup(ITS_GAUSS_SEIDEL): prefix_ready = 1 when the trips run with option "gs_prefix" (default 1): the
 * set-up has verified on the device that every row of A is [its row of crs_L | one diagonal entry | its row of crs_U],
 * bit for bit and in that order (what gauss_seidel.hpp:249-630 builds), so the chain of the recorded residual b - A x_new
 * (utility_funcs.hpp:444-447) starts with the chain the sweep (kernels.hpp:460-472) has just run: the sweep leaves its
 * sums behind and the residual pass walks the rows from the diagonal on only (nnz_du entries instead of nnz of A), same
 * bits.  kernel = the family that pass runs on (as its_spmv_kernel_info), "" when prefix_ready = 0 */
int its_gs_info(its_handle *h, int *prefix_ready, int64_t *nnz_du, char *kernel, int kernel_cap);
/* after its_solver_setup with crs_L: how the forward substitution of kernels.hpp:451-473 is scheduled.  n_levels = launches
 * (dependency levels, or colours) of the level schedule.  lines_workgroups > 0: the sweep runs instead as ONE launch
 * pipelined over grid lines (csrc/its_lines.hip; stencil matrices whose strictly lower offsets are {-1, -a[, -b]}; option
 * "sweep_lines": 0 never, 1 (default) wherever the matrix qualifies), bit-identical to the level schedule, x may be rhs;
 * line_length = a, rows_per_chunk = rows a lane advances per step */
int its_sweep_info(its_handle *h, int *lines_workgroups, int *line_length, int *rows_per_chunk, int *n_levels);

/* ---- Gauss-Seidel ordering helpers (host only, no GPU needed) -------------------------
 * methods/gauss_seidel.hpp sweeps in natural order; its_solver_setup schedules that sweep by
 * dependency level.  For a MULTICOLOUR sweep the caller permutes the system symmetrically with
 * the order below (rows colour by colour) and uploads P A P^T, P b, P x0: the levels of the
 * permuted strict-lower part are then the colours, and a CPU reference run on the same permuted
 * matrix performs the identical arithmetic. */
/* extract_diag (utility_funcs.hpp:157-174) and split_L_U + convert_to_crs (utility_funcs.hpp:240-324,
 * methods/gauss_seidel.hpp:433-450) on the host, OpenMP-parallel: what a caller that holds A in CRS needs before
 * its_upload_vectors / the ITS_MAT_L, ITS_MAT_U uploads.  L / U outputs sized like the inputs. */
int its_extract_diag(int n, const int *row_ptr, const int *col, const double *val, double *D);
int its_split_lu(int n, const int *row_ptr, const int *col, const double *val,
                 int *Lp, int *Lc, double *Lv, int *Up, int *Uc, double *Uv);
/* greedy distance-1 colouring of the graph of A + A^T in natural vertex order;
 * perm_new2old[n]: rows listed colour by colour, ascending inside a colour */
int its_multicolor_order(int n, const int *row_ptr, const int *col, int *perm_new2old, int *n_colors);
/* B = P A P^T in CRS, columns ascending inside a row (outputs sized like the inputs) */
int its_permute_crs(int n, const int *row_ptr, const int *col, const double *val, const int *perm_new2old,
                    int *row_ptr_out, int *col_out, double *val_out);

/* ---- multi-GPU (no counterpart in the reference, which is single-process:
 * "TODO: MPI preprocessing will go here", main.cpp:94) ----------------------------------------
 * One process per GPU, contiguous row blocks.  Order of calls on every rank:
 *   its_comm_init (or its_comm_init_host) -> its_upload_crs(ITS_MAT_A, n_rows = owned rows,
 *   n_cols = owned + halo columns, local column indices) -> its_dist_setup -> its_upload_vectors
 *   -> its_init_residual -> its_solver_setup -> its_iterate ...
 * Every SpMV of the solver loops then packs/exchanges the halo entries of its input vector
 * (overlapped with the interior rows) and every dot/norm is all-reduced, all on device streams. */
/* rank 0 makes an id (128 bytes, RCCL ncclUniqueId); the caller's control plane hands TWO ids
 * (halo communicator, reduction communicator) to every rank */
int its_comm_unique_id(void *id128);
/* Blocks until every rank has joined, at most ITS_COMM_TIMEOUT_S (default 180) seconds.  After ITS_ECOMM from a time-out
 * the handle is unusable for row-partitioned runs (a later its_comm_init / its_dist_setup on it returns ITS_ESTATE; the
 * communicators the late peers may still complete are aborted by the library): the rank must exit non-zero and the job be
 * restarted with FRESH processes — never re-exec a process that has initialised the GPU. */
int its_comm_init(its_handle *h, int rank, int world, const void *ids256);
/* Test transport: halo data and scalars are staged through the host and exchanged by the
 * caller's callbacks (return 0 on success), so several ranks can share one GPU.
 *   exchange : send[send_off[k]..send_off[k+1]) goes to rank nbr[k];
 *              recv[recv_off[k]..recv_off[k+1]) must be filled with what nbr[k] sent to us.
 *   allreduce: vals[0..count) replaced by the sum (op 0) or max (op 1) over all ranks. */
typedef int (*its_exchange_cb)(void *user, const double *send, const int *send_off, double *recv,
                               const int *recv_off, const int *nbr, int n_nbr);
typedef int (*its_allreduce_cb)(void *user, double *vals, int count, int op);
int its_comm_init_host(its_handle *h, int rank, int world, its_exchange_cb exchange,
                       its_allreduce_cb allreduce, void *user);
/* Exchange plan for the uploaded local matrix: n_nbr neighbour ranks (ascending);
 * send_idx[send_off[k]..send_off[k+1]) = owned rows whose x entries neighbour k needs, in the
 * order of that neighbour's halo; halo columns n_rows + [recv_off[k], recv_off[k+1]) come from
 * neighbour k.  Splits the row blocks into interior / boundary for the overlap. */
int its_dist_setup(its_handle *h, int n_nbr, const int *nbr, const int *send_off,
                   const int *send_idx, const int *recv_off);

/* Triangular sweeps (Gauss-Seidel, -p gs) across ranks (methods/gauss_seidel.hpp:144,177,227 on one process): the
 * natural-order sweep is sequential from rank to rank, so a row-partitioned sweep runs in GLOBAL multicolour order —
 * the iteration its_solver_setup's ITS_GS_MULTICOLOR documents, identical arithmetic to the reference's sweep on the
 * colour-permuted global matrix.  The caller orders every rank's owned rows colour by colour (colours from one global
 * colouring) and passes the cuts: color_ptr[n_colors + 1] over the owned rows, and per neighbour k the colour cuts of
 * its send list and halo range, send_color_off / recv_color_off[k * (n_colors + 1) + c] (absolute offsets; entry 0 and
 * entry n_colors of each row equal send_off / recv_off of its_dist_setup).  crs_L then holds, per row, the entries
 * whose column has an earlier colour (owned or halo), crs_U the rest off the diagonal.  Each colour step ends with an
 * exchange of that colour's boundary values.  Call after its_dist_setup, before its_solver_setup. */
int its_dist_set_colors(its_handle *h, int n_colors, const int *color_ptr, const int *send_color_off,
                        const int *recv_color_off);

#ifdef __cplusplus
}
#endif
#endif /* ITS_HIP_H */
