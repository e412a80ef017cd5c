/* its_oracle.h — CPU restatement of the DanecLacey/iterative_solvers hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under iterative_solvers_amd/ may include, link or load
 * this: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / reported baseline.  Parity is PINNED: every function below is validated in this
 * container against the compiled, unmodified reference (oracle/_ref, built by oracle/Makefile)
 * — identical printed residual histories at OMP_NUM_THREADS=1 — and against the known-answer
 * tables of /root/reference/test_suite/test_data.cpp (tests/test_oracle_*.py).
 *
 * Arithmetic note: the reference is compiled -O3 with FMA available, so g++ contracts every
 * `acc += a*b`, `a + s*b`, `a - s*b` into one fused operation.  Here the fusion is written out
 * (__builtin_fma) and the file is compiled -ffp-contract=off, so results do not depend on
 * compiler flags.
 */
#ifndef ITS_ORACLE_H
#define ITS_ORACLE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_JACOBI = 0, ORC_GAUSS_SEIDEL = 1, ORC_CG = 2, ORC_GMRES = 3 };
enum { ORC_SCALE_NONE = 0, ORC_SCALE_DIAG = 1, ORC_SCALE_MAX = 2 };
enum { ORC_PRECOND_NONE = 0, ORC_PRECOND_JACOBI = 1, ORC_PRECOND_GS = 2 };
/* Jacobi flavour: SHIPPED = bug-compatible with methods/jacobi.hpp:191 (swapped SpMV args);
 * REF = the unused, correct jacobi_iteration_ref_cpu (methods/jacobi.hpp:65-93). */
enum { ORC_JACOBI_SHIPPED = 0, ORC_JACOBI_REF = 1 };

/* ---- L0: input ----------------------------------------------------------------------- */
/* read_mtx (io_funcs.hpp:125-289): banner check, symmetric expansion, stable row sort.
 * Returns 0, or <0 on error.  Arrays are malloc'ed; release with orc_free. */
int  orc_read_mtx(const char *path, int *n_rows, int *n_cols, long *nnz,
                  int **I, int **J, double **V);
void orc_free(void *p);
/* preprocessing() equilibration (main.cpp:117-132) applied in place to COO values; the
 * per-row scaling vector used later by scale_vector (utility_funcs.hpp:518-530) is returned
 * in row_scale[n].  ORC_SCALE_MAX reproduces the shipped copy-paste bug (main.cpp:121-124). */
void orc_scale_coo(int n, long nnz, const int *I, const int *J, double *V, int scale_kind,
                   double *row_scale);
/* convert_to_crs (utility_funcs.hpp:326-377).  row_ptr[n+1], col[nnz], val[nnz] caller-owned. */
int  orc_coo_to_crs(int n, long nnz, const int *I, const int *J, const double *V,
                    int *row_ptr, int *col, double *val);
/* extract_diag (utility_funcs.hpp:157-174) from CRS; entries without a diagonal stay 0. */
void orc_extract_diag(int n, const int *row_ptr, const int *col, const double *val, double *D);
/* split_L_U (utility_funcs.hpp:240-324) + convert_to_crs, from CRS: strict lower / strict
 * upper.  Pass NULL arrays with *nnz_out to size first. */
void orc_split_LU(int n, const int *row_ptr, const int *col, const double *val,
                  int *L_row_ptr, int *L_col, double *L_val,
                  int *U_row_ptr, int *U_col, double *U_val);

/* ---- L1: kernels (kernels.hpp) ------------------------------------------------------- */
void   orc_spmv(int n, const int *row_ptr, const int *col, const double *val,
                const double *x, double *y);                         /* :380-409 */
double orc_dot(const double *a, const double *b, long n);            /* :170-183 */
void   orc_sum_vectors(double *out, const double *a, const double *b, long n, double s);      /* :54-71 */
void   orc_subtract_vectors(double *out, const double *a, const double *b, long n, double s); /* :73-87 */
void   orc_scale(double *out, const double *v, double s, long n);    /* :136-168 */
double orc_nrm2(const double *v, long n);                            /* :542-556 */
double orc_norm_inf(const double *v, long n);                        /* :558-575 */
void   orc_spltsv(int n, const int *L_row_ptr, const int *L_col, const double *L_val,
                  double *x, const double *D, const double *rhs);    /* :451-473 */
void   orc_residual(int n, const int *row_ptr, const int *col, const double *val,
                    const double *x, const double *b, double *r, double *tmp);   /* :670-726 */

/* ---- L2-L4: whole solve (solver.hpp + solver_harness.hpp + methods/...) ------------ */
typedef struct {
    int    solver;        /* ORC_* */
    int    jacobi_mode;   /* ORC_JACOBI_* */
    int    precond;       /* ORC_PRECOND_* (GMRES only, like the reference) */
    int    max_iters;     /* MAX_ITERS */
    int    restart_len;   /* GMRES_RESTART_LEN */
    double tol;           /* TOL */
    int    fixed_iters;   /* >0: ignore convergence, run exactly this many trips (timing) */
} orc_params;

typedef struct {
    int    iters;             /* loop_params->iter_count at exit */
    int    residual_count;    /* number of history lines the reference prints */
    int    converged;
    double stopping;          /* tol * ||r0|| (mixed-norm rule, structs.hpp:41-68) */
    double final_residual;    /* normed_residuals[residual_count] after the post-loop call */
    double r0_inf, r0_two;
    double solve_seconds;     /* wall time of the do..while loop only */
} orc_result;

/* b and x0 are the already-scaled vectors (main.cpp:270-276).  history must hold
 * max_iters + 1 doubles.  x_star[n] receives save_x_star's result. */
int orc_solve(int n, const int *row_ptr, const int *col, const double *val,
              const double *b, const double *x0, const orc_params *p,
              double *history, double *x_star, orc_result *out);

/* Matrix Market text of a CRS matrix ("%%MatrixMarket matrix coordinate real general", 1-based, %.17g): the input file of
 * the compiled reference for the CPU calibration leg of bench.py (io_funcs.hpp:125-289 reads it back).  0 = ok. */
int orc_write_mtx(const char *path, int n_rows, int n_cols, const int *row_ptr, const int *col, const double *val);

int orc_num_threads(void);
/* 1 thread = bit-identical to the compiled reference at OMP_NUM_THREADS=1 */
void orc_set_threads(int t);

#ifdef __cplusplus
}
#endif
#endif
