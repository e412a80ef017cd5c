// its_oracle.cpp — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY
// (see its_oracle.h).  Each function cites the reference file:line it restates.  The code is
// written from the reference's *behaviour*; it shares no text with it.
//
// Compile: g++ -O3 -fopenmp -mavx2 -mfma -ffp-contract=off (oracle/Makefile).
#include "its_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>
#include <omp.h>

#define FMA(a, b, c) __builtin_fma((a), (b), (c))

// ===================================================================== L0: input / layout
static void lower(char *s) { for (; *s; ++s) if (*s >= 'A' && *s <= 'Z') *s += 32; }

// Restates read_mtx (io_funcs.hpp:125-289) + mm_read_unsymmetric_sparse (mmio.h:133-251):
// coordinate real/integer/pattern, general/symmetric; pattern values = 0.01; 1-based -> 0-based;
// symmetric entries mirrored right after their original (io_funcs.hpp:217-231); finally a
// STABLE sort by row only (io_funcs.hpp:12-19,258), so within-row order is file order.
extern "C" int orc_read_mtx(const char *path, int *n_rows, int *n_cols, long *nnz_out,
                            int **I_out, int **J_out, double **V_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[1100];
    if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
    char banner[64], obj[64], fmt[64], field[64], symm[64];
    if (sscanf(line, "%63s %63s %63s %63s %63s", banner, obj, fmt, field, symm) != 5) { fclose(f); return -2; }
    if (strcmp(banner, "%%MatrixMarket") != 0) { fclose(f); return -2; }   // mmio.h:20
    lower(obj); lower(fmt); lower(field); lower(symm);
    bool pattern = !strcmp(field, "pattern");
    bool okfield = pattern || !strcmp(field, "real") || !strcmp(field, "integer");
    bool symmetric = !strcmp(symm, "symmetric");
    if (strcmp(obj, "matrix") || strcmp(fmt, "coordinate") || !okfield ||
        !(symmetric || !strcmp(symm, "general"))) { fclose(f); return -3; }
    int M = 0, N = 0, nz = 0;
    for (;;) {
        if (!fgets(line, sizeof line, f)) { fclose(f); return -4; }
        if (line[0] == '%') continue;
        if (sscanf(line, "%d %d %d", &M, &N, &nz) == 3) break;
    }
    if (M != N) { fclose(f); return -5; }                                   // io_funcs.hpp:186-190
    std::vector<int> ri, ci; std::vector<double> vv;
    ri.reserve(symmetric ? 2L * nz : nz); ci.reserve(ri.capacity()); vv.reserve(ri.capacity());
    for (int k = 0; k < nz; ++k) {
        int r, c; double v = 0.01;
        if (pattern) { if (fscanf(f, "%d %d\n", &r, &c) != 2) { fclose(f); return -6; } }
        else         { if (fscanf(f, "%d %d %lg\n", &r, &c, &v) != 3) { fclose(f); return -6; } }
        --r; --c;
        ri.push_back(r); ci.push_back(c); vv.push_back(v);
        if (symmetric && r != c) { ri.push_back(c); ci.push_back(r); vv.push_back(v); }
    }
    fclose(f);
    long nnz = (long)ri.size();
    std::vector<int> perm(nnz);
    for (long k = 0; k < nnz; ++k) perm[k] = (int)k;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return ri[a] < ri[b]; });
    int *I = (int *)malloc(sizeof(int) * std::max<long>(nnz, 1));
    int *J = (int *)malloc(sizeof(int) * std::max<long>(nnz, 1));
    double *V = (double *)malloc(sizeof(double) * std::max<long>(nnz, 1));
    for (long k = 0; k < nnz; ++k) { I[k] = ri[perm[k]]; J[k] = ci[perm[k]]; V[k] = vv[perm[k]]; }
    *n_rows = M; *n_cols = N; *nnz_out = nnz; *I_out = I; *J_out = J; *V_out = V;
    return 0;
}

extern "C" void orc_free(void *p) { free(p); }

// Restates the equilibration block of preprocessing() (main.cpp:117-132) with
// extract_diag(take_sqrt) (utility_funcs.hpp:157-174), extract_largest_{row,col}_elems
// (:760-798) and scale_matrix_{rows,cols} (:800-820).  Two successive divisions per value.
extern "C" void orc_scale_coo(int n, long nnz, const int *I, const int *J, double *V,
                              int scale_kind, double *row_scale)
{
    std::vector<double> cs(n, 1.0);
    for (int i = 0; i < n; ++i) row_scale[i] = 1.0;
    if (scale_kind == ORC_SCALE_MAX) {
        // shipped behaviour: BOTH passes write the row vector; the column vector stays 1.0
        for (long k = 0; k < nnz; ++k) { double a = std::fabs(V[k]); if (a > row_scale[I[k]]) row_scale[I[k]] = a; }
        for (long k = 0; k < nnz; ++k) { double a = std::fabs(V[k]); if (a > row_scale[J[k]]) row_scale[J[k]] = a; }
    } else if (scale_kind == ORC_SCALE_DIAG) {
        for (long k = 0; k < nnz; ++k)
            if (I[k] == J[k]) { double s = std::sqrt(std::fabs(V[k])); row_scale[I[k]] = s; cs[I[k]] = s; }
    }
    for (long k = 0; k < nnz; ++k) V[k] = V[k] / row_scale[I[k]];
    for (long k = 0; k < nnz; ++k) V[k] = V[k] / cs[J[k]];
}

// Restates convert_to_crs (utility_funcs.hpp:326-377): col/val copied in COO order, row_ptr
// from a per-row count (the COO is row-sorted already).
extern "C" int orc_coo_to_crs(int n, long nnz, const int *I, const int *J, const double *V,
                              int *row_ptr, int *col, double *val)
{
    for (long k = 0; k < nnz; ++k) { col[k] = J[k]; val[k] = V[k]; }
    std::vector<int> cnt(n, 0);
    for (long k = 0; k < nnz; ++k) ++cnt[I[k]];
    row_ptr[0] = 0;
    for (int i = 0; i < n; ++i) row_ptr[i + 1] = row_ptr[i] + cnt[i];
    return row_ptr[n] == nnz ? 0 : -1;
}

extern "C" void orc_extract_diag(int n, const int *row_ptr, const int *col, const double *val, double *D)
{
    for (int i = 0; i < n; ++i)
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k)
            if (col[k] == i) D[i] = val[k];     // later duplicates overwrite, like the COO sweep
}

// split_L_U (utility_funcs.hpp:240-324): col<row -> L, col>row -> U, diagonal dropped; order kept.
extern "C" void orc_split_LU(int n, const int *row_ptr, const int *col, const double *val,
                             int *Lp, int *Lc, double *Lv, int *Up, int *Uc, double *Uv)
{
    int nl = 0, nu = 0;
    Lp[0] = 0; Up[0] = 0;
    for (int i = 0; i < n; ++i) {
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            if (col[k] < i)      { if (Lc) { Lc[nl] = col[k]; Lv[nl] = val[k]; } ++nl; }
            else if (col[k] > i) { if (Uc) { Uc[nu] = col[k]; Uv[nu] = val[k]; } ++nu; }
        }
        Lp[i + 1] = nl; Up[i + 1] = nu;
    }
}

// ===================================================================== L1: kernels
// spmv_crs_cpu (kernels.hpp:380-409): per row, accumulator from 0, strictly left-to-right,
// each step fused (measured against the compiled reference).
extern "C" void orc_spmv(int n, const int *row_ptr, const int *col, const double *val,
                         const double *x, double *y)
{
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) acc = FMA(val[k], x[col[k]], acc);
        y[i] = acc;
    }
}

// dot (kernels.hpp:170-183).  As compiled by g++ 11 -O3 -mavx2 -mfma (the oracle/_ref recipe)
// the reduction loop is vectorised "in order": products are rounded, then added left to right;
// only a lone trailing element of an odd-length thread chunk is fused.  (nrm2, by contrast,
// compiles to a fused chain.)  Measured with oracle/ref_kernel_probe.cpp over many lengths.
extern "C" double orc_dot(const double *a, const double *b, long n)
{
    double sum = 0.0;
    #pragma omp parallel reduction(+ : sum)
    {
        long nt = omp_get_num_threads(), t = omp_get_thread_num();
        long q = n / nt, r = n % nt;
        long lo = t < r ? t * (q + 1) : r * (q + 1) + (t - r) * q;
        long len = t < r ? q + 1 : q, hi = lo + len;
        double acc = 0.0;
        long even_end = hi - (len & 1);
        for (long i = lo; i < even_end; ++i) { double pr = a[i] * b[i]; acc += pr; }
        if (len & 1) acc = FMA(a[hi - 1], b[hi - 1], acc);
        sum += acc;
    }
    return sum;
}

extern "C" void orc_sum_vectors(double *out, const double *a, const double *b, long n, double s)
{
    #pragma omp parallel for
    for (long i = 0; i < n; ++i) out[i] = FMA(s, b[i], a[i]);              // a + s*b
}

extern "C" void orc_subtract_vectors(double *out, const double *a, const double *b, long n, double s)
{
    #pragma omp parallel for
    for (long i = 0; i < n; ++i) out[i] = FMA(-s, b[i], a[i]);             // a - s*b (fnmadd)
}

extern "C" void orc_scale(double *out, const double *v, double s, long n)
{
    #pragma omp parallel for
    for (long i = 0; i < n; ++i) out[i] = v[i] * s;
}

// euclidean_vec_norm_cpu (kernels.hpp:542-556).  Compiled form (same recipe as dot above):
// whole groups of 4 are square-rounded then added in order; the (len mod 4) tail elements of a
// thread chunk (or all of a chunk shorter than 4) are fused.
extern "C" double orc_nrm2(const double *v, long n)
{
    double sum = 0.0;
    #pragma omp parallel reduction(+ : sum)
    {
        long nt = omp_get_num_threads(), t = omp_get_thread_num();
        long q = n / nt, r = n % nt;
        long lo = t < r ? t * (q + 1) : r * (q + 1) + (t - r) * q;
        long len = t < r ? q + 1 : q, hi = lo + len;
        long body_end = len > 3 ? lo + (len & ~3L) : lo;
        double acc = 0.0;
        for (long i = lo; i < body_end; ++i) { double pr = v[i] * v[i]; acc += pr; }
        for (long i = body_end; i < hi; ++i) acc = FMA(v[i], v[i], acc);
        sum += acc;
    }
    return std::sqrt(sum);
}

// infty_vec_norm_cpu (kernels.hpp:558-575): serial max-abs
extern "C" double orc_norm_inf(const double *v, long n)
{
    double m = 0.0;
    for (long i = 0; i < n; ++i) { double a = std::fabs(v[i]); if (a > m) m = a; }
    return m;
}

// spltsv_crs (kernels.hpp:451-473): serial forward substitution, natural row order
extern "C" void orc_spltsv(int n, const int *Lp, const int *Lc, const double *Lv,
                           double *x, const double *D, const double *rhs)
{
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = Lp[i]; k < Lp[i + 1]; ++k) s = FMA(Lv[k], x[Lc[k]], s);
        x[i] = (rhs[i] - s) / D[i];
    }
}

// calc_residual_cpu (kernels.hpp:670-726): tmp = A x ; r = b - 1.0*tmp
extern "C" void orc_residual(int n, const int *row_ptr, const int *col, const double *val,
                             const double *x, const double *b, double *r, double *tmp)
{
    orc_spmv(n, row_ptr, col, val, x, tmp);
    #pragma omp parallel for
    for (int i = 0; i < n; ++i) r[i] = b[i] - tmp[i];
}

// dense_MMM / dense_MMM_t_t (kernels.hpp:475-495,519-540): row-major naive, fused chain over k
static void dense_mm(const double *A, const double *B, double *C, int nrA, int ncA, int ncB)
{
    for (int i = 0; i < nrA; ++i)
        for (int j = 0; j < ncB; ++j) {
            double t = 0.0;
            for (int k = 0; k < ncA; ++k) t = FMA(A[(size_t)i * ncA + k], B[(size_t)k * ncB + j], t);
            C[(size_t)i * ncB + j] = t;
        }
}

extern "C" int orc_write_mtx(const char *path, int n_rows, int n_cols, const int *rp, const int *col, const double *val)
{
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n", n_rows, n_cols, n_rows > 0 ? rp[n_rows] : 0);
    // hand-rolled line assembly: integers by division, the %.17g text of a value cached for the last few distinct bit
    // patterns (stencil matrices have two), so that a 15 M-line file takes a second instead of ten
    std::vector<char> buf((1 << 22) + 128);
    size_t fill = 0;
    struct Memo { uint64_t bits; int len; char txt[32]; } memo[8];
    int n_memo = 0, next_memo = 0;
    auto put_int = [&](int v) {
        char tmp[12]; int q = 0;
        do { tmp[q++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (q) buf[fill++] = tmp[--q];
    };
    int bad = 0;
    for (int i = 0; i < n_rows; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k) {
            put_int(i + 1); buf[fill++] = ' ';
            put_int(col[k] + 1); buf[fill++] = ' ';
            uint64_t bits; memcpy(&bits, &val[k], 8);
            const Memo *hit = nullptr;
            for (int q = 0; q < n_memo; ++q) if (memo[q].bits == bits) { hit = &memo[q]; break; }
            if (!hit) {
                Memo &mm = memo[next_memo]; next_memo = (next_memo + 1) & 7; if (n_memo < 8) ++n_memo;
                mm.bits = bits; mm.len = snprintf(mm.txt, sizeof mm.txt, "%.17g", val[k]);
                hit = &mm;
            }
            memcpy(&buf[fill], hit->txt, (size_t)hit->len); fill += (size_t)hit->len;
            buf[fill++] = '\n';
            if (fill >= ((size_t)1 << 22)) { if (fwrite(buf.data(), 1, fill, f) != fill) bad = 1; fill = 0; }
        }
    if (fill && fwrite(buf.data(), 1, fill, f) != fill) bad = 1;
    bad |= ferror(f);
    return (fclose(f) != 0 || bad) ? -2 : 0;
}

extern "C" int orc_num_threads(void) { return omp_get_max_threads(); }
extern "C" void orc_set_threads(int t) { omp_set_num_threads(t > 0 ? t : 1); }

// ===================================================================== L2-L4: solve
namespace {

struct Crs { int n; const int *rp, *ci; const double *v; };

struct State {
    // Solver<VT> members (solver.hpp:23-34)
    std::vector<double> x_star, x, x_old_s, x_new_s, tmp, D, r_s, b;
    double *x_old, *x_new, *r;
    // cgArgs (cg.hpp:16-38)
    std::vector<double> p_old_s, p_new_s, r_old_s, r_new_s;
    double *p_old, *p_new, *r_old, *r_new;
    // gmresArgs (gmres.hpp:16-42)
    double beta = 0.0; int restart_count = 0, m = 0;
    std::vector<double> V, Vy, H, H_tmp, J, R, Q, Q_copy, g, g_copy, init_v;
    // GS split (gauss_seidel.hpp:249-630)
    std::vector<int> Lp, Lc, Up, Uc; std::vector<double> Lv, Uv;
};

void identity(std::vector<double> &M, int nr, int nc)   // init_identity (utility_funcs.hpp:47-66)
{
    for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) M[(size_t)nc * i + j] = (i == j) ? 1.0 : 0.0;
}

// init_gmres_structs (gmres.hpp:726-786)
void gmres_init(State &s, int n)
{
    int m = s.m;
    s.beta = orc_nrm2(s.r, n);
    orc_scale(s.init_v.data(), s.r, 1 / s.beta, n);
    std::fill(s.V.begin(), s.V.end(), 0.0);
    std::fill(s.Vy.begin(), s.Vy.end(), 0.0);
    std::copy(s.init_v.begin(), s.init_v.end(), s.V.begin());
    std::fill(s.H.begin(), s.H.end(), 0.0);
    identity(s.R, m, m + 1);
    identity(s.Q, m + 1, m + 1);
    identity(s.Q_copy, m + 1, m + 1);
    std::fill(s.g.begin(), s.g.end(), 0.0);      s.g[0] = s.beta;
    std::fill(s.g_copy.begin(), s.g_copy.end(), 0.0); s.g_copy[0] = s.beta;
}

// gmres_iteration_ref_cpu (gmres.hpp:69-670)
void gmres_trip(const Crs &A, State &s, int precond, int iter_count, double *residual_norm)
{
    const int n = A.n, m = s.m;
    const int k = iter_count - s.restart_count * m;                          // :113
    double *w = s.tmp.data();
    orc_spmv(n, A.rp, A.ci, A.v, &s.V[(size_t)k * n], w);                    // :234
    if (precond == ORC_PRECOND_GS)                                           // :251, :59-60
        orc_spltsv(n, s.Lp.data(), s.Lc.data(), s.Lv.data(), w, s.D.data(), w);
    for (int j = 0; j <= k; ++j) {                                           // :269 MGS
        s.H[k + (size_t)j * m] = orc_dot(w, &s.V[(size_t)j * n], n);         // :286
        orc_subtract_vectors(w, w, &s.V[(size_t)j * n], n, s.H[k + (size_t)j * m]);   // :314
    }
    s.H[(size_t)(k + 1) * m + k] = orc_nrm2(w, n);                           // :359
    orc_scale(&s.V[(size_t)(k + 1) * n], w, 1.0 / s.H[(size_t)(k + 1) * m + k], n);   // :389
    // ---- least squares by dense products, as written (:427-658)
    identity(s.J, m + 1, m + 1);                                             // :432
    identity(s.H_tmp, m + 1, m);                                             // :436
    if (k == 0) s.H_tmp = s.H;                                               // :452-460
    else dense_mm(s.Q.data(), s.H.data(), s.H_tmp.data(), m + 1, m + 1, m);  // :464
    double hkk = s.H_tmp[(size_t)k * m + k], hk1 = s.H_tmp[(size_t)(k + 1) * m + k];
    // std::pow(x,2) folds to x*x; the sum of the two squares is contracted by g++ into
    // fma(hkk, hkk, hk1*hk1) (checked against the compiled reference's printed history).
    double denom = std::sqrt(FMA(hkk, hkk, hk1 * hk1));                      // :484-485
    double c = hkk / denom, sn = hk1 / denom;                                // :487-488
    s.J[(size_t)k * (m + 1) + k] = c;       s.J[(size_t)k * (m + 1) + k + 1] = sn;
    s.J[(size_t)(k + 1) * (m + 1) + k] = -1.0 * sn; s.J[(size_t)(k + 1) * (m + 1) + k + 1] = c;
    dense_mm(s.J.data(), s.Q.data(), s.Q_copy.data(), m + 1, m + 1, m + 1);  // :528
    s.Q = s.Q_copy;                                                          // :533-537
    dense_mm(s.Q.data(), s.H.data(), s.R.data(), m + 1, m + 1, m);           // :558
    std::fill(s.g_copy.begin(), s.g_copy.end(), 0.0); s.g_copy[0] = s.beta;  // :638-641
    std::fill(s.g.begin(), s.g.end(), 0.0);           s.g[0] = s.beta;
    dense_mm(s.Q.data(), s.g.data(), s.g_copy.data(), m + 1, m + 1, 1);      // :642
    s.g = s.g_copy;
    *residual_norm = std::fabs(s.g[k + 1]);                                  // :658
}

// gmres_get_x (gmres.hpp:845-978)
void gmres_get_x(State &s, int n, double *x, const double *x0, int iter_count)
{
    const int m = s.m;
    std::vector<double> y(m, 0.0);
    double diag = 0.0;
    int k = iter_count - s.restart_count * m;                                // :880
    for (int row = k; row >= 0; --row) {                                     // :901-916
        double sum = 0.0;
        for (int c = row; c < m; ++c) {
            if (c == row) diag = s.R[(size_t)row * m + c];
            else sum = FMA(s.R[(size_t)row * m + c], y[c], sum);
        }
        y[row] = (s.g[row] - sum) / diag;
    }
    // dense_MMM_t(V, y, Vy, n, m, 1) (kernels.hpp:497-517): Vy[i] = sum_k V[k*n+i]*y[k]
    #pragma omp parallel for
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int kk = 0; kk < m; ++kk) t = FMA(s.V[(size_t)kk * n + i], y[kk], t);
        s.Vy[i] = t;
    }
    for (int i = 0; i < n; ++i) x[i] = x0[i] + s.Vy[i];                      // :971-972
}

}  // namespace

extern "C" int orc_solve(int n, const int *row_ptr, const int *col, const double *val,
                         const double *b_in, const double *x0, const orc_params *p,
                         double *hist, double *x_star_out, orc_result *out)
{
    Crs A{n, row_ptr, col, val};
    State s;
    // allocate_cpu_solver_structs (solver.hpp:151-379): everything zero-initialised
    s.x_star.assign(n, 0.0); s.x.assign(n, 0.0); s.x_old_s.assign(n, 0.0); s.x_new_s.assign(n, 0.0);
    s.tmp.assign(n, 0.0); s.D.assign(n, 0.0); s.r_s.assign(n, 0.0); s.b.assign(b_in, b_in + n);
    s.x_old = s.x_old_s.data(); s.x_new = s.x_new_s.data(); s.r = s.r_s.data();
    const int solver = p->solver;
    if (solver == ORC_GAUSS_SEIDEL || (solver == ORC_GMRES && p->precond == ORC_PRECOND_GS)) {
        s.Lp.resize(n + 1); s.Up.resize(n + 1);
        orc_split_LU(n, row_ptr, col, val, s.Lp.data(), nullptr, nullptr, s.Up.data(), nullptr, nullptr);
        s.Lc.resize(s.Lp[n]); s.Lv.resize(s.Lp[n]); s.Uc.resize(s.Up[n]); s.Uv.resize(s.Up[n]);
        orc_split_LU(n, row_ptr, col, val, s.Lp.data(), s.Lc.data(), s.Lv.data(),
                     s.Up.data(), s.Uc.data(), s.Uv.data());
    }
    if (solver == ORC_GMRES) {
        int m = s.m = p->restart_len;
        s.init_v.assign(n, 0.0); s.V.assign((size_t)n * (m + 1), 0.0); s.Vy.assign(n, 0.0);
        s.H.assign((size_t)(m + 1) * m, 0.0); s.H_tmp.assign((size_t)(m + 1) * m, 0.0);
        s.J.assign((size_t)(m + 1) * (m + 1), 0.0); s.R.assign((size_t)m * (m + 1), 0.0);
        s.Q.assign((size_t)(m + 1) * (m + 1), 0.0); s.Q_copy.assign((size_t)(m + 1) * (m + 1), 0.0);
        s.g.assign(m + 1, 0.0); s.g_copy.assign(m + 1, 0.0);
    }
    if (solver == ORC_CG) {
        s.p_old_s.assign(n, 0.0); s.p_new_s.assign(n, 0.0); s.r_old_s.assign(n, 0.0); s.r_new_s.assign(n, 0.0);
        s.p_old = s.p_old_s.data(); s.p_new = s.p_new_s.data();
        s.r_old = s.r_old_s.data(); s.r_new = s.r_new_s.data();
    }
    orc_extract_diag(n, row_ptr, col, val, s.D.data());                      // main.cpp:268
    std::copy(x0, x0 + n, s.x_old);                                          // main.cpp:275-276
    // init_residual (solver.hpp:838-855): r = b - A x_old
    orc_residual(n, row_ptr, col, val, s.x_old, s.b.data(), s.r, s.tmp.data());
    // init_structs (solver.hpp:381-401)
    if (solver == ORC_GMRES) gmres_init(s, n);
    if (solver == ORC_CG) {                                                  // cg.hpp:326-352
        std::copy(s.r, s.r + n, s.p_old); std::copy(s.r, s.r + n, s.r_old);
    }
    // compute_stopping_criteria (structs.hpp:41-68): mixed norms
    out->r0_inf = orc_norm_inf(s.r, n);
    out->r0_two = orc_nrm2(s.r, n);
    const double stopping = p->tol * (solver == ORC_GMRES ? out->r0_two : out->r0_inf);
    out->stopping = stopping;

    int iter_count = 0, residual_count = 0;
    bool convergence_flag = false;
    double residual_norm = 0.0;
    const int max_iters = p->fixed_iters > 0 ? p->fixed_iters : p->max_iters;

    // record_residual_norm (utility_funcs.hpp:379-504)
    auto record = [&](const double *x_for_js) {
        if (solver == ORC_JACOBI || solver == ORC_GAUSS_SEIDEL) {
            orc_residual(n, row_ptr, col, val, x_for_js, s.b.data(), s.r, s.tmp.data());
            residual_norm = orc_nrm2(s.r, n);
        } else if (solver == ORC_CG) {
            residual_norm = orc_nrm2(s.r, n);
        }                                   // GMRES: keeps |g[k+1]| from the trip
        hist[residual_count] = residual_norm;
        if (convergence_flag) {}
        else if (iter_count == max_iters) {}
        else if (residual_norm < stopping) {}
        else ++residual_count;
    };

    auto t0 = std::chrono::steady_clock::now();
    bool go;
    do {                                                                     // solver_harness.hpp:40-116
        // ---- Solver::iterate (solver.hpp:403-550)
        if (solver == ORC_JACOBI) {
            if (p->jacobi_mode == ORC_JACOBI_SHIPPED) {
                // jacobi.hpp:191 — y and x swapped: x_old <- A * x_new
                orc_spmv(n, row_ptr, col, val, s.x_new, s.x_old);
                // jacobi_normalize_x_cpu (jacobi.hpp:41-62)
                #pragma omp parallel for
                for (int i = 0; i < n; ++i) {
                    double adj = FMA(-s.D[i], s.x_old[i], s.x_new[i]);
                    s.x_new[i] = (s.b[i] - adj) / s.D[i];
                }
            } else {
                // jacobi_iteration_ref_cpu (jacobi.hpp:65-93); diag_elem carried across rows
                // like the reference's function-scope variable (serial semantics).
                double diag_elem = 0.0;
                for (int i = 0; i < n; ++i) {
                    double sum = 0.0;
                    for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
                        if (col[k] == i) diag_elem = val[k];
                        else sum = FMA(val[k], s.x_old[col[k]], sum);
                    }
                    s.x_new[i] = (s.b[i] - sum) / diag_elem;
                }
            }
        } else if (solver == ORC_GAUSS_SEIDEL) {                             // gauss_seidel.hpp:144,177,227
            orc_spmv(n, s.Up.data(), s.Uc.data(), s.Uv.data(), s.x.data(), s.tmp.data());
            orc_subtract_vectors(s.tmp.data(), s.b.data(), s.tmp.data(), n, 1.0);
            orc_spltsv(n, s.Lp.data(), s.Lc.data(), s.Lv.data(), s.x.data(), s.D.data(), s.tmp.data());
        } else if (solver == ORC_GMRES) {
            gmres_trip(A, s, p->precond, iter_count, &residual_norm);
        } else {                                                             // cg.hpp:40-281
            orc_spmv(n, row_ptr, col, val, s.p_old, s.tmp.data());
            double num = orc_dot(s.r_old, s.r_old, n);
            double den = orc_dot(s.tmp.data(), s.p_old, n);
            double alpha = num / den;
            orc_sum_vectors(s.x_new, s.x_old, s.p_old, n, alpha);
            orc_subtract_vectors(s.r_new, s.r_old, s.tmp.data(), n, alpha);
            num = orc_dot(s.r_new, s.r_new, n);
            den = orc_dot(s.r_old, s.r_old, n);
            double beta = num / den;
            orc_sum_vectors(s.p_new, s.r_new, s.p_old, n, beta);
            std::swap(s.r, s.r_new);                                         // solver.hpp:545
        }
        // ---- record (residual_check_len == 1, main.cpp:564)
        record(solver == ORC_JACOBI ? s.x_new : s.x.data());
        // ---- exchange_arrays (solver.hpp:772-836)
        if (solver == ORC_JACOBI) std::copy(s.x_new, s.x_new + n, s.x_old);
        else if (solver == ORC_CG) {
            std::swap(s.p_old, s.p_new); std::swap(s.r_old, s.r); std::swap(s.x_old, s.x_new);
        }
        // ---- restart (solver_harness.hpp:92-104, solver.hpp:552-656)
        if (solver == ORC_GMRES && residual_norm > stopping && iter_count < max_iters &&
            (iter_count + 1) % s.m == 0) {
            gmres_get_x(s, n, s.x.data(), s.x_old, iter_count);
            orc_residual(n, row_ptr, col, val, s.x.data(), s.b.data(), s.r, s.tmp.data());
            if (p->precond == ORC_PRECOND_GS)
                orc_spltsv(n, s.Lp.data(), s.Lc.data(), s.Lv.data(), s.r, s.D.data(), s.r);
            gmres_init(s, n);
            ++s.restart_count;
            std::copy(s.x.begin(), s.x.end(), s.x_old);
        }
        ++iter_count;
        go = iter_count < max_iters && !std::isinf(residual_norm);
        if (p->fixed_iters <= 0) go = go && residual_norm > stopping;
    } while (go);
    out->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    convergence_flag = (residual_norm <= stopping) && !std::isinf(residual_norm);
    // save_x_star (solver.hpp:658-726)
    if (solver == ORC_JACOBI || solver == ORC_CG) std::copy(s.x_old, s.x_old + n, s.x_star.begin());
    else if (solver == ORC_GAUSS_SEIDEL) s.x_star = s.x;
    else gmres_get_x(s, n, s.x_star.data(), s.x_old, iter_count - 1);
    // post-loop record (solver_harness.hpp:122-149)
    record(s.x_star.data());

    out->iters = iter_count; out->residual_count = residual_count; out->converged = convergence_flag;
    out->final_residual = hist[residual_count];
    if (x_star_out) std::copy(s.x_star.begin(), s.x_star.end(), x_star_out);
    return 0;
}
