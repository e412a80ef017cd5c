// its_krylov_cr.hip — communication-reducing Krylov variants, opt-in and OUTSIDE the parity
// contract: they change the residual history (different recurrences), not the solution they converge to.
//
//   pipelined CG (Ghysels & Vanroose 2014, unpreconditioned): ONE reduction point per trip — (r,r) and (w,r) are
//       summed together — instead of the two of methods/cg.hpp:166-213, and the product q = A w does not depend on
//       it.  A trip is the SpMV plus ONE fused vector pass over x, r, w, p, s, z.
//   CGS-2 Arnoldi (classical Gram-Schmidt, twice): the k+1 dot products of a GMRES trip as one batched kernel and
//       one reduction per pass — 3 reductions per trip whatever k — instead of modified Gram-Schmidt's one per j
//       (methods/gmres.hpp:269-329).
// The reference has neither; both are there for the row-partitioned runs, where a reduction is an all-reduce.
#include "its_kernels.h"
#include <algorithm>

namespace its {

typedef double cr_v2d __attribute__((ext_vector_type(2)));

// partial sums of (r,r) and (w,r) in one pass: pair[0..NPART) and pair[NPART..2 NPART)
__global__ __launch_bounds__(ITS_BLOCK) void pcg_dots_kernel(const double *__restrict__ r, const double *__restrict__ w,
                                                              int64_t n, double *__restrict__ pair)
{
    __shared__ double s_red[4];
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x, nth = (int64_t)gridDim.x * ITS_BLOCK;
    double g = 0.0, d = 0.0;
    for (int64_t i = tid; i < n; i += nth) { const double rv = r[i]; g = fma(rv, rv, g); d = fma(w[i], rv, d); }
    const double bg = block_sum(g, s_red);
    const double bd = block_sum(d, s_red);
    store_partial(pair, bg);
    store_partial(pair + ITS_NPART, bd);
}

// The fused vector pass of a pipelined-CG trip.  gamma = sum pair_in[0..), delta = sum pair_in[NPART..) (already
// summed over the ranks); beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old)  (first trip:
// beta = 0, alpha = gamma / delta);  z = q + beta z;  s = w + beta s;  p = r + beta p;  x += alpha p;  r -= alpha s;
// w -= alpha z;  partial sums of the new (r,r), (w,r) into pair_out.  Scalars are double-buffered by trip parity
// (sc_in read by every workgroup, sc_out written by workgroup 0), like the rr slots of the plain CG kernels.
__global__ __launch_bounds__(ITS_BLOCK) void pcg_update_kernel(
    double *__restrict__ x, double *__restrict__ r, double *__restrict__ w, double *__restrict__ p, double *__restrict__ s,
    double *__restrict__ z, const double *__restrict__ q, int64_t n, const double *__restrict__ pair_in,
    double *__restrict__ pair_out, const double *__restrict__ sc_in, double *__restrict__ sc_out, int first)
{
    __shared__ double s_red[4];
    const double gamma = sum_partials(pair_in, s_red);
    const double delta = sum_partials(pair_in + ITS_NPART, s_red);
    double beta = 0.0, alpha = gamma / delta;
    if (!first) {
        beta = gamma / sc_in[0];
        alpha = gamma / (delta - beta * gamma / sc_in[1]);
    }
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x, nth = (int64_t)gridDim.x * ITS_BLOCK;
    double g = 0.0, d = 0.0;
    const int64_t n2 = n / 2;                       // solver vectors are 256-byte aligned
#define PCG_ELEM(X, R, W, P, S, Z, Q)                 \
    {                                                 \
        Z = fma(beta, Z, Q);                          \
        S = fma(beta, S, W);                          \
        P = fma(beta, P, R);                          \
        X = fma(alpha, P, X);                         \
        R = fma(-alpha, S, R);                        \
        W = fma(-alpha, Z, W);                        \
        g = fma(R, R, g); d = fma(W, R, d);           \
    }
    for (int64_t i = tid; i < n2; i += nth) {
        cr_v2d xv = reinterpret_cast<cr_v2d *>(x)[i], rv = reinterpret_cast<cr_v2d *>(r)[i], wv = reinterpret_cast<cr_v2d *>(w)[i];
        cr_v2d pv = reinterpret_cast<cr_v2d *>(p)[i], sv = reinterpret_cast<cr_v2d *>(s)[i], zv = reinterpret_cast<cr_v2d *>(z)[i];
        const cr_v2d qv = __builtin_nontemporal_load(reinterpret_cast<const cr_v2d *>(q) + i);
        PCG_ELEM(xv.x, rv.x, wv.x, pv.x, sv.x, zv.x, qv.x)
        PCG_ELEM(xv.y, rv.y, wv.y, pv.y, sv.y, zv.y, qv.y)
        reinterpret_cast<cr_v2d *>(x)[i] = xv; reinterpret_cast<cr_v2d *>(r)[i] = rv; reinterpret_cast<cr_v2d *>(w)[i] = wv;
        reinterpret_cast<cr_v2d *>(p)[i] = pv; reinterpret_cast<cr_v2d *>(s)[i] = sv; reinterpret_cast<cr_v2d *>(z)[i] = zv;
    }
    for (int64_t i = 2 * n2 + tid; i < n; i += nth) {
        double xv = x[i], rv = r[i], wv = w[i], pv = p[i], sv = s[i], zv = z[i];
        const double qv = q[i];
        PCG_ELEM(xv, rv, wv, pv, sv, zv, qv)
        x[i] = xv; r[i] = rv; w[i] = wv; p[i] = pv; s[i] = sv; z[i] = zv;
    }
#undef PCG_ELEM
    const double bg = block_sum(g, s_red);
    const double bd = block_sum(d, s_red);
    store_partial(pair_out, bg);
    store_partial(pair_out + ITS_NPART, bd);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc_out[0] = gamma; sc_out[1] = alpha; sc_out[2] = beta; }
}

static inline dim3 cr_grid(int64_t n)
{
    int64_t g = (n + 4 * ITS_BLOCK - 1) / (4 * ITS_BLOCK);
    return dim3((unsigned)std::min<int64_t>(std::max<int64_t>(g, 8), ITS_NPART));
}

void launch_pcg_dots(const double *r, const double *w, int64_t n, double *pair, hipStream_t s)
{ hipLaunchKernelGGL(pcg_dots_kernel, cr_grid(n), dim3(ITS_BLOCK), 0, s, r, w, n, pair); }

void launch_pcg_update(double *x, double *r, double *w, double *p, double *sv, double *z, const double *q, int64_t n,
                       const double *pair_in, double *pair_out, const double *sc_in, double *sc_out, int first, hipStream_t s)
{ hipLaunchKernelGGL(pcg_update_kernel, cr_grid(n), dim3(ITS_BLOCK), 0, s, x, r, w, p, sv, z, q, n, pair_in, pair_out, sc_in, sc_out, first); }

// ---------------------------------------------------------------------------------------
// CGS-2: batched dot products w . V_j, j < nv, and the batched update w -= sum_j h_j V_j
// ---------------------------------------------------------------------------------------
constexpr int GRAM_BLOCKS = 512;       // workgroups of the batched-dot kernel = partials per j
constexpr int GRAM_J = 8;              // basis vectors per pass over w inside the kernel

// part[j * GRAM_BLOCKS + block] = this block's share of w . V_j.  Two elements per lane (w and the basis vectors are
// 256-byte aligned, ld is a multiple of 32), GRAM_J basis vectors per pass over w: 8 independent 16-byte loads in
// flight per lane.
__global__ __launch_bounds__(ITS_BLOCK) void gram_dots_kernel(const double *__restrict__ w, const double *__restrict__ V,
                                                              int64_t ld, int nv, int64_t n, double *__restrict__ part)
{
    __shared__ double s_red[4];
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x, nth = (int64_t)gridDim.x * ITS_BLOCK;
    const int64_t n2 = n / 2;
    const cr_v2d *w2 = reinterpret_cast<const cr_v2d *>(w);
    for (int j0 = 0; j0 < nv; j0 += GRAM_J) {
        double acc[GRAM_J];
#pragma unroll
        for (int u = 0; u < GRAM_J; ++u) acc[u] = 0.0;
        for (int64_t i = tid; i < n2; i += nth) {
            const cr_v2d wv = w2[i];
            cr_v2d v[GRAM_J];
#pragma unroll
            for (int u = 0; u < GRAM_J; ++u) {
                const int j = j0 + u < nv ? j0 + u : j0;             // clamped: the load is unconditional
                v[u] = reinterpret_cast<const cr_v2d *>(V + (int64_t)j * ld)[i];
            }
#pragma unroll
            for (int u = 0; u < GRAM_J; ++u) acc[u] = fma(wv.y, v[u].y, fma(wv.x, v[u].x, acc[u]));
        }
        for (int64_t i = 2 * n2 + tid; i < n; i += nth) {
            const double wv = w[i];
#pragma unroll
            for (int u = 0; u < GRAM_J; ++u)
                if (j0 + u < nv) acc[u] = fma(wv, V[(int64_t)(j0 + u) * ld + i], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < GRAM_J; ++u) {
            const double bs = block_sum(acc[u], s_red);
            if (threadIdx.x == 0 && j0 + u < nv) part[(int64_t)(j0 + u) * gridDim.x + blockIdx.x] = bs;
        }
    }
}

// h[j] = fixed-order sum of the nb partials of j (one workgroup per j)
__global__ __launch_bounds__(ITS_BLOCK) void gram_finalize_kernel(const double *__restrict__ part, int nb, double *__restrict__ h)
{
    __shared__ double s_red[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += ITS_BLOCK) s += part[(int64_t)blockIdx.x * nb + b];
    const double t = block_sum(s, s_red);
    if (threadIdx.x == 0) h[blockIdx.x] = t;
}

// w -= sum_j h[j] V_j; optionally the partial sums of the new w . w.  Same access shape as gram_dots_kernel.
__global__ __launch_bounds__(ITS_BLOCK) void gram_axpy_kernel(double *__restrict__ w, const double *__restrict__ V, int64_t ld,
                                                              int nv, const double *__restrict__ h, int64_t n,
                                                              double *__restrict__ part_ww)
{
    extern __shared__ double s_h[];
    __shared__ double s_red[4];
    for (int j = threadIdx.x; j < nv + GRAM_J; j += ITS_BLOCK) s_h[j] = j < nv ? h[j] : 0.0;
    __syncthreads();
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x, nth = (int64_t)gridDim.x * ITS_BLOCK;
    const int64_t n2 = n / 2;
    double ww = 0.0;
    for (int64_t i = tid; i < n2; i += nth) {
        cr_v2d t; t.x = 0.0; t.y = 0.0;
        for (int j0 = 0; j0 < nv; j0 += GRAM_J) {
            cr_v2d v[GRAM_J];
#pragma unroll
            for (int u = 0; u < GRAM_J; ++u) {
                const int j = j0 + u < nv ? j0 + u : j0;             // clamped load, coefficient 0 beyond nv
                v[u] = reinterpret_cast<const cr_v2d *>(V + (int64_t)j * ld)[i];
            }
#pragma unroll
            for (int u = 0; u < GRAM_J; ++u) { t.x = fma(s_h[j0 + u], v[u].x, t.x); t.y = fma(s_h[j0 + u], v[u].y, t.y); }
        }
        cr_v2d wv = reinterpret_cast<cr_v2d *>(w)[i];
        wv.x -= t.x; wv.y -= t.y;
        reinterpret_cast<cr_v2d *>(w)[i] = wv;
        ww = fma(wv.y, wv.y, fma(wv.x, wv.x, ww));
    }
    for (int64_t i = 2 * n2 + tid; i < n; i += nth) {
        double t = 0.0;
        for (int j = 0; j < nv; ++j) t = fma(s_h[j], V[(int64_t)j * ld + i], t);
        const double wv = w[i] - t;
        w[i] = wv;
        ww = fma(wv, wv, ww);
    }
    if (part_ww) {
        const double bs = block_sum(ww, s_red);
        store_partial(part_ww, bs);
    }
}

// hcol[j] = h1[j] + h2[j]: the Hessenberg column of a CGS-2 trip (projection + its correction)
__global__ void gram_combine_kernel(const double *__restrict__ h1, const double *__restrict__ h2, int nv, double *__restrict__ hcol)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nv) hcol[j] = h1[j] + h2[j];
}

int gram_blocks() { return GRAM_BLOCKS; }

void launch_gram_dots(const double *w, const double *V, int64_t ld, int nv, int64_t n, double *part, hipStream_t s)
{ hipLaunchKernelGGL(gram_dots_kernel, dim3(GRAM_BLOCKS), dim3(ITS_BLOCK), 0, s, w, V, ld, nv, n, part); }
void launch_gram_finalize(const double *part, int nv, double *h, hipStream_t s)
{ hipLaunchKernelGGL(gram_finalize_kernel, dim3(nv), dim3(ITS_BLOCK), 0, s, part, GRAM_BLOCKS, h); }
void launch_gram_axpy(double *w, const double *V, int64_t ld, int nv, const double *h, int64_t n, double *part_ww, hipStream_t s)
{ hipLaunchKernelGGL(gram_axpy_kernel, cr_grid(n), dim3(ITS_BLOCK), (size_t)(nv + GRAM_J) * sizeof(double), s, w, V, ld, nv, h, n, part_ww); }
void launch_gram_combine(const double *h1, const double *h2, int nv, double *hcol, hipStream_t s)
{ hipLaunchKernelGGL(gram_combine_kernel, dim3((nv + 255) / 256), dim3(256), 0, s, h1, h2, nv, hcol); }

}  // namespace its
