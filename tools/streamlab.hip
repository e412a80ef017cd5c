// streamlab — where does the exact SpMV's time go?  Standalone microbenchmark (not part of the
// library): rebuilds the stream kernel stage by stage on the 256^3 7-pt Laplacian and times each
// stage, next to a plain grid-stride read of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/streamlab tools/streamlab.hip
//   ./tools/streamlab [N=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BLK = 256;
constexpr int CAP = 1536;

// ---------------------------------------------------------------- plain read, grid-stride
template <int UNR>
__global__ __launch_bounds__(BLK) void k_read(const int4 *__restrict__ col4, const double2 *__restrict__ val2, int64_t n4, double *out)
{
    const int64_t stride = (int64_t)gridDim.x * BLK;
    double acc = 0.0; int ia = 0;
    int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
        int4 c[UNR]; double2 v0[UNR], v1[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { c[u] = col4[i + u * stride]; v0[u] = val2[2 * (i + u * stride)]; v1[u] = val2[2 * (i + u * stride) + 1]; }
#pragma unroll
        for (int u = 0; u < UNR; ++u) { ia += c[u].x ^ c[u].y ^ c[u].z ^ c[u].w; acc += v0[u].x + v0[u].y + v1[u].x + v1[u].y; }
    }
    for (; i < n4; i += stride) { int4 c = col4[i]; double2 a = val2[2 * i], b = val2[2 * i + 1]; ia += c.x ^ c.y ^ c.z ^ c.w; acc += a.x + a.y + b.x + b.y; }
    if (acc == 1.2345e-300 || ia == 0x7fffffff) out[threadIdx.x] = acc;
}

// like k_read<1>, but the two 16-byte val loads of a wave cover two disjoint contiguous KB instead of
// interleaving inside the same 2 KB (no second request to a cache line that is still in flight)
__global__ __launch_bounds__(BLK) void k_read_split(const int4 *__restrict__ col4, const double2 *__restrict__ val2, int64_t n4, double *out)
{
    const int64_t stride = (int64_t)gridDim.x * BLK;
    const int lane = threadIdx.x & 63;
    double acc = 0.0; int ia = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n4; i += stride) {
        const int64_t w0 = 2 * (i - lane);
        int4 c = col4[i]; double2 a = val2[w0 + lane], b = val2[w0 + 64 + lane];
        ia += c.x ^ c.y ^ c.z ^ c.w; acc += a.x + a.y + b.x + b.y;
    }
    if (acc == 1.2345e-300 || ia == 0x7fffffff) out[threadIdx.x] = acc;
}

// grid-stride read with a 4-byte coalesced write per 48 bytes read.  MODE: 0 plain store, 1 nontemporal,
// 2 plain store into a 1 MB ring (stays in L2), 4 system-scope store, 5 one float4 per 4 trips
template <int MODE>
__global__ __launch_bounds__(BLK) void k_read_w(const int4 *__restrict__ col4, const double2 *__restrict__ val2, int64_t n4, float *__restrict__ w)
{
    const int64_t stride = (int64_t)gridDim.x * BLK;
    const int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
    float q[4]; int k = 0; int64_t it = 0;
    for (int64_t i = g; i < n4; i += stride, ++it) {
        int4 c = col4[i]; double2 a = val2[2 * i], b = val2[2 * i + 1];
        const float v = (float)(a.x + a.y + b.x + b.y) + (float)(c.x ^ c.y ^ c.z ^ c.w);
        if (MODE == 6) { const float old = w[i]; w[i] = v + (old == 1.25f ? 1.0f : 0.0f); }   // read-modify-write of the output line
        else if (MODE == 0) w[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, w + i);
        else if (MODE == 2) w[i & 0x3ffff] = v;
        else if (MODE == 4) __hip_atomic_store(w + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (MODE == 5) {
            q[k++] = v;
            if (k == 4) { reinterpret_cast<float4 *>(w)[(it >> 2) * stride + g] = make_float4(q[0], q[1], q[2], q[3]); k = 0; }
        }
    }
    if (MODE == 5 && k) w[n4 - 1 - g] = q[0];
}

// plain read where each block owns contiguous chunks (persistent, XCD map like the library)
__global__ __launch_bounds__(BLK) void k_read_chunks(const int4 *__restrict__ col4, const double2 *__restrict__ val2, int64_t n4, int chunk4, double *out)
{
    const int G = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int64_t nchunks = (n4 + chunk4 - 1) / chunk4;
    const int64_t X0 = xcd * nchunks / 8, X1 = (xcd + 1) * nchunks / 8;
    double acc = 0.0; int ia = 0;
    for (int64_t c = X0 + jx; c < X1; c += G >> 3) {
        const int64_t base = c * chunk4;
        for (int o = threadIdx.x; o < chunk4; o += BLK) {
            const int64_t i = base + o;
            if (i < n4) { int4 cc = col4[i]; double2 a = val2[2 * i], b = val2[2 * i + 1]; ia += cc.x ^ cc.y ^ cc.z ^ cc.w; acc += a.x + a.y + b.x + b.y; }
        }
    }
    if (acc == 1.2345e-300 || ia == 0x7fffffff) out[threadIdx.x] = acc;
}

// ---------------------------------------------------------------- the stream kernel in stages
// STAGE 0: prefetch loads only, summed in registers (no LDS, no barrier)
//       1: + commit to LDS + one barrier per row block
//       2: + per-lane walk of the row out of LDS (sum of val, no gather)
//       3: + x gathers and FMA
//       4: + y store                                   (== library kernel, EPI_STORE, no dot)
template <int STAGE, int CAPT>
__global__ __launch_bounds__(BLK) void k_stream(int nb, const int4 *__restrict__ desc, const int *__restrict__ rp,
                                                 const int *__restrict__ col, const double *__restrict__ val,
                                                 const double *x, double *__restrict__ y, double *out)
{
    __shared__ __attribute__((aligned(16))) double s_val[2][CAPT + 8];
    __shared__ __attribute__((aligned(16))) int s_col[2][CAPT + 8];
    __shared__ int s_rp[2][BLK + 4];
    const int t = threadIdx.x, G = gridDim.x;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int X0 = (int)(((int64_t)xcd * nb) / 8), X1 = (int)(((int64_t)(xcd + 1) * nb) / 8);
    const int rb0 = X0 + jx, rb1 = X1, rstep = G >> 3;
    const int niter = rb0 < rb1 ? (rb1 - rb0 + rstep - 1) / rstep : 0;
    const int rb_last = rb0 + (niter - 1) * rstep;
    int4 rcA, rcB; double2 rvA0, rvA1, rvB0, rvB1; int rrp = 0;
    rcA = rcB = make_int4(0, 0, 0, 0); rvA0 = rvA1 = rvB0 = rvB1 = make_double2(0, 0);
    double sink = 0.0; int isink = 0;
#define ISSUE(d) do { const int base_ = (d).z & ~3; const int iA_ = base_ + 4 * t, iB_ = iA_ + 4 * BLK; \
        if (iA_ < (d).w) { rcA = *reinterpret_cast<const int4 *>(col + iA_); rvA0 = *reinterpret_cast<const double2 *>(val + iA_); rvA1 = *reinterpret_cast<const double2 *>(val + iA_ + 2); } \
        if (iB_ < (d).w) { rcB = *reinterpret_cast<const int4 *>(col + iB_); rvB0 = *reinterpret_cast<const double2 *>(val + iB_); rvB1 = *reinterpret_cast<const double2 *>(val + iB_ + 2); } \
        if (t < (d).y) rrp = rp[(d).x + t]; } while (0)
#define COMMIT(d, buf) do { const int base_ = (d).z & ~3; const int oA_ = 4 * t, oB_ = oA_ + 4 * BLK; \
        if (base_ + oA_ < (d).w) { *reinterpret_cast<int4 *>(&s_col[buf][oA_]) = rcA; *reinterpret_cast<double2 *>(&s_val[buf][oA_]) = rvA0; *reinterpret_cast<double2 *>(&s_val[buf][oA_ + 2]) = rvA1; } \
        if (base_ + oB_ < (d).w) { *reinterpret_cast<int4 *>(&s_col[buf][oB_]) = rcB; *reinterpret_cast<double2 *>(&s_val[buf][oB_]) = rvB0; *reinterpret_cast<double2 *>(&s_val[buf][oB_ + 2]) = rvB1; } \
        if (t < (d).y) s_rp[buf][t] = rrp; if (t == 0) s_rp[buf][(d).y] = (d).w; } while (0)
    int cur = 0; bool have = false;
    double pend = 0.0; int pend_row = -1;
    int4 d = make_int4(0, 0, 0, 0), dn = d;
    if (niter > 0) { d = desc[rb0]; dn = desc[min(rb0 + rstep, rb_last)]; }
    for (int it = 0; it < niter; ++it) {
        const int rb = rb0 + it * rstep;
        const int4 dnn = desc[min(rb + 2 * rstep, rb_last)];
        const int nr = d.y, row = d.x + t;
        if (!have) ISSUE(d);
        if (STAGE >= 1) { COMMIT(d, cur); __syncthreads(); }
        else { isink += rcA.x ^ rcB.y ^ rrp; sink += rvA0.x + rvA1.y + rvB0.x + rvB1.y; }
        have = false;
        if (STAGE == 5 && pend_row >= 0) { y[pend_row] = pend; pend_row = -1; }   // last trip's result goes out BEFORE the prefetch
        if (it + 1 < niter) { ISSUE(dn); have = true; }
        if (STAGE >= 2 && t < nr) {
            const int base = d.z & ~3;
            const int j0 = s_rp[cur][t] - base, j1 = s_rp[cur][t + 1] - base;
            double acc = 0.0;
            for (int j = j0; j < j1; j += 8) {
                int c[8]; double v[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int jj = (j + u < j1) ? j + u : j; c[u] = s_col[cur][jj]; v[u] = s_val[cur][jj]; }
                if (STAGE == 11 || STAGE == 12 || STAGE == 13) {
                    // probe: how much do gathers into cache lines that are still in flight cost?  11 drops the
                    // x-1 / x+1 gathers (slots 2 and 4 of an interior 7-pt row), 12 also the padding slot 7,
                    // 13 keeps only the two far (z-1, z+1) gathers.  Timing only (wrong results).
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool drop = STAGE == 13 ? (u != 0 && u != 6) : ((u == 2 || u == 4) || (STAGE == 12 && u == 7));
                        xv[u] = drop ? (double)c[u] : x[c[u]];
                    }
                } else if (STAGE >= 3 && STAGE != 6) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = x[c[u]];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = (double)c[u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j + u < j1) acc = fma(v[u], xv[u], acc);
            }
            if (STAGE == 5) { pend = acc; pend_row = row; }
            else if (STAGE == 8) const_cast<double *>(x)[row] = acc * 0.05;
            else if (STAGE == 9) y[row & 0xffff] = acc;
            else if (STAGE == 14) { if ((it & 1) == 0) y[row] = acc; else sink += acc; }      // half the store instructions
            else if (STAGE == 15) { if ((t & 1) == 0) y[row] = acc; else sink += acc; }       // same instructions and lines, half the bytes
            else if (STAGE == 16) { if ((it & 3) == 0) y[row] = acc; else sink += acc; }      // a quarter of the store instructions
            else if (STAGE >= 4 && STAGE < 11) y[row] = acc; else sink += acc;
        }
        cur ^= 1; d = dn; dn = dnn;
    }
    if (STAGE == 5 && pend_row >= 0) y[pend_row] = pend;
    if (sink == 1.2345e-300 || isink == 0x7fffffff) out[t] = sink;
}

// burst-store variant: each workgroup takes KB consecutive row blocks per turn, parks the results in LDS
// and writes them as one contiguous burst (KB * ~1.75 KB) every KB trips
template <int KB>
__global__ __launch_bounds__(BLK) void k_stream_burst(int nb, const int4 *__restrict__ desc, const int *__restrict__ rp,
                                                      const int *__restrict__ col, const double *__restrict__ val,
                                                      const double *x, double *__restrict__ y, double *out)
{
    __shared__ __attribute__((aligned(16))) double s_val[2][CAP + 8];
    __shared__ __attribute__((aligned(16))) int s_col[2][CAP + 8];
    __shared__ int s_rp[2][BLK + 4];
    __shared__ double s_y[KB * BLK];
    const int t = threadIdx.x, G = gridDim.x;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int X0 = (int)(((int64_t)xcd * nb) / 8), X1 = (int)(((int64_t)(xcd + 1) * nb) / 8);
    const int rstep = G >> 3;
#define RBX(i) (X0 + ((i) / KB) * (rstep * KB) + jx * KB + ((i) % KB))
    int niter = 0;
    while (RBX(niter) < X1) ++niter;
    int4 rcA, rcB; double2 rvA0, rvA1, rvB0, rvB1; int rrp = 0;
    rcA = rcB = make_int4(0, 0, 0, 0); rvA0 = rvA1 = rvB0 = rvB1 = make_double2(0, 0);
    int cur = 0; bool have = false;
    int gstart = 0;
    int4 d = make_int4(0, 0, 0, 0), dn = d;
    if (niter > 0) { d = desc[RBX(0)]; dn = desc[RBX(niter > 1 ? 1 : 0)]; }
    for (int it = 0; it < niter; ++it) {
        const int4 dnn = desc[RBX(min(it + 2, niter - 1))];
        const int nr = d.y, row = d.x + t;
        if (it % KB == 0) gstart = d.x;
        if (!have) ISSUE(d);
        COMMIT(d, cur); __syncthreads();
        have = false;
        if (it + 1 < niter) { ISSUE(dn); have = true; }
        if (t < nr) {
            const int base = d.z & ~3;
            const int j0 = s_rp[cur][t] - base, j1 = s_rp[cur][t + 1] - base;
            double acc = 0.0;
            for (int j = j0; j < j1; j += 8) {
                int c[8]; double v[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int jj = (j + u < j1) ? j + u : j; c[u] = s_col[cur][jj]; v[u] = s_val[cur][jj]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = x[c[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j + u < j1) acc = fma(v[u], xv[u], acc);
            }
            s_y[row - gstart] = acc;
        }
        if (it % KB == KB - 1 || it == niter - 1) {
            const int gcount = d.x + nr - gstart;
            __syncthreads();
            for (int i = t; i < gcount; i += BLK) y[gstart + i] = s_y[i];
        }
        cur ^= 1; d = dn; dn = dnn;
    }
#undef RBX
}

// ---------------------------------------------------------------- wave-specialised: a fifth wave does the y stores
// (stores and loads share the in-order vmcnt queue on gfx9: a walker wave that also stores must
// wait for the store's acknowledge before it sees its next prefetch)
template <int CAPT, int GATHER>
__global__ __launch_bounds__(320) void k_stream_ws(int nb, const int4 *__restrict__ desc, const int *__restrict__ rp,
                                                   const int *__restrict__ col, const double *__restrict__ val,
                                                   const double *x, double *__restrict__ y, double *out)
{
    __shared__ __attribute__((aligned(16))) double s_val[2][CAPT + 8];
    __shared__ __attribute__((aligned(16))) int s_col[2][CAPT + 8];
    __shared__ int s_rp[2][BLK + 4];
    __shared__ double s_y[2][BLK];
    const int t = threadIdx.x, G = gridDim.x;
    const bool writer = t >= BLK;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int X0 = (int)(((int64_t)xcd * nb) / 8), X1 = (int)(((int64_t)(xcd + 1) * nb) / 8);
    const int rb0 = X0 + jx, rb1 = X1, rstep = G >> 3;
    const int niter = rb0 < rb1 ? (rb1 - rb0 + rstep - 1) / rstep : 0;
    const int rb_last = rb0 + (niter - 1) * rstep;
    int4 rcA, rcB; double2 rvA0, rvA1, rvB0, rvB1; int rrp = 0;
    rcA = rcB = make_int4(0, 0, 0, 0); rvA0 = rvA1 = rvB0 = rvB1 = make_double2(0, 0);
    int cur = 0; bool have = false;
    int4 d = make_int4(0, 0, 0, 0), dn = d, dp = d;
    if (niter > 0) { d = desc[rb0]; dn = desc[min(rb0 + rstep, rb_last)]; }
    for (int it = 0; it < niter; ++it) {
        const int rb = rb0 + it * rstep;
        const int4 dnn = desc[min(rb + 2 * rstep, rb_last)];
        const int nr = d.y;
        if (!writer) { if (!have) ISSUE(d); COMMIT(d, cur); }
        __syncthreads();
        if (writer) {
            if (it > 0) for (int k = t - BLK; k < dp.y; k += 64) y[dp.x + k] = s_y[cur ^ 1][k];
        } else {
            have = false;
            if (it + 1 < niter) { ISSUE(dn); have = true; }
            if (t < nr) {
                const int base = d.z & ~3;
                const int j0 = s_rp[cur][t] - base, j1 = s_rp[cur][t + 1] - base;
                double acc = 0.0;
                for (int j = j0; j < j1; j += 8) {
                    int c[8]; double v[8], xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int jj = (j + u < j1) ? j + u : j; c[u] = s_col[cur][jj]; v[u] = s_val[cur][jj]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = GATHER ? x[c[u]] : (double)c[u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (j + u < j1) acc = fma(v[u], xv[u], acc);
                }
                s_y[cur][t] = acc;
            }
        }
        cur ^= 1; dp = d; d = dn; dn = dnn;
    }
    __syncthreads();
    if (writer && niter > 0) for (int k = t - BLK; k < dp.y; k += 64) y[dp.x + k] = s_y[cur ^ 1][k];
}

// ---------------------------------------------------------------- host
static float time_ms(hipStream_t s, int reps, auto &&launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    const int maxrows = argc > 2 ? atoi(argv[2]) : BLK;
    const int64_t n = (int64_t)N * N * N;
    std::vector<int> rp(n + 1); std::vector<int> col; std::vector<double> val;
    col.reserve(7 * n); val.reserve(7 * n);
    rp[0] = 0;
    for (int z = 0; z < N; ++z) for (int yy = 0; yy < N; ++yy) for (int xx = 0; xx < N; ++xx) {
        const int64_t r = ((int64_t)z * N + yy) * N + xx;
        if (z > 0) { col.push_back((int)(r - (int64_t)N * N)); val.push_back(-1.0); }
        if (yy > 0) { col.push_back((int)(r - N)); val.push_back(-1.0); }
        if (xx > 0) { col.push_back((int)(r - 1)); val.push_back(-1.0); }
        col.push_back((int)r); val.push_back(6.0);
        if (xx < N - 1) { col.push_back((int)(r + 1)); val.push_back(-1.0); }
        if (yy < N - 1) { col.push_back((int)(r + N)); val.push_back(-1.0); }
        if (z < N - 1) { col.push_back((int)(r + (int64_t)N * N)); val.push_back(-1.0); }
        rp[r + 1] = (int)col.size();
    }
    const int64_t nnz = col.size();
    std::vector<int4> desc;
    for (int64_t r = 0; r < n;) {
        int64_t e = r;
        while (e < n && e - r < maxrows && rp[e + 1] - rp[r] <= CAP) ++e;
        desc.push_back(make_int4((int)r, (int)(e - r), rp[r], rp[e]));
        r = e;
    }
    const int nb = (int)desc.size();
    printf("N=%d n=%lld nnz=%lld row blocks=%d\n", N, (long long)n, (long long)nnz, nb);
    int *d_rp, *d_col; double *d_val, *d_x, *d_y, *d_out; int4 *d_desc;
    CK(hipMalloc(&d_rp, (n + 1) * 4)); CK(hipMalloc(&d_col, (nnz + 4096) * 4)); CK(hipMalloc(&d_val, (nnz + 4096) * 8));
    CK(hipMalloc(&d_x, n * 8)); CK(hipMalloc(&d_y, n * 8)); CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_desc, nb * sizeof(int4)));
    CK(hipMemset(d_col, 0, (nnz + 4096) * 4)); CK(hipMemset(d_val, 0, (nnz + 4096) * 8));
    CK(hipMemcpy(d_rp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, val.data(), nnz * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_desc, desc.data(), nb * sizeof(int4), hipMemcpyHostToDevice));
    std::vector<double> hx(n, 1.0); CK(hipMemcpy(d_x, hx.data(), n * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int64_t n4 = nnz / 4;
    const double stream_bytes = 12.0 * nnz, full_bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    const int reps = 30;
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        float ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_read<1>, dim3(g), dim3(BLK), 0, s, (const int4 *)d_col, (const double2 *)d_val, n4, d_out); });
        float ms2 = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_read<2>, dim3(g), dim3(BLK), 0, s, (const int4 *)d_col, (const double2 *)d_val, n4, d_out); });
        float ms4 = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_read<4>, dim3(g), dim3(BLK), 0, s, (const int4 *)d_col, (const double2 *)d_val, n4, d_out); });
        printf("k_read grid %5d: unr1 %.4f ms (%.0f GB/s)  unr2 %.4f (%.0f)  unr4 %.4f (%.0f)\n", g, ms, stream_bytes / ms / 1e6, ms2, stream_bytes / ms2 / 1e6, ms4, stream_bytes / ms4 / 1e6);
    }
    for (int g : {1024, 2048, 16384}) {
        float ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_read_split, dim3(g), dim3(BLK), 0, s, (const int4 *)d_col, (const double2 *)d_val, n4 & ~63LL, d_out); });
        printf("k_read_split grid %5d: %.4f ms (%.0f GB/s)\n", g, ms, stream_bytes / ms / 1e6);
    }
    for (int g : {1024, 2048}) for (int ch : {384, 1024, 4096}) {
        float ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_read_chunks, dim3(g), dim3(BLK), 0, s, (const int4 *)d_col, (const double2 *)d_val, n4, ch, d_out); });
        printf("k_read_chunks grid %5d chunk %5d int4: %.4f ms (%.0f GB/s)\n", g, ch, ms, stream_bytes / ms / 1e6);
    }
#define RW(M) for (int g : {1024, 16384}) { \
        float ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_read_w<M>, dim3(g), dim3(BLK), 0, s, (const int4 *)d_col, (const double2 *)d_val, n4, (float *)d_y); }); \
        printf("k_read_w mode %d grid %5d: %.4f ms (read %.0f GB/s, r+w %.0f GB/s)\n", M, g, ms, stream_bytes / ms / 1e6, (stream_bytes + 4.0 * n4) / ms / 1e6); }
    RW(0) RW(6) RW(1)
    { float ms = time_ms(s, reps, [&] { CK(hipMemsetAsync(d_y, 0, n * 8, s)); }); printf("memset %lld MB: %.4f ms (%.0f GB/s)\n", (long long)(n * 8 >> 20), ms, n * 8.0 / ms / 1e6); }
#define RUN(ST) do { float ms = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream<ST, CAP>), dim3(1024), dim3(BLK), 0, s, nb, d_desc, d_rp, d_col, d_val, d_x, d_y, d_out); }); \
        printf("k_stream stage %d: %.4f ms (stream %.0f GB/s, full %.0f GB/s)\n", ST, ms, stream_bytes / ms / 1e6, full_bytes / ms / 1e6); } while (0)
    RUN(3); RUN(4); RUN(14); RUN(15); RUN(16);
#define RUNB(K, GRID) do { float ms = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_burst<K>), dim3(GRID), dim3(BLK), 0, s, nb, d_desc, d_rp, d_col, d_val, d_x, d_y, d_out); }); \
        printf("k_stream_burst K=%d grid %d: %.4f ms (full %.0f GB/s)\n", K, GRID, ms, full_bytes / ms / 1e6); } while (0)
    RUNB(1, 1024); RUNB(2, 1024); RUNB(4, 768); RUNB(4, 1024); RUNB(8, 768); RUNB(16, 512); RUNB(8, 512);
    {
        constexpr int CAPW = 1280;
        std::vector<int4> dw;
        for (int64_t r = 0; r < n;) { int64_t e = r; while (e < n && e - r < maxrows && rp[e + 1] - rp[r] <= CAPW) ++e; dw.push_back(make_int4((int)r, (int)(e - r), rp[r], rp[e])); r = e; }
        int4 *d_dw; CK(hipMalloc(&d_dw, dw.size() * sizeof(int4))); CK(hipMemcpy(d_dw, dw.data(), dw.size() * sizeof(int4), hipMemcpyHostToDevice));
        const int nbw = (int)dw.size();
        CK(hipMemsetAsync(d_y, 0, n * 8, s));
        float ms0 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_ws<CAPW, 0>), dim3(1024), dim3(320), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); });
        float ms1 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_ws<CAPW, 1>), dim3(1024), dim3(320), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); });
        printf("grid 1024 k_stream_ws (writer wave, cap %d, %d row blocks): no-gather %.4f ms   full %.4f ms (full %.0f GB/s)\n", CAPW, nbw, ms0, ms1, full_bytes / ms1 / 1e6);
    }
    {
        constexpr int CAPW = 1024;
        std::vector<int4> dw;
        for (int64_t r = 0; r < n;) { int64_t e = r; while (e < n && e - r < maxrows && rp[e + 1] - rp[r] <= CAPW) ++e; dw.push_back(make_int4((int)r, (int)(e - r), rp[r], rp[e])); r = e; }
        int4 *d_dw; CK(hipMalloc(&d_dw, dw.size() * sizeof(int4))); CK(hipMemcpy(d_dw, dw.data(), dw.size() * sizeof(int4), hipMemcpyHostToDevice));
        const int nbw = (int)dw.size();
        CK(hipMemsetAsync(d_y, 0, n * 8, s));
        float ms0 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_ws<CAPW, 0>), dim3(1024), dim3(320), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); });
        float ms1 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_ws<CAPW, 1>), dim3(1024), dim3(320), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); });
        printf("grid 1024 k_stream_ws (writer wave, cap %d, %d row blocks): no-gather %.4f ms   full %.4f ms (full %.0f GB/s)\n", CAPW, nbw, ms0, ms1, full_bytes / ms1 / 1e6);
    }
    {
        constexpr int CAPW = 1024;
        std::vector<int4> dw;
        for (int64_t r = 0; r < n;) { int64_t e = r; while (e < n && e - r < maxrows && rp[e + 1] - rp[r] <= CAPW) ++e; dw.push_back(make_int4((int)r, (int)(e - r), rp[r], rp[e])); r = e; }
        int4 *d_dw; CK(hipMalloc(&d_dw, dw.size() * sizeof(int4))); CK(hipMemcpy(d_dw, dw.data(), dw.size() * sizeof(int4), hipMemcpyHostToDevice));
        const int nbw = (int)dw.size();
        CK(hipMemsetAsync(d_y, 0, n * 8, s));
        float ms0 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_ws<CAPW, 0>), dim3(1280), dim3(320), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); });
        float ms1 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_ws<CAPW, 1>), dim3(1280), dim3(320), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); });
        printf("grid 1280 k_stream_ws (writer wave, cap %d, %d row blocks): no-gather %.4f ms   full %.4f ms (full %.0f GB/s)\n", CAPW, nbw, ms0, ms1, full_bytes / ms1 / 1e6);
    }
    {
        constexpr int CAPW = 768;
        std::vector<int4> dw;
        for (int64_t r = 0; r < n;) { int64_t e = r; while (e < n && e - r < maxrows && rp[e + 1] - rp[r] <= CAPW) ++e; dw.push_back(make_int4((int)r, (int)(e - r), rp[r], rp[e])); r = e; }
        int4 *d_dw; CK(hipMalloc(&d_dw, dw.size() * sizeof(int4))); CK(hipMemcpy(d_dw, dw.data(), dw.size() * sizeof(int4), hipMemcpyHostToDevice));
        const int nbw = (int)dw.size();
        CK(hipMemsetAsync(d_y, 0, n * 8, s));
        float ms0 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_ws<CAPW, 0>), dim3(1792), dim3(320), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); });
        float ms1 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream_ws<CAPW, 1>), dim3(1792), dim3(320), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); });
        printf("grid 1792 k_stream_ws (writer wave, cap %d, %d row blocks): no-gather %.4f ms   full %.4f ms (full %.0f GB/s)\n", CAPW, nbw, ms0, ms1, full_bytes / ms1 / 1e6);
    }
#define SWEEP(CAPW, GRID) do { \
        std::vector<int4> dw; \
        for (int64_t r = 0; r < n;) { int64_t e = r; while (e < n && e - r < maxrows && rp[e + 1] - rp[r] <= CAPW) ++e; dw.push_back(make_int4((int)r, (int)(e - r), rp[r], rp[e])); r = e; } \
        int4 *d_dw; CK(hipMalloc(&d_dw, dw.size() * sizeof(int4))); CK(hipMemcpy(d_dw, dw.data(), dw.size() * sizeof(int4), hipMemcpyHostToDevice)); \
        const int nbw = (int)dw.size(); \
        float ms3 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream<3, CAPW>), dim3(GRID), dim3(BLK), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); }); \
        float ms4 = time_ms(s, reps, [&] { hipLaunchKernelGGL((k_stream<4, CAPW>), dim3(GRID), dim3(BLK), 0, s, nbw, d_dw, d_rp, d_col, d_val, d_x, d_y, d_out); }); \
        printf("k_stream cap %d grid %d (%d row blocks): stage3 %.4f  stage4 %.4f ms (full %.0f GB/s)\n", CAPW, GRID, nbw, ms3, ms4, full_bytes / ms4 / 1e6); \
        CK(hipFree(d_dw)); } while (0)

    { std::vector<double> hy(n); CK(hipMemcpy(hy.data(), d_y, n * 8, hipMemcpyDeviceToHost)); double mx = 0; int64_t bad = 0; for (int64_t i = 0; i < n; ++i) { double e = 0; for (int j = rp[i]; j < rp[i + 1]; ++j) e += val[j]; if (hy[i] != e) ++bad; } printf("final y check: %lld wrong rows\n", (long long)bad); }
    CK(hipDeviceSynchronize());
    return 0;
}
