// elllab — lab for the "sliced, unsorted ELL" form of the exact SpMV (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/elllab tools/elllab.hip
//   ./tools/elllab [N=256] [reps=20]
// Layout under test: slices of S = 64*R rows; entry k (CRS order) of row j of slice s sits at
// val[(s*W + k)*S + j]; one u32 per row holds W 4-bit dictionary indices of col - row (0xF = no entry).
// A lane owns R rows and runs the in-order fma chain of spmv_crs_cpu (kernels.hpp:380-409): bit-exact.
// No LDS staging, no row_ptr, no barriers: every load is a coalesced wave access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BLK = 256;
typedef double v2d_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- reference: one lane per row, plain CSR
__global__ __launch_bounds__(BLK) void k_csr_ref(int n, const int *__restrict__ rp, const int *__restrict__ col,
                                                 const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t nth = (int64_t)gridDim.x * BLK;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += nth) {
        double acc = 0.0;
        for (int j = rp[i]; j < rp[i + 1]; ++j) acc = fma(val[j], x[col[j]], acc);
        y[i] = acc;
    }
}

__global__ void k_cmp(int n, const double *a, const double *b, unsigned long long *bad)
{
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nth)
        c += __double_as_longlong(a[i]) != __double_as_longlong(b[i]);
    if (c) atomicAdd(bad, c);
}

// ---------------------------------------------------------------- plain streaming read of the same bytes
__global__ __launch_bounds__(BLK) void k_read(const v2d_t *__restrict__ v, int64_t n2, double *out)
{
    const int64_t nth = (int64_t)gridDim.x * BLK;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n2; i += nth) { const v2d_t a = v[i]; acc += a.x + a.y; }
    if (acc == 1.2345e-300) out[threadIdx.x] = acc;
}

// ---------------------------------------------------------------- sliced ELL, R rows per lane
// work map: MAP 0 = XCD e (blockIdx % 8) owns the e-th contiguous eighth of the row blocks;
//           MAP 1 = period map: row block rb goes to XCD ((rb % PB) * 8) / PB with PB = blocks per period (the
//                   largest |col - row|, a z-plane): an XCD then sweeps a y-slab of every plane and the planes
//                   z-1 / z+1 it gathers from are the ones it has just read / will read next: x stays in its L2
//           MAP 2 = plain grid-stride over row blocks
// STAGE 0 loads only (val + idx), 1 + gathers + chain, 2 + y store, 3 + fused dot partial (CG form)
template <int W, int R, int STAGE, int MAP, int NTS>
__global__ __launch_bounds__(BLK) void k_ell(int n_rows, int n_blocks, int PB, const double *__restrict__ val,
                                             const uint32_t *__restrict__ idx, const int *__restrict__ dict,
                                             const double *__restrict__ x, double *__restrict__ y, double *__restrict__ part)
{
    __shared__ int s_dict[16];
    __shared__ double s_red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 16) s_dict[t] = dict[t];
    __syncthreads();
    constexpr int S = 64 * R;                 // rows per slice (one wave)
    constexpr int RB = 4 * S;                 // rows per row block (one workgroup trip)
    const int G = gridDim.x;
    int rb, rb_end, rb_step, per_next = 0, per_left = 0, per_skip = 0;
    if (MAP == 0) {
        const int e = blockIdx.x & 7, j = blockIdx.x >> 3;
        rb = (int)(((int64_t)e * n_blocks) / 8) + j; rb_end = (int)(((int64_t)(e + 1) * n_blocks) / 8); rb_step = G >> 3;
    } else if (MAP == 1) {
        // XCD e owns blocks [e*PB/8, (e+1)*PB/8) of every period; its G/8 workgroups interleave over that list
        const int e = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int b0 = (int)(((int64_t)e * PB) / 8), b1 = (int)(((int64_t)(e + 1) * PB) / 8);
        const int len = b1 - b0;              // blocks of this XCD per period
        // position q in the XCD's list <-> block (q / len) * PB + b0 + q % len
        per_left = len; per_skip = PB; per_next = b0;
        rb = j; rb_end = (n_blocks / PB) * len; rb_step = G >> 3;      // rb counts list positions here
    } else {
        rb = blockIdx.x; rb_end = n_blocks; rb_step = G;
    }
    double lane_sum = 0.0;
    for (; rb < rb_end; rb += rb_step) {
        int blk = rb;
        if (MAP == 1) blk = (rb / per_left) * per_skip + per_next + rb % per_left;
        const int sl = blk * 4 + w;                           // slice of this wave
        const int row0 = sl * S + lane * R;                   // first row of this lane
        if (row0 >= n_rows) continue;
        const double *vp = val + (size_t)sl * W * S + lane * R;
        double v[W][R];
        uint32_t iw[R];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (R >= 2) {
#pragma unroll
                for (int r = 0; r < R; r += 2) { const v2d_t q = *reinterpret_cast<const v2d_t *>(vp + (size_t)k * S + r); v[k][r] = q.x; v[k][r + (R > 1)] = q.y; }
            } else v[k][0] = vp[(size_t)k * S];
        }
        if (R == 4) { const uint4 q = *reinterpret_cast<const uint4 *>(idx + row0); iw[0] = q.x; iw[R > 1] = q.y; iw[R > 2 ? 2 : 0] = q.z; iw[R - 1] = q.w; }
        else if (R == 2) { const uint2 q = *reinterpret_cast<const uint2 *>(idx + row0); iw[0] = q.x; iw[R - 1] = q.y; }
        else iw[0] = idx[row0];
        if (STAGE == 0) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int r = 0; r < R; ++r) s += v[k][r];
#pragma unroll
            for (int r = 0; r < R; ++r) s += (double)iw[r];
            lane_sum += s;
            continue;
        }
        if (STAGE == 4) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double s = (double)iw[r];
#pragma unroll
                for (int k = 0; k < W; ++k) s += v[k][r];
                y[row0 + r] = s;
            }
            continue;
        }
        int c[W][R];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t nib = (iw[r] >> (4 * k)) & 15u;
                c[k][r] = nib == 15u ? row0 + r : row0 + r + s_dict[nib];
            }
        double xv[W][R];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r) xv[k][r] = x[c[k][r]];
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k)
                if (((iw[r] >> (4 * k)) & 15u) != 15u) acc[r] = fma(v[k][r], xv[k][r], acc[r]);
        }
        if (STAGE == 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) lane_sum += acc[r];
            continue;
        }
        if (R >= 2) {
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                v2d_t q; q.x = acc[r]; q.y = acc[r + (R > 1)];
                if (NTS) __builtin_nontemporal_store(q, reinterpret_cast<v2d_t *>(y + row0 + r));
                else *reinterpret_cast<v2d_t *>(y + row0 + r) = q;
            }
        } else {
            if (NTS) __builtin_nontemporal_store(acc[0], y + row0); else y[row0] = acc[0];
        }
        if (STAGE == 3) {
#pragma unroll
            for (int r = 0; r < R; ++r) lane_sum = fma(acc[r], x[row0 + r], lane_sum);
        }
    }
    if (STAGE != 2 && STAGE != 4) {
        double s = lane_sum;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) s_red[w] = s;
        __syncthreads();
        if (t == 0) part[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

// ---------------------------------------------------------------- pipelined form (CG form: store + dot)
// Per trip: columns of block i (index words already in registers) -> x gathers of block i -> loads of block i+1
// (index words + W value loads, unconditional, clamped) -> chain of block i -> store.  Vector memory returns in
// issue order, so the gathers go out BEFORE the prefetch and the wait in front of the chain leaves it in flight.
template <int W, int R, int NTS>
__global__ __launch_bounds__(BLK) void k_ell_pipe(int n_rows, int n_blocks, int PB, const double *__restrict__ val,
                                                  const uint32_t *__restrict__ idx, const int *__restrict__ dict,
                                                  const double *__restrict__ x, double *__restrict__ y, double *__restrict__ part)
{
    __shared__ int s_dict[16];
    __shared__ double s_red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 16) s_dict[t] = dict[t];
    __syncthreads();
    constexpr int S = 64 * R;
    const int G = gridDim.x;
    const int e = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b0 = (int)(((int64_t)e * PB) / 8), b1 = (int)(((int64_t)(e + 1) * PB) / 8);
    const int len = b1 - b0;
    const int q_end = (n_blocks / PB) * len, q_step = G >> 3;
    double lane_sum = 0.0;
    if (j >= q_end) { if (t == 0) part[blockIdx.x] = 0.0; return; }
    double v[W][R], vn[W][R];
    uint32_t iw[R], iwn[R];
#define LOADBLK(VV, II, qq)                                                                         \
    {                                                                                               \
        const int blk_ = ((qq) / len) * PB + b0 + (qq) % len;                                       \
        const int sl_ = blk_ * 4 + w;                                                               \
        const int r0_ = sl_ * S + lane * R;                                                         \
        if (R == 2) { const uint2 q2 = *reinterpret_cast<const uint2 *>(idx + r0_); II[0] = q2.x; II[R - 1] = q2.y; } \
        else II[0] = idx[r0_];                                                                      \
        const double *vp_ = val + (size_t)sl_ * W * S + lane * R;                                   \
        _Pragma("unroll") for (int k = 0; k < W; ++k) {                                             \
            if (R == 2) { const v2d_t q2 = *reinterpret_cast<const v2d_t *>(vp_ + (size_t)k * S); VV[k][0] = q2.x; VV[k][R - 1] = q2.y; } \
            else VV[k][0] = vp_[(size_t)k * S];                                                     \
        }                                                                                           \
    }
    LOADBLK(v, iw, j)
    for (int q = j; q < q_end; q += q_step) {
        const int blk = (q / len) * PB + b0 + q % len;
        const int row0 = (blk * 4 + w) * S + lane * R;
        int c[W][R];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t nib = (iw[r] >> (4 * k)) & 15u;
                c[k][r] = nib == 15u ? 0 : row0 + r + s_dict[nib];
            }
        double xv[W][R], xd[R];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r) xv[k][r] = x[c[k][r]];
        if (R == 2) { const v2d_t q2 = *reinterpret_cast<const v2d_t *>(x + row0); xd[0] = q2.x; xd[R - 1] = q2.y; }
        else xd[0] = x[row0];
        const int qn = q + q_step < q_end ? q + q_step : q;
        LOADBLK(vn, iwn, qn)
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k)
                if (((iw[r] >> (4 * k)) & 15u) != 15u) acc[r] = fma(v[k][r], xv[k][r], acc[r]);
            lane_sum = fma(acc[r], xd[r], lane_sum);
        }
        if (R == 2) {
            v2d_t q2; q2.x = acc[0]; q2.y = acc[R - 1];
            if (NTS) __builtin_nontemporal_store(q2, reinterpret_cast<v2d_t *>(y + row0));
            else *reinterpret_cast<v2d_t *>(y + row0) = q2;
        } else {
            if (NTS) __builtin_nontemporal_store(acc[0], y + row0); else y[row0] = acc[0];
        }
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r) v[k][r] = vn[k][r];
#pragma unroll
        for (int r = 0; r < R; ++r) iw[r] = iwn[r];
    }
#undef LOADBLK
    double s = lane_sum;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) s_red[w] = s;
    __syncthreads();
    if (t == 0) part[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ---------------------------------------------------------------- host
static inline uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline double u01(uint64_t k) { return (double)(mix(k) >> 11) * (1.0 / 9007199254740992.0); }

struct Lab {
    int N, n; int64_t nnz;
    int *rp, *col; double *val, *x, *y_ref, *y, *part;
    double *ev[5]; uint32_t *ei[5]; int *dict;      // ELL images for R = 1, 2 (index by R)
    hipEvent_t e0, e1;
    int reps;
};

template <class F> static float time_it(Lab &L, F f)
{
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(L.e0));
    for (int i = 0; i < L.reps; ++i) f();
    CK(hipEventRecord(L.e1));
    CK(hipEventSynchronize(L.e1));
    float ms; CK(hipEventElapsedTime(&ms, L.e0, L.e1));
    CK(hipGetLastError());
    return ms / L.reps;
}

static unsigned long long mismatches(Lab &L)
{
    unsigned long long *bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_cmp, dim3(1024), dim3(256), 0, 0, L.n, L.y_ref, L.y, bad);
    unsigned long long h; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost)); CK(hipFree(bad));
    return h;
}

template <int R, int STAGE, int MAP, int NTS>
static void run_ell(Lab &L, int grid, const char *tag)
{
    constexpr int W = 7;
    const int RB = 256 * R;
    const int n_blocks = (L.n + RB - 1) / RB;
    const int PB = (L.N * L.N) / RB;            // row blocks per z-plane
    if (MAP == 1 && (PB < 8 || n_blocks % PB)) { printf("%-44s skipped (period map needs >= 8 blocks per plane)\n", tag); return; }
    CK(hipMemset(L.y, 0xff, (size_t)L.n * 8));
    auto f = [&] { hipLaunchKernelGGL((k_ell<W, R, STAGE, MAP, NTS>), dim3(grid), dim3(BLK), 0, 0, L.n, n_blocks, PB, L.ev[R], L.ei[R], L.dict, L.x, L.y, L.part); };
    const float ms = time_it(L, f);
    const double alg = 12.0 * L.nnz + 4.0 * (L.n + 1) + 16.0 * L.n;
    const double str = 8.0 * 7 * L.n + 4.0 * L.n + 16.0 * L.n;
    char chk[64] = "";
    if (STAGE == 2 || STAGE == 3) snprintf(chk, sizeof chk, " mismatches=%llu", mismatches(L));
    printf("%-44s grid %5d  %.4f ms  alg %.0f GB/s (%.3f of 8T)  streamed %.0f GB/s%s\n", tag, grid, ms, alg / ms / 1e6, alg / ms / 1e6 / 8000.0, str / ms / 1e6, chk);
    fflush(stdout);
}

template <int R, int NTS>
static void run_pipe(Lab &L, int grid, const char *tag)
{
    constexpr int W = 7;
    const int RB = 256 * R;
    const int n_blocks = (L.n + RB - 1) / RB;
    const int PB = (L.N * L.N) / RB;
    if (PB < 8 || n_blocks % PB) { printf("%-44s skipped\n", tag); return; }
    CK(hipMemset(L.y, 0xff, (size_t)L.n * 8));
    auto f = [&] { hipLaunchKernelGGL((k_ell_pipe<W, R, NTS>), dim3(grid), dim3(BLK), 0, 0, L.n, n_blocks, PB, L.ev[R], L.ei[R], L.dict, L.x, L.y, L.part); };
    const float ms = time_it(L, f);
    const double alg = 12.0 * L.nnz + 4.0 * (L.n + 1) + 16.0 * L.n;
    printf("%-44s grid %5d  %.4f ms  alg %.0f GB/s (%.3f of 8T) mismatches=%llu\n", tag, grid, ms, alg / ms / 1e6, alg / ms / 1e6 / 8000.0, mismatches(L));
    fflush(stdout);
}

int main(int argc, char **argv)
{
    Lab L;
    L.N = argc > 1 ? atoi(argv[1]) : 256;
    L.reps = argc > 2 ? atoi(argv[2]) : 20;
    const int N = L.N;
    const int n = L.n = N * N * N;
    std::vector<int> rp(n + 1);
    std::vector<int> col; col.reserve((size_t)7 * n);
    std::vector<double> val; val.reserve((size_t)7 * n);
    const int offs[7] = {-N * N, -N, -1, 0, 1, N, N * N};
    rp[0] = 0;
    for (int z = 0; z < N; ++z) for (int y = 0; y < N; ++y) for (int xx = 0; xx < N; ++xx) {
        const int r = (z * N + y) * N + xx;
        const bool ok[7] = {z > 0, y > 0, xx > 0, true, xx < N - 1, y < N - 1, z < N - 1};
        for (int k = 0; k < 7; ++k) if (ok[k]) { col.push_back(r + offs[k]); val.push_back(k == 3 ? 6.0 + u01(r) : -u01((uint64_t)r * 7 + k + 1000003)); }
        rp[r + 1] = (int)col.size();
    }
    L.nnz = (int64_t)col.size();
    std::vector<double> x(n);
    for (int i = 0; i < n; ++i) x[i] = u01((uint64_t)i + 77777777) - 0.5;
    printf("7-pt %d^3: n = %d, nnz = %lld\n", N, n, (long long)L.nnz);
    CK(hipMalloc(&L.rp, (size_t)(n + 1) * 4)); CK(hipMalloc(&L.col, (size_t)L.nnz * 4)); CK(hipMalloc(&L.val, (size_t)L.nnz * 8));
    CK(hipMalloc(&L.x, (size_t)n * 8)); CK(hipMalloc(&L.y, (size_t)n * 8)); CK(hipMalloc(&L.y_ref, (size_t)n * 8)); CK(hipMalloc(&L.part, 65536 * 8));
    CK(hipMemcpy(L.rp, rp.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(L.col, col.data(), (size_t)L.nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(L.val, val.data(), (size_t)L.nnz * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(L.x, x.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    CK(hipEventCreate(&L.e0)); CK(hipEventCreate(&L.e1));
    // dictionary = the 7 offsets; ELL images for R = 1, 2
    int dict[16] = {0}; for (int k = 0; k < 7; ++k) dict[k] = offs[k];
    CK(hipMalloc(&L.dict, 64)); CK(hipMemcpy(L.dict, dict, 64, hipMemcpyHostToDevice));
    for (int R = 1; R <= 4; R += (R == 2 ? 2 : 1)) {
        const int S = 64 * R;
        const int ns = (n + S - 1) / S;
        std::vector<double> ev((size_t)ns * 7 * S, 0.0);
        std::vector<uint32_t> ei((size_t)ns * S, 0xffffffffu);
        for (int r = 0; r < n; ++r) {
            const int s = r / S, j = r % S;
            uint32_t w = 0xffffffffu;
            for (int q = rp[r]; q < rp[r + 1]; ++q) {
                const int k = q - rp[r];
                ev[((size_t)s * 7 + k) * S + j] = val[q];
                int d = 0; while (dict[d] != col[q] - r) ++d;
                w = (w & ~(15u << (4 * k))) | ((uint32_t)d << (4 * k));
            }
            ei[r] = w;
        }
        CK(hipMalloc(&L.ev[R], ev.size() * 8)); CK(hipMalloc(&L.ei[R], ei.size() * 4));
        CK(hipMemcpy(L.ev[R], ev.data(), ev.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(L.ei[R], ei.data(), ei.size() * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_csr_ref, dim3(4096), dim3(BLK), 0, 0, n, L.rp, L.col, L.val, L.x, L.y_ref);
    CK(hipDeviceSynchronize());
    {
        auto f = [&] { hipLaunchKernelGGL(k_csr_ref, dim3(4096), dim3(BLK), 0, 0, n, L.rp, L.col, L.val, L.x, L.y); };
        const float ms = time_it(L, f);
        printf("%-44s %.4f ms\n", "naive CSR, one lane per row", ms);
        auto g = [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(BLK), 0, 0, reinterpret_cast<const v2d_t *>(L.ev[1]), (int64_t)7 * n / 2, L.part); };
        const float ms2 = time_it(L, g);
        printf("%-44s %.4f ms  %.0f GB/s\n", "plain read of the ELL val array (16 B/lane)", ms2, 56.0 * n / ms2 / 1e6);
    }
    const int sweep = argc > 3 ? atoi(argv[3]) : 0;
    if (sweep == 1) {
        for (int g : {256, 512, 768, 1024, 2048}) run_pipe<2, 0>(L, g, "PIPE R2 grid sweep");
        for (int g : {256, 512, 768, 1024, 2048}) run_pipe<2, 1>(L, g, "PIPE R2 nt grid sweep");
        for (int g : {256, 512, 768, 1024, 2048}) run_pipe<1, 0>(L, g, "PIPE R1 grid sweep");
        for (int g : {256, 512, 768, 1024, 2048}) run_pipe<1, 1>(L, g, "PIPE R1 nt grid sweep");
        for (int g : {512, 768}) run_ell<2, 3, 1, 1>(L, g, "R2 stage3 map1 nt (no pipe)");
    }
    if (sweep == 0) {
        run_ell<1, 0, 0, 0>(L, 2048, "R1 stage0 (val+idx loads) map0");
        run_ell<1, 1, 1, 0>(L, 2048, "R1 stage1 map1 (period)");
        run_ell<1, 4, 1, 0>(L, 2048, "R1 stage4 (loads + store, no gathers) map1");
        run_ell<1, 3, 1, 0>(L, 2048, "R1 stage3 map1 (period)");
        for (int g : {256, 512, 768, 1024}) run_ell<1, 3, 1, 0>(L, g, "R1 stage3 map1 grid sweep");
        for (int g : {256, 512, 768, 1024}) run_ell<1, 3, 1, 1>(L, g, "R1 stage3 map1 nt grid sweep");
        run_ell<2, 1, 1, 0>(L, 512, "R2 stage1 map1");
        run_ell<2, 4, 1, 0>(L, 512, "R2 stage4 (loads + store, no gathers) map1");
        for (int g : {256, 384, 512, 640, 768, 1024}) run_ell<2, 3, 1, 0>(L, g, "R2 stage3 map1 grid sweep");
        for (int g : {256, 384, 512, 640, 768, 1024}) run_ell<2, 3, 1, 1>(L, g, "R2 stage3 map1 nt grid sweep");
        for (int g : {256, 512, 768}) run_ell<2, 3, 0, 0>(L, g, "R2 stage3 map0 grid sweep");
        for (int g : {256, 512, 768}) run_ell<2, 2, 1, 0>(L, g, "R2 stage2 map1 grid sweep");
        for (int g : {256, 512, 768}) run_ell<4, 3, 1, 0>(L, g, "R4 stage3 map1 grid sweep");
        for (int g : {256, 512, 768}) run_ell<4, 3, 1, 1>(L, g, "R4 stage3 map1 nt grid sweep");
    }
    return 0;
}
