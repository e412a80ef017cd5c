// What one wave pays per LDS read (all lanes the same address, as in the walks of spmv_long_kernel / sweep_long_rows)
// and per dependent fp64 multiply-add, alone and together.   hipcc --offload-arch=gfx950 -O3 -o ldsrate ldsrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int N = 4096;                    // entries in the strip
// MODE 0: 16 b128 reads per trip, summed without a chain (reads only matter)   1: the same with b64 reads
// MODE 2: dependent multiply-adds on registers only                             3: batch of 16 reads, then the 16 multiply-adds
template <int MODE>
__global__ __launch_bounds__(64) void k(const double2 *src, double *out, int reps)
{
    __shared__ double2 s[N + 16];
    for (int i = threadIdx.x; i < N + 16; i += 64) s[i] = src[i % N];
    __syncthreads();
    double acc = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int q = 0; q < N; q += 16) {
            if (MODE == 0) {
                double2 p[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) p[u] = s[q + u];
#pragma unroll
                for (int u = 0; u < 16; u += 4) { acc += p[u].x; a2 += p[u + 1].x; a3 += p[u + 2].x; a4 += p[u + 3].x; }
            } else if (MODE == 1) {
                const double *sd = reinterpret_cast<const double *>(s);
                double p[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) p[u] = sd[2 * (q + u)];
#pragma unroll
                for (int u = 0; u < 16; u += 4) { acc += p[u]; a2 += p[u + 1]; a3 += p[u + 2]; a4 += p[u + 3]; }
            } else if (MODE == 2) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = fma(a2, a3, acc);
                asm volatile("" : "+v"(acc));
            } else {
                double2 p[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) p[u] = s[q + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = fma(p[u].x, p[u].y, acc);
            }
        }
    }
    out[threadIdx.x] = acc + a2 + a3 + a4;
}
template <int MODE> static int run(const double2 *src, double *out, const char *what)
{
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int reps = 200;
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, src, out, 2);
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, src, out, reps);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    printf("%-60s %.2f ns per entry\n", what, ms * 1e6 / ((double)reps * N));
    return 0;
}
int main()
{
    double2 *src; double *out;
    CHK(hipMalloc(&src, N * sizeof(double2))); CHK(hipMalloc(&out, 64 * sizeof(double)));
    CHK(hipMemset(src, 0, N * sizeof(double2)));
    if (run<0>(src, out, "ds_read_b128, one address for the wave, no chain")) return 1;
    if (run<1>(src, out, "ds_read_b64, one address for the wave, no chain")) return 1;
    if (run<2>(src, out, "dependent v_fma_f64 on registers")) return 1;
    if (run<3>(src, out, "16 ds_read_b128 then their 16 dependent v_fma_f64")) return 1;
    return 0;
}
