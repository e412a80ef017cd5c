// gatherlab — ceiling of random 8-byte gathers from a table that sits in the Infinity Cache but not in L2
// (the x vector of the 10 M-row power-law SpMV: 80 MB).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gatherlab tools/gatherlab.hip && ./tools/gatherlab [n_table=10000000] [n_gather=84000000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U, int NT>
__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ idx, const double *__restrict__ x, int64_t n, double *out)
{
    const int64_t nth = (int64_t)gridDim.x * 256;
    double acc = 0.0;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * nth < n; i += U * nth) {
        int c[U]; double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = idx[i + u * nth];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + c[u]) : x[c[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n; i += nth) acc += x[idx[i]];
    if (acc == 1.2345e-300) out[threadIdx.x] = acc;
}

// 16-byte variant: lanes gather double2 at even indices (half as many requests for the same lines? no: same lines)
int main(int argc, char **argv)
{
    const int nt = argc > 1 ? atoi(argv[1]) : 10000000;
    const int64_t ng = argc > 2 ? atoll(argv[2]) : 84000000;
    std::vector<int> idx(ng);
    uint64_t s = 12345;
    for (int64_t i = 0; i < ng; ++i) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; idx[i] = (int)(z % (uint64_t)nt); }
    int *didx; double *dx, *dout;
    CK(hipMalloc(&didx, ng * 4)); CK(hipMalloc(&dx, (size_t)nt * 8)); CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(didx, idx.data(), ng * 4, hipMemcpyHostToDevice)); CK(hipMemset(dx, 0, (size_t)nt * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *tag, auto kern, int grid) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, didx, dx, ng, dout);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, didx, dx, ng, dout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("%-28s grid %5d  %.4f ms  %.1f G gathers/s  (%.2f TB/s at 64 B per gather)\n", tag, grid, ms, ng / ms / 1e6, ng * 64.0 / ms / 1e9);
    };
    printf("table %d doubles (%.0f MB), %lld gathers\n", nt, nt * 8e-6, (long long)ng);
    for (int g : {1024, 2048, 4096, 8192}) run("U=4", k_gather<4, 0>, g);
    for (int g : {1024, 2048, 4096, 8192}) run("U=8", k_gather<8, 0>, g);
    for (int g : {1024, 2048, 4096}) run("U=16", k_gather<16, 0>, g);
    for (int g : {2048, 4096}) run("U=8 nontemporal", k_gather<8, 1>, g);
    return 0;
}
