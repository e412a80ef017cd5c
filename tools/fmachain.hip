// Latency of a dependent fp64 fused multiply-add chain on one wave (what bounds a long row's in-order walk):
//   regs   operands in registers
//   lds1   operands read from LDS by one lane, one batch of 8 ahead
//   lds64  same with all 64 lanes reading the same address
// build: hipcc --offload-arch=gfx950 -O3 -o tools/fmachain tools/fmachain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain_regs(double *out, double a, double b, int n)
{
    double acc = out[threadIdx.x];
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(a, b, acc);
    }
    out[threadIdx.x] = acc;
}
template <bool ONE>
__global__ void chain_lds(double *out, int n)
{
    __shared__ double2 s[2048 + 8];
    for (int i = threadIdx.x; i < 2048 + 8; i += blockDim.x) s[i] = make_double2(1.0 + i * 1e-9, 1e-9);
    __syncthreads();
    double acc = 0.0;
    if (!ONE || threadIdx.x == 0) {
        for (int rep = 0; rep < n / 2048; ++rep) {
            double2 pa[8], pb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pa[u] = s[u];
            for (int q = 0; q < 2048; q += 16) {
#pragma unroll
                for (int u = 0; u < 8; ++u) pb[u] = s[q + 8 + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma(pa[u].x, pa[u].y, acc);
#pragma unroll
                for (int u = 0; u < 8; ++u) pa[u] = s[(q + 16 + u) & 2047];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma(pb[u].x, pb[u].y, acc);
            }
        }
    }
    out[threadIdx.x] = acc;
}
int main()
{
    double *d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 1 << 22;
    for (int k = 0; k < 3; ++k) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL(chain_regs, dim3(1), dim3(64), 0, 0, d, 1.0000001, 1e-9, n);
            if (k == 1) hipLaunchKernelGGL(chain_lds<true>, dim3(1), dim3(64), 0, 0, d, n);
            if (k == 2) hipLaunchKernelGGL(chain_lds<false>, dim3(1), dim3(64), 0, 0, d, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.2f ns per multiply-add\n", k == 0 ? "regs" : k == 1 ? "lds1" : "lds64", ms * 1e6 / n);
        }
    }
    return 0;
}
