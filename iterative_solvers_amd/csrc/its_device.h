// its_device.h — device-side building blocks shared by all gfx950 kernels of libits_hip.
// Wavefront = 64 lanes; workgroups are 256 threads (4 waves, one per SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ITS_BLOCK 256
#define ITS_WAVE 64

// Number of partial sums every reduction leaves behind (one per workgroup of the producing
// kernel, zero-filled up to this count).  Consumers re-sum all of them in a fixed order, so a
// scalar is bit-identical in every workgroup and from run to run.
#define ITS_NPART 2048

namespace its {

__device__ __forceinline__ double wave_sum(double v)
{
    // butterfly over the 64 lanes: fixed order, every lane ends with the same value
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, ITS_WAVE);
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, ITS_WAVE));
    return v;
}

// Sum over the 256 threads of a workgroup; result returned to every thread.  `red` is 4 doubles
// of LDS.  Two barriers; safe to call repeatedly with the same scratch.
__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();                 // previous users of red[] are done
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ double block_max(double v, double *red)
{
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// One workgroup's partial of a reduction: slot blockIdx.x, and zeros in the slots a smaller-than-ITS_NPART grid
// leaves unowned, so that consumers can always sum all ITS_NPART.  Call from EVERY thread of the workgroup (v is
// taken from thread 0; the zero slots are shared out over the threads).
__device__ __forceinline__ void store_partial(double *__restrict__ part, double v)
{
    if (threadIdx.x == 0) part[blockIdx.x] = v;
    for (int z = blockIdx.x + gridDim.x * (1 + (int)threadIdx.x); z < ITS_NPART; z += gridDim.x * ITS_BLOCK) part[z] = 0.0;
}

// Fixed-order sum of the ITS_NPART partials a previous kernel wrote (kernel boundary makes
// them visible).  Every workgroup gets the same bits.
__device__ __forceinline__ double sum_partials(const double *__restrict__ part, double *red)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < ITS_NPART / ITS_BLOCK; ++k) s += part[threadIdx.x + k * ITS_BLOCK];
    return block_sum(s, red);
}

__device__ __forceinline__ double max_partials(const double *__restrict__ part, double *red)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < ITS_NPART / ITS_BLOCK; ++k) s = fmax(s, part[threadIdx.x + k * ITS_BLOCK]);
    return block_max(s, red);
}

// XCD-aware work placement: workgroups are dealt round-robin over the 8 XCDs, so groups g and
// g+8 share an L2.  Map group g of a grid G (G % 8 == 0) to slot (g%8)*(G/8) + g/8: each XCD then
// owns one contiguous eighth of the work and its L2 holds that eighth's x window.  Speed only.
__device__ __forceinline__ int xcd_slot(int g, int G) { return (g & 7) * (G >> 3) + (g >> 3); }

}  // namespace its
