// its_ellg.hip — the exact SpMV on the sliced ELL image with EXPLICIT columns: any matrix whose longest row has at most
// 8 entries, whatever its columns (round 3; its_ell.hip needs a matrix-wide dictionary of <= 15 / <= 255 col - row).
//
// Same slices, same walk, same bits as spmv_ell_kernel: slices of ELL_S = 128 rows, values slot-major
// (ell_val[(s*W + k)*128 + j]), a lane owns two adjacent rows and runs each row's in-order fused multiply-add chain of
// spmv_crs_cpu (kernels.hpp:380-409) from 0.0; every load is a coalesced 16-byte-per-lane wave access, no LDS, no
// barrier, persistent grid with the XCD period map.  What differs is how a column is named:
//     col = ellg_base[s*W + k] + delta,
// one 32-bit base per (slice, slot) — the smallest column any row of the slice has in that slot — and a delta of 8, 16
// or 32 bits, the width chosen PER SLICE (ellg_desc[s] = first index word of the slice / 128 << 2 | width code).
// Index words are slot-major like the values: word q of row j at idx[(first + q)*128 + j], holding 4 / 2 / 1 deltas;
// all-ones = no entry.
// Why that is small for matrices with any locality: the contract fixes the ORDER of a row's multiply-adds, not the slot an
// entry sits in, so the builder may leave slots empty.  It aligns every row of a slice to the slice's longest row (greedy,
// in order: an entry takes the first free slot whose reference entry has the same col - row, else the next free slot),
// so that rows which lack an entry — the boundary rows of a grid — keep their other entries in the slots their
// neighbours use; a slot then spans little more than the slice's 128 rows and the deltas fit a byte.  7-pt 256^3: every
// slice 8-bit, 56 (values) + 8 (deltas) + 0.25 (bases, descriptor) + 16 (x, y) = 80 bytes per row against 104 of CRS and
// 76 of the nibble-dictionary form.  A matrix without locality (randomly permuted rows) gets 32-bit deltas: 100 bytes per
// row, still without row_ptr, staging or barriers.
#include "its_kernels.h"
#include "its_epilogue.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace its {

typedef double ellg_v2d __attribute__((ext_vector_type(2)));

template <int W, int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_ellg_kernel(
    int n_rows, int n_blocks, int pb, const double *__restrict__ val, const uint32_t *__restrict__ idx,
    const uint32_t *__restrict__ desc, const int *__restrict__ base, int nt_store, SpmvArgs a)
{
    __shared__ double s_red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int G = gridDim.x;
    const int e = blockIdx.x & 7, j = blockIdx.x >> 3;
    // list of row blocks of XCD e: position q <-> block (q / len) * per + b0 + q % len   (as spmv_ell_kernel)
    int b0, len, per, q_end;
    if (pb >= 8) {
        b0 = (int)(((int64_t)e * pb) / 8);
        len = (int)(((int64_t)(e + 1) * pb) / 8) - b0;
        per = pb;
        q_end = ((n_blocks + pb - 1) / pb) * len;
    } else {
        b0 = (int)(((int64_t)e * n_blocks) / 8);
        len = (int)(((int64_t)(e + 1) * n_blocks) / 8) - b0;
        per = 0;
        q_end = len;
    }
    double lane_sum = 0.0, lane_max = 0.0;
    for (int q = j; q < q_end; q += G >> 3) {
        const int blk = (q / len) * per + b0 + q % len;
        if (blk >= n_blocks) continue;
        const int sl = __builtin_amdgcn_readfirstlane(blk * 4 + w);     // one slice per wave: descriptor and bases are wave-uniform
        const int row0 = sl * ELL_S + lane * 2;
        const bool any = row0 < n_rows, full = row0 + 1 < n_rows;
        const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)desc[sl]);
        const int wc = (int)(d & 3u);
        const uint32_t *ip = idx + (size_t)(d >> 2) * ELL_S + lane * 2;
        int bs[W];
#pragma unroll
        for (int k = 0; k < W; ++k) bs[k] = __builtin_amdgcn_readfirstlane(base[(size_t)sl * W + k]);
        // index words first (the gathers depend on them), then the values
        int c[W][2];
        uint32_t present[2] = {0u, 0u};
        if (wc == 0) {
            constexpr int WQ = (W + 3) / 4;
            uint2 iw[WQ];
#pragma unroll
            for (int qq = 0; qq < WQ; ++qq) iw[qq] = *reinterpret_cast<const uint2 *>(ip + (size_t)qq * ELL_S);
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t dl = ((r ? iw[k >> 2].y : iw[k >> 2].x) >> (8 * (k & 3))) & 255u;
                    const bool p = dl != 255u;
                    c[k][r] = p ? bs[k] + (int)dl : 0;                     // no entry: gather x[0], never used
                    present[r] |= (p ? 1u : 0u) << k;
                }
        } else if (wc == 1) {
            constexpr int WQ = (W + 1) / 2;
            uint2 iw[WQ];
#pragma unroll
            for (int qq = 0; qq < WQ; ++qq) iw[qq] = *reinterpret_cast<const uint2 *>(ip + (size_t)qq * ELL_S);
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t dl = ((r ? iw[k >> 1].y : iw[k >> 1].x) >> (16 * (k & 1))) & 65535u;
                    const bool p = dl != 65535u;
                    c[k][r] = p ? bs[k] + (int)dl : 0;
                    present[r] |= (p ? 1u : 0u) << k;
                }
        } else {
            uint2 iw[W];
#pragma unroll
            for (int k = 0; k < W; ++k) iw[k] = *reinterpret_cast<const uint2 *>(ip + (size_t)k * ELL_S);
#pragma unroll
            for (int k = 0; k < W; ++k)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t dl = r ? iw[k].y : iw[k].x;
                    const bool p = dl != 0xffffffffu;
                    c[k][r] = p ? (int)((uint32_t)bs[k] + dl) : 0;
                    present[r] |= (p ? 1u : 0u) << k;
                }
        }
        const double *vp = val + (size_t)sl * W * ELL_S + lane * 2;
        double v[W][2];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const ellg_v2d q2 = *reinterpret_cast<const ellg_v2d *>(vp + (size_t)k * ELL_S);
            v[k][0] = q2.x; v[k][1] = q2.y;
        }
        double e0[2] = {0.0, 0.0}, e1[2] = {0.0, 0.0}, e2[2] = {0.0, 0.0};
#define ELL_LD2(dst, ptr)                                                                 \
        do {                                                                              \
            if (full) { const ellg_v2d o2 = *reinterpret_cast<const ellg_v2d *>((ptr) + row0); dst[0] = o2.x; dst[1] = o2.y; } \
            else if (any) dst[0] = (ptr)[row0];                                           \
        } while (0)
        if (EPI == EPI_STORE && DOT) ELL_LD2(e0, a.dotv);
        if (EPI == EPI_RESIDUAL || EPI == EPI_JACOBI_REF || EPI == EPI_GS_RHS || EpiKind<EPI>::fused) ELL_LD2(e0, a.b);
        if (EPI == EPI_JACOBI_SHIPPED) { ELL_LD2(e0, a.D); ELL_LD2(e1, a.x); ELL_LD2(e2, a.b); }
        double a0[2] = {0.0, 0.0};                               // where the full chain of the fused Gauss-Seidel pass starts (SpmvArgs::acc0)
        if (EPI == EPI_GS_FUSED && a.acc0) ELL_LD2(a0, a.acc0);
#undef ELL_LD2
        double xv[W][2];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r) xv[k][r] = a.x[c[k][r]];
        double outv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0, acc2 = a0[r], diag = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k)
                if ((present[r] >> k) & 1u) chain_step<EPI>(c[k][r], row0 + r, v[k][r], xv[k][r], acc, acc2, diag);
            outv[r] = 0.0;
            if (r == 0 ? any : full) outv[r] = row_finish<EPI, DOT>(acc, acc2, diag, e0[r], e1[r], e2[r], lane_sum, lane_max);
        }
        if (a.y) {                                              // y == nullptr: only the norm is wanted
            if (full) {
                ellg_v2d o2; o2.x = outv[0]; o2.y = outv[1];
                if (nt_store) __builtin_nontemporal_store(o2, reinterpret_cast<ellg_v2d *>(a.y + row0));
                else *reinterpret_cast<ellg_v2d *>(a.y + row0) = o2;
            } else if (any) a.y[row0] = outv[0];
        }
    }
    if (DOT || EpiKind<EPI>::norm) {
        const double bsum = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) {
            a.part_sum[blockIdx.x] = bsum;
            for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_sum[z] = 0.0;
        }
        if (EpiKind<EPI>::norm && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (t == 0) {
                a.part_max[blockIdx.x] = bm;
                for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_max[z] = 0.0;
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// Image builder, on the device from the CRS arrays just uploaded: one workgroup of 128 lanes per slice, a lane per row.
// Pass WRITE = false: the delta width the slice needs (one byte per slice; the host turns the widths into the slices'
// first index words).  Pass WRITE = true: bases, values, index words.  Both passes run the same alignment.
// --------------------------------------------------------------------------------------
template <bool WRITE>
__global__ __launch_bounds__(ELL_S) void ellg_build_kernel(
    int n_rows, int W, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
    unsigned char *__restrict__ width, const uint32_t *__restrict__ desc, double *__restrict__ ell_val,
    uint32_t *__restrict__ ell_idx, int *__restrict__ base)
{
    __shared__ int s_key, s_ref[8], s_min[8], s_max[8];
    const int s = blockIdx.x, j = threadIdx.x;
    const int64_t row64 = (int64_t)s * ELL_S + j;
    const int row = (int)row64;
    int len = 0, j0 = 0;
    if (row64 < n_rows) { j0 = rp[row]; len = rp[row + 1] - j0; }
    int cc[8]; double vv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { cc[k] = 0; vv[k] = 0.0; if (k < len) { cc[k] = col[j0 + k]; vv[k] = val[j0 + k]; } }
    if (j == 0) s_key = -1;
    if (j < 8) { s_min[j] = 2147483647; s_max[j] = -2147483647 - 1; s_ref[j] = 0; }
    __syncthreads();
    // reference row: the longest of the slice, the first of those
    atomicMax(&s_key, (len << 8) | (ELL_S - 1 - j));
    __syncthreads();
    const int key = s_key, wref = key >> 8, jref = ELL_S - 1 - (key & 255);
    if (j == jref) {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < len) s_ref[k] = cc[k] - row;
    }
    __syncthreads();
    int ref[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ref[k] = s_ref[k];
    // greedy in-order alignment: entry e may take slots [smin, wref - (len - e)]
    int slot[8];
    int smin = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        slot[e] = -1;
        if (e < len) {
            const int hi = wref - (len - e), off = cc[e] - row;
            int pick = smin;
            bool found = false;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2)
                if (!found && s2 >= smin && s2 <= hi && ref[s2] == off) { pick = s2; found = true; }
            slot[e] = pick;
            smin = pick + 1;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e < len) { atomicMin(&s_min[slot[e]], cc[e]); atomicMax(&s_max[slot[e]], cc[e]); }
    __syncthreads();
    uint32_t range = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (s_min[k] != 2147483647) range = max(range, (uint32_t)s_max[k] - (uint32_t)s_min[k]);
    const int code = range < 255u ? 0 : range < 65535u ? 1 : 2;
    if (!WRITE) {
        if (j == 0) width[s] = (unsigned char)code;
        return;
    }
    // slot -> entry of this row
    int bsk[8]; uint32_t dl[8]; double ev[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bsk[k] = s_min[k] == 2147483647 ? 0 : s_min[k];
        dl[k] = 0xffffffffu; ev[k] = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < len && slot[e] == k) { dl[k] = (uint32_t)cc[e] - (uint32_t)bsk[k]; ev[k] = vv[e]; }
    }
    if (j < W) base[(size_t)s * W + j] = s_min[j] == 2147483647 ? 0 : s_min[j];
    for (int k = 0; k < W; ++k) ell_val[((size_t)s * W + k) * ELL_S + j] = ev[k];
    uint32_t *ip = ell_idx + (size_t)(desc[s] >> 2) * ELL_S + j;
    if (code == 0) {
        const int WQ = (W + 3) / 4;
        for (int qq = 0; qq < WQ; ++qq) {
            uint32_t wd = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t by = 255u;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k == 4 * qq + u && k < W && dl[k] != 0xffffffffu) by = dl[k];
                wd |= by << (8 * u);
            }
            ip[(size_t)qq * ELL_S] = wd;
        }
    } else if (code == 1) {
        const int WQ = (W + 1) / 2;
        for (int qq = 0; qq < WQ; ++qq) {
            uint32_t wd = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint32_t hw = 65535u;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k == 2 * qq + u && k < W && dl[k] != 0xffffffffu) hw = dl[k];
                wd |= hw << (16 * u);
            }
            ip[(size_t)qq * ELL_S] = wd;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < W) ip[(size_t)k * ELL_S] = dl[k];
    }
}

static inline int ellg_words(int W, int code) { return code == 0 ? (W + 3) / 4 : code == 1 ? (W + 1) / 2 : W; }

// Builds the image of A (rows <= 8 entries; checked by the caller).  Allocations are the caller's except the index words,
// whose count is only known after the first pass: *idx_out is hipMalloc'ed here.  Returns a hipError_t.
hipError_t ellg_build(CsrDev &A, hipStream_t s, int64_t *idx_words_out, int *hist /*[3]*/)
{
    const int n_slices = A.ell_blocks * (ELL_RB / ELL_S);
    const int W = A.ell_w;
    unsigned char *width_d = nullptr;
    hipError_t rc = hipMalloc((void **)&width_d, (size_t)n_slices);
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL(ellg_build_kernel<false>, dim3(n_slices), dim3(ELL_S), 0, s, A.n_rows, W, A.row_ptr, A.col, A.val,
                       width_d, (const uint32_t *)nullptr, (double *)nullptr, (uint32_t *)nullptr, (int *)nullptr);
    std::vector<unsigned char> width((size_t)n_slices);
    rc = hipMemcpyAsync(width.data(), width_d, (size_t)n_slices, hipMemcpyDeviceToHost, s);
    if (rc == hipSuccess) rc = hipStreamSynchronize(s);
    (void)hipFree(width_d);
    if (rc != hipSuccess) return rc;
    std::vector<uint32_t> desc((size_t)n_slices);
    uint64_t off = 0;
    hist[0] = hist[1] = hist[2] = 0;
    for (int i = 0; i < n_slices; ++i) {
        if (off >= (1ull << 30)) return hipErrorInvalidValue;
        desc[i] = (uint32_t)(off << 2) | width[i];
        off += (uint64_t)ellg_words(W, width[i]);
        ++hist[width[i]];
    }
    *idx_words_out = (int64_t)(off * ELL_S);
    if ((rc = hipMalloc((void **)&A.ell_idx, std::max<size_t>(off * ELL_S, 1) * sizeof(uint32_t))) != hipSuccess) return rc;
    if ((rc = hipMemcpyAsync(A.ellg_desc, desc.data(), desc.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s)) != hipSuccess) return rc;
    if ((rc = hipStreamSynchronize(s)) != hipSuccess) return rc;         // desc is a host temporary
    hipLaunchKernelGGL(ellg_build_kernel<true>, dim3(n_slices), dim3(ELL_S), 0, s, A.n_rows, W, A.row_ptr, A.col, A.val,
                       (unsigned char *)nullptr, A.ellg_desc, A.ell_val, A.ell_idx, A.ellg_base);
    if ((rc = hipGetLastError()) != hipSuccess) return rc;
    return hipStreamSynchronize(s);
}

template <int W>
static void launch_ellg_w(const CsrDev &A, const SpmvArgs &a, int epi, int grid, int pb, int nt, hipStream_t s)
{
    dim3 g(grid), b(ITS_BLOCK);
#define ITS_LAUNCHG(E, D) \
    hipLaunchKernelGGL((spmv_ellg_kernel<W, E, D>), g, b, 0, s, A.n_rows, A.ell_blocks, pb, A.ell_val, A.ell_idx, A.ellg_desc, A.ellg_base, nt, a)
    switch (epi) {
    case EPI_STORE: if (a.dotv) ITS_LAUNCHG(EPI_STORE, true); else ITS_LAUNCHG(EPI_STORE, false); break;
    case EPI_RESIDUAL: ITS_LAUNCHG(EPI_RESIDUAL, false); break;
    case EPI_JACOBI_SHIPPED: ITS_LAUNCHG(EPI_JACOBI_SHIPPED, false); break;
    case EPI_JACOBI_REF: ITS_LAUNCHG(EPI_JACOBI_REF, false); break;
    case EPI_GS_RHS: ITS_LAUNCHG(EPI_GS_RHS, false); break;
    case EPI_JACOBI_FUSED: ITS_LAUNCHG(EPI_JACOBI_FUSED, false); break;
    case EPI_GS_FUSED: ITS_LAUNCHG(EPI_GS_FUSED, false); break;
    default: break;
    }
#undef ITS_LAUNCHG
}

void launch_spmv_ellg(const CsrDev &A, const SpmvArgs &a, int epi, int grid, int pb, int nt, hipStream_t s)
{
    switch (A.ell_w) {
    case 1: launch_ellg_w<1>(A, a, epi, grid, pb, nt, s); break;
    case 2: launch_ellg_w<2>(A, a, epi, grid, pb, nt, s); break;
    case 3: launch_ellg_w<3>(A, a, epi, grid, pb, nt, s); break;
    case 4: launch_ellg_w<4>(A, a, epi, grid, pb, nt, s); break;
    case 5: launch_ellg_w<5>(A, a, epi, grid, pb, nt, s); break;
    case 6: launch_ellg_w<6>(A, a, epi, grid, pb, nt, s); break;
    case 7: launch_ellg_w<7>(A, a, epi, grid, pb, nt, s); break;
    default: launch_ellg_w<8>(A, a, epi, grid, pb, nt, s); break;
    }
}

}  // namespace its
