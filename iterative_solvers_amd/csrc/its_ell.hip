// its_ell.hip — the exact SpMV for short regular rows on a sliced, UNSORTED ELL image of the CRS matrix.
//
// Eligible (checked at upload, its_api.hip): longest row <= 8 entries, at most 15 distinct values of col - row
// (every 5-/7-point stencil, also after the halo renumbering of a row-block partition), little padding.
// Layout: slices of ELL_S = 128 rows; entry k (CRS order) of row j of slice s sits at
//     ell_val[(s*W + k)*128 + j],        W = longest row of the matrix,
// and one 32-bit word per row holds W 4-bit dictionary indices of col - row (nibble k = entry k, 0xF = no entry).
// A lane owns two adjacent rows and runs, for each, the in-order fused multiply-add chain from 0.0 of
// spmv_crs_cpu (kernels.hpp:380-409) — the same operations in the same order, so results are bit-identical to the
// CRS kernels — but every load is a fully coalesced 16-byte-per-lane wave access straight from HBM: no row_ptr
// stream, no LDS staging, no barrier.  Streamed bytes per row of a 7-pt matrix: 56 (values) + 4 (index word)
// + 8 (x) + 8 (y) = 76, against 84 + 4 + 16 = 104 of the CRS form and 63 + 4 + 16 = 83 of the
// 1-byte offset stream of spmv_stream4_kernel.
//
// Work map.  A workgroup trip is one row block of 512 rows (4 slices, one per wave); the grid is persistent
// (2 workgroups per CU measured best on MI355X: 512 -> 0.226 ms, 768 -> 0.245, 1024 -> 0.262, 2048 -> 0.266 ms on the
// 256^3 7-pt matrix, tools/elllab.hip).  With a dictionary whose largest |col - row| spans B >= 8 row blocks (the
// z-plane of a 3-D stencil) the blocks are dealt to the 8 XCDs by their position inside that period: XCD e owns
// the e-th eighth of EVERY period, so the far planes it gathers x from are the ones it has just streamed or streams
// next, and its 4 MB L2 holds them (a contiguous eighth per XCD re-fetches x through L2 three times: 1.17x
// over-fetch in the round-1 PMC record).  Otherwise XCD e owns the e-th contiguous eighth of the row blocks.
#include "its_kernels.h"
#include "its_epilogue.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace its {

typedef double ell_v2d __attribute__((ext_vector_type(2)));
constexpr int ELL_NONE = -2147483647 - 1;       // column of an empty slot of the byte-index form

template <int W, int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_ell_kernel(
    int n_rows, int n_blocks, int pb, const double *__restrict__ val, const uint32_t *__restrict__ idx,
    const int *__restrict__ dict, int nt_store, SpmvArgs a)
{
    __shared__ int s_dict[16];
    __shared__ double s_red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 16) s_dict[t] = t < 15 ? dict[t] : 0;
    __syncthreads();
    const int G = gridDim.x;
    const int e = blockIdx.x & 7, j = blockIdx.x >> 3;
    // list of row blocks of XCD e: position q <-> block (q / len) * per + b0 + q % len
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
        const int sl = blk * 4 + w;
        const int row0 = sl * ELL_S + lane * 2;
        const bool any = row0 < n_rows, full = row0 + 1 < n_rows;
        // index words first: the gathers depend on them, the values are only needed by the chain
        const uint2 iw2 = *reinterpret_cast<const uint2 *>(idx + row0);
        const uint32_t iw[2] = {iw2.x, iw2.y};
        const double *vp = val + (size_t)sl * W * ELL_S + lane * 2;
        double v[W][2];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const ell_v2d q2 = *reinterpret_cast<const ell_v2d *>(vp + (size_t)k * ELL_S);
            v[k][0] = q2.x; v[k][1] = q2.y;
        }
        // per-row operands of the epilogue (two adjacent rows: one 16-byte load each when both rows exist)
        double e0[2] = {0.0, 0.0}, e1[2] = {0.0, 0.0}, e2[2] = {0.0, 0.0};
#define ELL_LD2(dst, ptr)                                                                 \
        do {                                                                              \
            if (full) { const ell_v2d o2 = *reinterpret_cast<const ell_v2d *>((ptr) + row0); dst[0] = o2.x; dst[1] = o2.y; } \
            else if (any) dst[0] = (ptr)[row0];                                           \
        } while (0)
        if (EPI == EPI_STORE && DOT) ELL_LD2(e0, a.dotv);
        if (EPI == EPI_RESIDUAL || EPI == EPI_JACOBI_REF || EPI == EPI_GS_RHS || EpiKind<EPI>::fused) ELL_LD2(e0, a.b);
        if (EPI == EPI_JACOBI_SHIPPED) { ELL_LD2(e0, a.D); ELL_LD2(e1, a.x); ELL_LD2(e2, a.b); }
        if (EPI == EPI_SWEEP) { ELL_LD2(e0, a.b); ELL_LD2(e1, a.D); }
        const bool pre = EPI == EPI_GS_FUSED && a.y2 && row0 >= a.y2_lo && row0 < a.y2_hi;   // rows of the entry-free first level
        if (pre) ELL_LD2(e1, a.D);
        double a0[2] = {0.0, 0.0};                               // where the full chain of the fused Gauss-Seidel pass starts (SpmvArgs::acc0)
        if (EPI == EPI_GS_FUSED && a.acc0 && !pre) ELL_LD2(a0, a.acc0);
#undef ELL_LD2
        int c[W][2];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t nib = (iw[r] >> (4 * k)) & 15u;
                c[k][r] = nib == 15u ? 0 : row0 + r + s_dict[nib];        // no entry: gather x[0], never used
            }
        double xv[W][2];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r) xv[k][r] = a.x[c[k][r]];
        double outv[2], accs[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0, acc2 = a0[r], diag = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k)
                if (((iw[r] >> (4 * k)) & 15u) != 15u) chain_step<EPI>(c[k][r], row0 + r, v[k][r], xv[k][r], acc, acc2, diag);
            accs[r] = acc;
            outv[r] = 0.0;
            if (r == 0 ? any : full) outv[r] = row_finish<EPI, DOT>(acc, acc2, diag, e0[r], e1[r], e2[r], lane_sum, lane_max);
        }
        if (pre) {                                              // next iterate of these rows: (b - U x) / D, as sweep_empty_kernel
            if (full) { ell_v2d o2; o2.x = (outv[0] - 0.0) / e1[0]; o2.y = (outv[1] - 0.0) / e1[1]; *reinterpret_cast<ell_v2d *>(a.y2 + row0) = o2; }
            else if (any) a.y2[row0] = (outv[0] - 0.0) / e1[0];
        } else
        if (a.y) {                                              // y == nullptr: only the norm is wanted
            if (full) {
                ell_v2d o2; o2.x = outv[0]; o2.y = outv[1];
                if (nt_store) __builtin_nontemporal_store(o2, reinterpret_cast<ell_v2d *>(a.y + row0));
                else *reinterpret_cast<ell_v2d *>(a.y + row0) = o2;
            } else if (any) a.y[row0] = outv[0];
        }
        if (EPI == EPI_SWEEP && a.acc_out) {                    // the sweep's chain sums (SpmvArgs::acc_out)
            if (full) { ell_v2d o2; o2.x = accs[0]; o2.y = accs[1]; *reinterpret_cast<ell_v2d *>(a.acc_out + row0) = o2; }
            else if (any) a.acc_out[row0] = accs[0];
        }
    }
    if (DOT || EpiKind<EPI>::norm) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) {
            a.part_sum[blockIdx.x] = bs;
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
// The same layout for WIDER rows (9..32 entries, up to 255 distinct col - row: the 27-point stencils): byte
// dictionary indices, four to a 32-bit word, stored like the values — word q of row j of slice s at
// idx[(s*WQ + q)*128 + j], WQ = ceil(W / 4), 0xFF = no entry.  A row is walked in batches of 8 slots (two index
// words): the batch's 8 value loads and 16 gathers are in flight together, the chain still runs in CRS order.
// NB = batches per row (compile time); W itself is a run-time argument, slots beyond it are not loaded.
// --------------------------------------------------------------------------------------
template <int NB, int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_ellw_kernel(
    int n_rows, int n_blocks, int pb, int W, const double *__restrict__ val, const uint32_t *__restrict__ idx,
    const int *__restrict__ dict, int nt_store, SpmvArgs a)
{
    __shared__ int s_dict[256];
    __shared__ double s_red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    s_dict[t] = t < 255 ? dict[t] : 0;
    __syncthreads();
    const int G = gridDim.x;
    const int e = blockIdx.x & 7, j = blockIdx.x >> 3;
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
    const int WQ = (W + 3) >> 2;
    double lane_sum = 0.0, lane_max = 0.0;
    for (int q = j; q < q_end; q += G >> 3) {
        const int blk = (q / len) * per + b0 + q % len;
        if (blk >= n_blocks) continue;
        const int sl = blk * 4 + w;
        const int row0 = sl * ELL_S + lane * 2;
        const bool any = row0 < n_rows, full = row0 + 1 < n_rows;
        const uint32_t *ip = idx + (size_t)sl * WQ * ELL_S + lane * 2;
        const double *vp = val + (size_t)sl * W * ELL_S + lane * 2;
        // all index words of the two rows first (the gathers depend on them)
        uint32_t iw[2 * NB][2];
#pragma unroll
        for (int qq = 0; qq < 2 * NB; ++qq) {
            iw[qq][0] = 0xffffffffu; iw[qq][1] = 0xffffffffu;
            if (qq < WQ) { const uint2 i2 = *reinterpret_cast<const uint2 *>(ip + (size_t)qq * ELL_S); iw[qq][0] = i2.x; iw[qq][1] = i2.y; }
        }
        double e0[2] = {0.0, 0.0}, e1[2] = {0.0, 0.0}, e2[2] = {0.0, 0.0};
#define ELL_LD2(dst, ptr)                                                                 \
        do {                                                                              \
            if (full) { const ell_v2d o2 = *reinterpret_cast<const ell_v2d *>((ptr) + row0); dst[0] = o2.x; dst[1] = o2.y; } \
            else if (any) dst[0] = (ptr)[row0];                                           \
        } while (0)
        if (EPI == EPI_STORE && DOT) ELL_LD2(e0, a.dotv);
        if (EPI == EPI_RESIDUAL || EPI == EPI_JACOBI_REF || EPI == EPI_GS_RHS || EpiKind<EPI>::fused) ELL_LD2(e0, a.b);
        if (EPI == EPI_JACOBI_SHIPPED) { ELL_LD2(e0, a.D); ELL_LD2(e1, a.x); ELL_LD2(e2, a.b); }
        if (EPI == EPI_SWEEP) { ELL_LD2(e0, a.b); ELL_LD2(e1, a.D); }
        const bool pre = EPI == EPI_GS_FUSED && a.y2 && row0 >= a.y2_lo && row0 < a.y2_hi;   // rows of the entry-free first level
        if (pre) ELL_LD2(e1, a.D);
        double a0[2] = {0.0, 0.0};                               // where the full chain of the fused Gauss-Seidel pass starts (SpmvArgs::acc0)
        if (EPI == EPI_GS_FUSED && a.acc0 && !pre) ELL_LD2(a0, a.acc0);
#undef ELL_LD2
        double acc[2] = {0.0, 0.0}, acc2[2] = {a0[0], a0[1]}, diag[2] = {0.0, 0.0};
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            double v[8][2], xv[8][2];
            int c[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 8 * bt + u;
                v[u][0] = 0.0; v[u][1] = 0.0;
                if (k < W) { const ell_v2d q2 = *reinterpret_cast<const ell_v2d *>(vp + (size_t)k * ELL_S); v[u][0] = q2.x; v[u][1] = q2.y; }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t by = (iw[2 * bt + (u >> 2)][r] >> (8 * (u & 3))) & 255u;
                    c[u][r] = by == 255u ? ELL_NONE : row0 + r + s_dict[by];     // (a sweep view's columns may be negative: x is shifted)
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < 2; ++r) xv[u][r] = a.x[c[u][r] == ELL_NONE ? 0 : c[u][r]];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c[u][r] != ELL_NONE) chain_step<EPI>(c[u][r], row0 + r, v[u][r], xv[u][r], acc[r], acc2[r], diag[r]);
        }
        double outv[2] = {0.0, 0.0};
        if (any) outv[0] = row_finish<EPI, DOT>(acc[0], acc2[0], diag[0], e0[0], e1[0], e2[0], lane_sum, lane_max);
        if (full) outv[1] = row_finish<EPI, DOT>(acc[1], acc2[1], diag[1], e0[1], e1[1], e2[1], lane_sum, lane_max);
        if (pre) {
            if (full) { ell_v2d o2; o2.x = (outv[0] - 0.0) / e1[0]; o2.y = (outv[1] - 0.0) / e1[1]; *reinterpret_cast<ell_v2d *>(a.y2 + row0) = o2; }
            else if (any) a.y2[row0] = (outv[0] - 0.0) / e1[0];
        } else
        if (a.y) {
            if (full) {
                ell_v2d o2; o2.x = outv[0]; o2.y = outv[1];
                if (nt_store) __builtin_nontemporal_store(o2, reinterpret_cast<ell_v2d *>(a.y + row0));
                else *reinterpret_cast<ell_v2d *>(a.y + row0) = o2;
            } else if (any) a.y[row0] = outv[0];
        }
        if (EPI == EPI_SWEEP && a.acc_out) {                    // the sweep's chain sums (SpmvArgs::acc_out)
            if (full) { ell_v2d o2; o2.x = acc[0]; o2.y = acc[1]; *reinterpret_cast<ell_v2d *>(a.acc_out + row0) = o2; }
            else if (any) a.acc_out[row0] = acc[0];
        }
    }
    if (DOT || EpiKind<EPI>::norm) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) {
            a.part_sum[blockIdx.x] = bs;
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

// wide image: values as in ell_build_kernel, byte indices packed four to a word
__global__ __launch_bounds__(ITS_BLOCK) void ellw_build_kernel(
    int n_rows, int n_pad, int W, const int *__restrict__ rp, const unsigned char *__restrict__ cidx,
    const double *__restrict__ val, double *__restrict__ ell_val, uint32_t *__restrict__ ell_idx)
{
    const int row = blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (row >= n_pad) return;
    const int s = row / ELL_S, jr = row % ELL_S;
    const int WQ = (W + 3) >> 2;
    int j0 = 0, len = 0;
    if (row < n_rows) { j0 = rp[row]; len = rp[row + 1] - j0; }
    for (int qq = 0; qq < WQ; ++qq) {
        uint32_t wd = 0xffffffffu;
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * qq + u;
            if (k < len) wd = (wd & ~(255u << (8 * u))) | ((uint32_t)cidx[j0 + k] << (8 * u));   // the offset stream's own indices
        }
        ell_idx[((size_t)s * WQ + qq) * ELL_S + jr] = wd;
    }
    for (int k = 0; k < W; ++k) ell_val[((size_t)s * W + k) * ELL_S + jr] = k < len ? val[j0 + k] : 0.0;
}

// ELL image from the device-resident CRS arrays: one lane per (padded) row
__global__ __launch_bounds__(ITS_BLOCK) void ell_build_kernel(
    int n_rows, int n_pad, int W, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
    const int *__restrict__ dict, int n_off, double *__restrict__ ell_val, uint32_t *__restrict__ ell_idx, int row_base)
{
    __shared__ int s_dict[16];
    if (threadIdx.x < 16) s_dict[threadIdx.x] = threadIdx.x < n_off ? dict[threadIdx.x] : 0;
    __syncthreads();
    const int row = blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (row >= n_pad) return;
    const int s = row / ELL_S, jr = row % ELL_S;
    int j0 = 0, len = 0;
    if (row < n_rows) { j0 = rp[row]; len = rp[row + 1] - j0; }
    uint32_t wd = 0xffffffffu;
    for (int k = 0; k < W; ++k) {
        double v = 0.0;
        if (k < len) {
            v = val[j0 + k];
            const int off = col[j0 + k] - (row + row_base);      // row_base: first row of a view of a larger matrix
            uint32_t d = 0;
            while (d + 1 < (uint32_t)n_off && s_dict[d] != off) ++d;
            wd = (wd & ~(15u << (4 * k))) | (d << (4 * k));
        }
        ell_val[((size_t)s * W + k) * ELL_S + jr] = v;
    }
    ell_idx[row] = wd;
}

void launch_ell_build(const CsrDev &A, hipStream_t s, int row_base)
{
    const int n_pad = A.ell_blocks * ELL_RB;
    if (A.ell_kind == 2) {
        hipLaunchKernelGGL(ellw_build_kernel, dim3((n_pad + ITS_BLOCK - 1) / ITS_BLOCK), dim3(ITS_BLOCK), 0, s, A.n_rows, n_pad,
                           A.ell_w, A.row_ptr, A.cidx, A.val, A.ell_val, A.ell_idx);
        return;
    }
    hipLaunchKernelGGL(ell_build_kernel, dim3((n_pad + ITS_BLOCK - 1) / ITS_BLOCK), dim3(ITS_BLOCK), 0, s, A.n_rows, n_pad,
                       A.ell_w, A.row_ptr, A.col, A.val, A.off_dict, A.n_off, A.ell_val, A.ell_idx, row_base);
}

// Period (in row blocks) of the work map for a matrix whose col - row values are `dict`: the P for which (nearly) every
// offset is close to a multiple of P rows, so that the rows an XCD owns — the same eighth of every period — gather x
// from what that XCD streams anyway.  Natural-order 3-D stencils: the plane (7-pt 256^3: 128 blocks).  The same stencil
// in red-black order has offsets n/2 + {0, +-1, +-line/2, +-plane/2}: the largest one says nothing, the period is the
// half plane (64 blocks; GS trip at 256^3 0.509 -> 0.47 ms).  Candidates are the offsets and their pairwise differences,
// rounded to whole blocks; a candidate may leave a quarter of the offsets unexplained (the halo columns of a row-block
// partition); fewest unexplained, then smallest residue, then shortest period wins.  0 = no period (contiguous eighths).
int ell_choose_period(const int *dict, int n_off)
{
    if (n_off < 1) return 0;
    std::vector<int64_t> mag;
    for (int i = 0; i < n_off; ++i) mag.push_back(std::llabs((long long)dict[i]));
    std::vector<int> cand;
    auto add = [&](int64_t rows) { const int64_t pb = (rows + ELL_RB / 2) / ELL_RB; if (pb >= 8 && pb < (1 << 24)) cand.push_back((int)pb); };
    for (int i = 0; i < n_off; ++i) { add(mag[i]); for (int j = 0; j < i; ++j) add(std::llabs(mag[i] - mag[j])); }
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    int best = 0, best_bad = n_off + 1; int64_t best_res = 0;
    for (int pb : cand) {
        const int64_t P = (int64_t)pb * ELL_RB;
        int bad = 0; int64_t res = 0;
        for (int i = 0; i < n_off; ++i) {
            const int64_t m = mag[i] % P, r = std::min(m, P - m);
            if (r > P / 8) ++bad; else res = std::max(res, r);
        }
        if (bad > n_off / 4) continue;
        if (bad < best_bad || (bad == best_bad && res < best_res)) { best = pb; best_bad = bad; best_res = res; }
    }
    return best;
}

static int ell_env(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static int g_ell_on = ell_env("ITS_SPMV_ELL", 1);
static int g_ell_grid = ell_env("ITS_SPMV_ELL_GRID", 512);
static int g_ell_nt = ell_env("ITS_SPMV_ELL_NT", 1);
static int g_ellw_grid = ell_env("ITS_SPMV_ELLW_GRID", 768);      // 3 workgroups per CU: the byte-index kernel holds 3 waves per SIMD
static int g_ell_wide = ell_env("ITS_SPMV_ELL_WIDE", 1);      // 0 = rows of 9..32 entries stay on the CRS stream kernels
static int g_ell_map = ell_env("ITS_SPMV_ELL_MAP", 1);       // 0 = contiguous eighth per XCD even for far-banded matrices
static int g_ell_dict = ell_env("ITS_SPMV_ELL_DICT", 1);     // 0 = uploads build no dictionary image: eligible rows get the explicit-column one
int ell_dict_forms_enabled() { return g_ell_dict; }

int ell_tune(const char *name, int value)
{
    if (!strcmp(name, "spmv_ell")) { if (value < 0 || value > 1) return -1; g_ell_on = value; return 0; }
    if (!strcmp(name, "spmv_ell_grid")) { if (value < 8 || value > ITS_NPART || (value & 7)) return -1; g_ell_grid = value; return 0; }
    if (!strcmp(name, "spmv_ell_nt")) { if (value < 0 || value > 1) return -1; g_ell_nt = value; return 0; }
    if (!strcmp(name, "spmv_ellw_grid")) { if (value < 8 || value > ITS_NPART || (value & 7)) return -1; g_ellw_grid = value; return 0; }
    if (!strcmp(name, "spmv_ell_wide")) { if (value < 0 || value > 1) return -1; g_ell_wide = value; return 0; }
    if (!strcmp(name, "spmv_ell_dict")) { if (value < 0 || value > 1) return -1; g_ell_dict = value; return 0; }
    if (!strcmp(name, "spmv_ell_map")) { if (value < 0 || value > 1) return -1; g_ell_map = value; return 0; }
    return -2;
}

static inline bool al16p(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool spmv_ell_eligible(const CsrDev &A, const SpmvArgs &a, int epi)
{
    if (!g_ell_on || !A.ell_val || epi == EPI_SWEEP || spmv_valdict_selected(A)) return false;
    if (A.ell_kind == 2 && !g_ell_wide) return false;
    if (A.ell_kind == 3 && a.y2) return false;                   // (the first-level store of EPI_GS_FUSED is not in that kernel)
    // two rows per lane move as 16-byte accesses: every per-row vector must be 16-byte aligned (solver vectors are)
    return al16p(a.y) && al16p(a.b) && al16p(a.D) && al16p(a.dotv) && (epi != EPI_JACOBI_SHIPPED || al16p(a.x));
}

template <int W>
static void launch_ell_w(const CsrDev &A, const SpmvArgs &a, int epi, int grid, int pb, int nt, hipStream_t s)
{
    dim3 g(grid), b(ITS_BLOCK);
#define ITS_LAUNCHE(E, D) \
    hipLaunchKernelGGL((spmv_ell_kernel<W, E, D>), g, b, 0, s, A.n_rows, A.ell_blocks, pb, A.ell_val, A.ell_idx, A.off_dict, nt, a)
    switch (epi) {
    case EPI_STORE: if (a.dotv) ITS_LAUNCHE(EPI_STORE, true); else ITS_LAUNCHE(EPI_STORE, false); break;
    case EPI_RESIDUAL: ITS_LAUNCHE(EPI_RESIDUAL, false); break;
    case EPI_JACOBI_SHIPPED: ITS_LAUNCHE(EPI_JACOBI_SHIPPED, false); break;
    case EPI_JACOBI_REF: ITS_LAUNCHE(EPI_JACOBI_REF, false); break;
    case EPI_GS_RHS: ITS_LAUNCHE(EPI_GS_RHS, false); break;
    case EPI_JACOBI_FUSED: ITS_LAUNCHE(EPI_JACOBI_FUSED, false); break;
    case EPI_GS_FUSED: ITS_LAUNCHE(EPI_GS_FUSED, false); break;
    case EPI_SWEEP: ITS_LAUNCHE(EPI_SWEEP, false); break;
    default: break;
    }
#undef ITS_LAUNCHE
}

template <int NB>
static void launch_ellw_nb(const CsrDev &A, const SpmvArgs &a, int epi, int grid, int pb, int nt, hipStream_t s)
{
    dim3 g(grid), b(ITS_BLOCK);
#define ITS_LAUNCHW(E, D) \
    hipLaunchKernelGGL((spmv_ellw_kernel<NB, E, D>), g, b, 0, s, A.n_rows, A.ell_blocks, pb, A.ell_w, A.ell_val, A.ell_idx, A.off_dict, nt, a)
    switch (epi) {
    case EPI_STORE: if (a.dotv) ITS_LAUNCHW(EPI_STORE, true); else ITS_LAUNCHW(EPI_STORE, false); break;
    case EPI_RESIDUAL: ITS_LAUNCHW(EPI_RESIDUAL, false); break;
    case EPI_JACOBI_SHIPPED: ITS_LAUNCHW(EPI_JACOBI_SHIPPED, false); break;
    case EPI_JACOBI_REF: ITS_LAUNCHW(EPI_JACOBI_REF, false); break;
    case EPI_GS_RHS: ITS_LAUNCHW(EPI_GS_RHS, false); break;
    case EPI_JACOBI_FUSED: ITS_LAUNCHW(EPI_JACOBI_FUSED, false); break;
    case EPI_GS_FUSED: ITS_LAUNCHW(EPI_GS_FUSED, false); break;
    case EPI_SWEEP: ITS_LAUNCHW(EPI_SWEEP, false); break;
    default: break;
    }
#undef ITS_LAUNCHW
}

void launch_spmv_ell(const CsrDev &A, const SpmvArgs &a_in, int epi, hipStream_t s)
{
    SpmvArgs a = a_in;
    a.part_accum = 0;
    if (A.ell_kind == 2) {
        int grid = g_ellw_grid;
        const int need = ((A.ell_blocks + 7) / 8) * 8;
        if (grid > need) grid = need < 8 ? 8 : need;
        const int pb = g_ell_map ? A.ell_pb : 0;
        const int nb = (A.ell_w + 7) / 8;
        if (nb <= 2) launch_ellw_nb<2>(A, a, epi, grid, pb, g_ell_nt, s);
        else if (nb == 3) launch_ellw_nb<3>(A, a, epi, grid, pb, g_ell_nt, s);
        else launch_ellw_nb<4>(A, a, epi, grid, pb, g_ell_nt, s);
        return;
    }
    int grid = g_ell_grid;
    // small matrices: no more workgroups than row blocks (multiple of 8: the XCD map)
    const int need = ((A.ell_blocks + 7) / 8) * 8;
    if (grid > need) grid = need < 8 ? 8 : need;
    const int pb = g_ell_map ? A.ell_pb : 0;
    // the result of a CG / Jacobi product is read by the very next kernel: plain store; non-temporal otherwise
    const int nt = g_ell_nt;
    if (A.ell_kind == 3) { launch_spmv_ellg(A, a, epi, grid, pb, nt, s); return; }
    switch (A.ell_w) {
    case 1: launch_ell_w<1>(A, a, epi, grid, pb, nt, s); break;
    case 2: launch_ell_w<2>(A, a, epi, grid, pb, nt, s); break;
    case 3: launch_ell_w<3>(A, a, epi, grid, pb, nt, s); break;
    case 4: launch_ell_w<4>(A, a, epi, grid, pb, nt, s); break;
    case 5: launch_ell_w<5>(A, a, epi, grid, pb, nt, s); break;
    case 6: launch_ell_w<6>(A, a, epi, grid, pb, nt, s); break;
    case 7: launch_ell_w<7>(A, a, epi, grid, pb, nt, s); break;
    default: launch_ell_w<8>(A, a, epi, grid, pb, nt, s); break;
    }
}

// One contiguous level (a colour of a multicolour ordering) of crs_L swept on its own ELL image: rows [lo, lo + cnt) of L
// as a matrix of their own whose vectors are the level's slices of x, rhs and D.  Column offsets are those of L, so the
// gathers go through x + lo (columns relative to the view may be negative: everything a level reads lies before it).
void launch_sweep_ell_view(const CsrDev &L, const SweepEll &v, double *x, const double *rhs, const double *D, hipStream_t s, double *acc_out)
{
    CsrDev V;
    V.n_rows = v.cnt; V.n_cols = L.n_cols; V.off_dict = L.off_dict; V.n_off = L.n_off;
    V.ell_val = v.val; V.ell_idx = v.idx; V.ell_w = v.w; V.ell_blocks = v.blocks; V.ell_kind = v.kind; V.ell_pb = v.pb;
    SpmvArgs a; a.x = x + v.lo; a.y = x + v.lo; a.b = rhs + v.lo; a.D = D + v.lo;
    if (acc_out) a.acc_out = acc_out + v.lo;
    launch_spmv_ell(V, a, EPI_SWEEP, s);
}

}  // namespace its
