// its_ellm.hip — adaptive-precision SpMV (ITS_FMT_MIXED) on a TIERED sliced-ELL image, for the matrices the narrow ELL image
// of its_ell.hip serves (rows of <= 8 entries, <= 15 distinct col - row: the 5-/7-point stencil class).
//
// The reference's USE_AP splits the matrix by magnitude thresholds into dp / sp / hp parts and hands each to a kernel of
// the absent USpMV library (main.cpp:171-184, config.mk:56-63).  Here the split is a property of the STORAGE of one image:
// slices of 128 rows, entry k (CRS order) of the rows of slice s in one contiguous run of 128 values — as in its_ell.hip —
// but every (slice, slot) run is stored in the narrowest type that all of its entries' tiers allow (its_ap.h): 1024, 512
// or 256 bytes, or nothing at all when the slot is empty in all 128 rows.  An entry whose own tier is narrower than its
// run's is ROUNDED to its own tier before it is stored (fp32 value in an fp64 run), so the product is that of the
// tier-rounded matrix whatever the layout — and since a lane still runs each row's fused multiply-adds from 0.0 in CRS
// order, it equals spmv_crs_cpu (kernels.hpp:380-409) on that matrix BIT FOR BIT: what ITS_FMT_MIXED gives up is stated
// entirely by its_ap::stored_value, nothing is lost to reordering.
//   tier word per slice: 2 bits per slot (0 fp64, 1 fp32, 2 fp16, 3 none), wave-uniform — a slice is a wave's work;
//   offset per slice: start of its runs in units of 256 bytes; index words: the nibble words of the fp64 image (shared).
// 7-pt 256^3 with AP_THRESHOLD_1 = 2 (diagonal 6 in fp64, the six -1 in fp32 / fp16): 8 + 6 x 4 + 4 = 36 bytes of matrix
// per row in ap[dp_sp], 8 + 6 x 2 + 4 = 24 in ap[dp_hp], against 60 for the fp64 image.
#include "its_kernels.h"
#include "its_epilogue.h"
#include "its_ap.h"
#include <cstdlib>
#include <vector>

namespace its {

typedef double ellm_v2d __attribute__((ext_vector_type(2)));
typedef float ellm_v2f __attribute__((ext_vector_type(2)));
typedef _Float16 ellm_v2h __attribute__((ext_vector_type(2)));

// pass 1, one workgroup of 128 lanes per slice: tier word of the slice (per slot the widest type any of its entries
// needs) and the slice's size in 256-byte units
__global__ __launch_bounds__(ELL_S) void ellm_tier_kernel(
    int n_rows, int W, const int *__restrict__ rp, const double *__restrict__ val, int mode, double t1, double t2,
    uint32_t *__restrict__ tier_word, uint32_t *__restrict__ units)
{
    __shared__ int s_min[ELL_MAX_W];
    const int s = blockIdx.x, jr = threadIdx.x, row = s * ELL_S + jr;
    if (jr < ELL_MAX_W) s_min[jr] = its_ap::TIER_NONE;
    __syncthreads();
    if (row < n_rows) {
        const int j0 = rp[row], len = rp[row + 1] - j0;
        for (int k = 0; k < W && k < len; ++k) atomicMin(&s_min[k], its_ap::tier(val[j0 + k], mode, t1, t2));
    }
    __syncthreads();
    if (jr == 0) {
        uint32_t wd = 0, u = 0;
        for (int k = 0; k < ELL_MAX_W; ++k) {
            const int t = k < W ? s_min[k] : its_ap::TIER_NONE;
            wd |= (uint32_t)t << (2 * k);
            u += t == its_ap::TIER_FP64 ? 4u : t == its_ap::TIER_FP32 ? 2u : t == its_ap::TIER_FP16 ? 1u : 0u;
        }
        tier_word[s] = wd; units[s] = u;
    }
}

// pass 2, one lane per padded row: its entries into the runs of its slice, each rounded to its own tier first
__global__ __launch_bounds__(ITS_BLOCK) void ellm_fill_kernel(
    int n_rows, int n_pad, int W, const int *__restrict__ rp, const double *__restrict__ val, int mode, double t1, double t2,
    const uint32_t *__restrict__ tier_word, const uint32_t *__restrict__ off, unsigned char *__restrict__ img)
{
    const int row = blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (row >= n_pad) return;
    const int s = row / ELL_S, jr = row % ELL_S;
    int j0 = 0, len = 0;
    if (row < n_rows) { j0 = rp[row]; len = rp[row + 1] - j0; }
    const uint32_t wd = tier_word[s];
    unsigned char *p = img + (size_t)off[s] * 256;
    for (int k = 0; k < W; ++k) {
        const int rt = (wd >> (2 * k)) & 3;                              // the run's type
        double v = 0.0;
        if (k < len) { const double a = val[j0 + k]; v = its_ap::stored_value(a, its_ap::tier(a, mode, t1, t2)); }
        if (rt == its_ap::TIER_FP64) { reinterpret_cast<double *>(p)[jr] = v; p += 1024; }
        else if (rt == its_ap::TIER_FP32) { reinterpret_cast<float *>(p)[jr] = (float)v; p += 512; }
        else if (rt == its_ap::TIER_FP16) { reinterpret_cast<_Float16 *>(p)[jr] = (_Float16)(float)v; p += 256; }
    }
}

// The product.  Work map, epilogues and reductions as spmv_ell_kernel (its_ell.hip); EPI_STORE (with or without the fused
// dot) and EPI_RESIDUAL, the forms ITS_FMT_MIXED serves.
template <int W, int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_ellm_kernel(
    int n_rows, int n_blocks, int pb, const unsigned char *__restrict__ img, const uint32_t *__restrict__ tier_word,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ idx, const int *__restrict__ dict, int nt_store, SpmvArgs a)
{
    __shared__ int s_dict[16];
    __shared__ double s_red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 16) s_dict[t] = t < 15 ? dict[t] : 0;
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
    double lane_sum = 0.0, lane_max = 0.0;
    // tier word and offset of a slice are what its value loads wait for: they are fetched one trip ahead
    const int n_sl = n_blocks * 4;
    auto slice_of = [&](int qq) { return qq < q_end ? ((qq / len) * per + b0 + qq % len) * 4 + w : n_sl; };
    uint32_t wd_n = 0, off_n = 0;
    { const int s0 = slice_of(j); if (s0 < n_sl) { wd_n = tier_word[s0]; off_n = off[s0]; } }
    for (int q = j; q < q_end; q += G >> 3) {
        const int blk = (q / len) * per + b0 + q % len;
        const uint32_t wd_c = wd_n, off_c = off_n;
        { const int s1 = slice_of(q + (G >> 3)); if (s1 < n_sl) { wd_n = tier_word[s1]; off_n = off[s1]; } }
        if (blk >= n_blocks) continue;
        const int sl = blk * 4 + w;
        const int row0 = sl * ELL_S + lane * 2;
        const bool any = row0 < n_rows, full = row0 + 1 < n_rows;
        const uint2 iw2 = *reinterpret_cast<const uint2 *>(idx + row0);
        const uint32_t iw[2] = {iw2.x, iw2.y};
        // the slice's tier word and offset are the same for the whole wave
        const uint32_t wd = __builtin_amdgcn_readfirstlane(wd_c);
        const unsigned char *p = img + (size_t)__builtin_amdgcn_readfirstlane(off_c) * 256;
        // Every run is read the same way — 16 bytes per lane at lane x (bytes per value pair), no branch on the run's type: a
        // lane of an fp32 / fp16 run reads past its own 8 / 4 bytes into its neighbours' (same cache lines; the image ends
        // with slack), and the decode below picks what is its own.  (A load per type inside wave-uniform branches made
        // every run wait for its own round trip — the compiler reuses the address registers as destinations across the
        // branches: 0.407 ms on the 7-pt 256^3 matrix against 0.235 ms of the all-fp64 image it is meant to undercut.)
        uint4 raw[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const uint32_t rt = (wd >> (2 * k)) & 3u;                     // wave-uniform
            const uint32_t lane_bytes = rt == its_ap::TIER_NONE ? 0u : (16u >> rt);
            __builtin_memcpy(&raw[k], p + lane * lane_bytes, 16);
            p += rt == its_ap::TIER_NONE ? 0u : (1024u >> rt);
        }
        double e0[2] = {0.0, 0.0};
        if (EPI == EPI_STORE && DOT) { if (full) { const ellm_v2d o2 = *reinterpret_cast<const ellm_v2d *>(a.dotv + row0); e0[0] = o2.x; e0[1] = o2.y; } else if (any) e0[0] = a.dotv[row0]; }
        if (EPI == EPI_RESIDUAL) { if (full) { const ellm_v2d o2 = *reinterpret_cast<const ellm_v2d *>(a.b + row0); e0[0] = o2.x; e0[1] = o2.y; } else if (any) e0[0] = a.b[row0]; }
        int c[W][2];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t nib = (iw[r] >> (4 * k)) & 15u;
                c[k][r] = nib == 15u ? 0 : row0 + r + s_dict[nib];
            }
        double xv[W][2];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r) xv[k][r] = a.x[c[k][r]];
        double v[W][2];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const uint32_t rt = (wd >> (2 * k)) & 3u;
            const double d0 = __hiloint2double((int)raw[k].y, (int)raw[k].x), d1 = __hiloint2double((int)raw[k].w, (int)raw[k].z);
            const double f0 = (double)__uint_as_float(raw[k].x), f1 = (double)__uint_as_float(raw[k].y);
            ellm_v2h hh; __builtin_memcpy(&hh, &raw[k].x, 4);
            const double h0 = (double)(float)hh.x, h1 = (double)(float)hh.y;
            v[k][0] = rt == its_ap::TIER_FP64 ? d0 : rt == its_ap::TIER_FP32 ? f0 : rt == its_ap::TIER_FP16 ? h0 : 0.0;
            v[k][1] = rt == its_ap::TIER_FP64 ? d1 : rt == its_ap::TIER_FP32 ? f1 : rt == its_ap::TIER_FP16 ? h1 : 0.0;
        }
        double outv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0, acc2 = 0.0, diag = 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k)
                if (((iw[r] >> (4 * k)) & 15u) != 15u) chain_step<EPI>(c[k][r], row0 + r, v[k][r], xv[k][r], acc, acc2, diag);
            outv[r] = 0.0;
            if (r == 0 ? any : full) outv[r] = row_finish<EPI, DOT>(acc, acc2, diag, e0[r], 0.0, 0.0, lane_sum, lane_max);
        }
        if (a.y) {
            if (full) {
                ellm_v2d o2; o2.x = outv[0]; o2.y = outv[1];
                if (nt_store) __builtin_nontemporal_store(o2, reinterpret_cast<ellm_v2d *>(a.y + row0));
                else *reinterpret_cast<ellm_v2d *>(a.y + row0) = o2;
            } else if (any) a.y[row0] = outv[0];
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

// Builds the tiered image of A (which must hold the narrow ELL image: its index words are shared).  Allocates
// A.ellm_tier / ellm_off / ellm_img; returns hipSuccess with A.ellm_img == nullptr when the matrix is not of that class.
hipError_t ellm_build(CsrDev &A, int mode, double t1, double t2, hipStream_t s)
{
    A.ellm_bytes = 0;
    if (!A.ell_val || A.ell_kind != 1 || A.n_rows < 1) return hipSuccess;
    const int W = A.ell_w;
    const size_t slices = (size_t)A.ell_blocks * (ELL_RB / ELL_S);
    uint32_t *units = nullptr;
    hipError_t e;
    if ((e = hipMalloc((void **)&A.ellm_tier, slices * sizeof(uint32_t) + 256)) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&A.ellm_off, slices * sizeof(uint32_t) + 256)) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&units, slices * sizeof(uint32_t) + 256)) != hipSuccess) return e;
    hipLaunchKernelGGL(ellm_tier_kernel, dim3((unsigned)slices), dim3(ELL_S), 0, s, A.n_rows, W, A.row_ptr, A.val, mode, t1, t2, A.ellm_tier, units);
    std::vector<uint32_t> hu(slices), ho(slices);
    if ((e = hipMemcpyAsync(hu.data(), units, slices * sizeof(uint32_t), hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess) { (void)hipFree(units); return e; }
    (void)hipFree(units);
    uint64_t tot = 0;
    for (size_t i = 0; i < slices; ++i) { ho[i] = (uint32_t)tot; tot += hu[i]; }
    if (tot >= (1ull << 32)) return hipErrorInvalidValue;                // (1 TB of values)
    // a lane reads up to 16 bytes per run: a little slack behind the last run
    if ((e = hipMalloc((void **)&A.ellm_img, (size_t)tot * 256 + 1024)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(A.ellm_off, ho.data(), slices * sizeof(uint32_t), hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    const int n_pad = A.ell_blocks * ELL_RB;
    hipLaunchKernelGGL(ellm_fill_kernel, dim3((n_pad + ITS_BLOCK - 1) / ITS_BLOCK), dim3(ITS_BLOCK), 0, s, A.n_rows, n_pad, W,
                       A.row_ptr, A.val, mode, t1, t2, A.ellm_tier, A.ellm_off, A.ellm_img);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;           // (ho goes out of scope)
    A.ellm_bytes = (int64_t)tot * 256;
    return hipSuccess;
}

template <int W>
static void launch_ellm_w(const CsrDev &A, const SpmvArgs &a, int epi, int grid, int pb, int nt, hipStream_t s)
{
    dim3 g(grid), b(ITS_BLOCK);
#define ITS_LAUNCHEM(E, D) \
    hipLaunchKernelGGL((spmv_ellm_kernel<W, E, D>), g, b, 0, s, A.n_rows, A.ell_blocks, pb, A.ellm_img, A.ellm_tier, A.ellm_off, A.ell_idx, A.off_dict, nt, a)
    if (epi == EPI_RESIDUAL) ITS_LAUNCHEM(EPI_RESIDUAL, false);
    else if (a.dotv) ITS_LAUNCHEM(EPI_STORE, true);
    else ITS_LAUNCHEM(EPI_STORE, false);
#undef ITS_LAUNCHEM
}

static inline bool al16m(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool spmv_ellm_eligible(const CsrDev &A, const SpmvArgs &a, int epi)
{
    return A.ellm_img && (epi == EPI_STORE || epi == EPI_RESIDUAL) && al16m(a.y) && al16m(a.b) && al16m(a.dotv);
}

void launch_spmv_ellm(const CsrDev &A, const SpmvArgs &a, int epi, hipStream_t s)
{
    static const int env_grid = []{ const char *e = getenv("ITS_SPMV_ELLM_GRID"); const int v = e ? atoi(e) : 0; return (v >= 8 && v <= ITS_NPART && !(v & 7)) ? v : 0; }();
    // 3 workgroups per CU (7-pt 256^3, threshold 2, ms for ap[dp_sp] / ap[dp_hp]: 512 -> 0.231 / 0.222, 768 -> 0.198 / 0.184,
    // 1024 -> 0.205 / 0.182, 1536 -> 0.219 / 0.197; the all-fp64 image: 0.235 at its best grid, 512): with fewer bytes per
    // row the trip is latency, and a third workgroup hides more of it than it costs the caches
    int grid = env_grid ? env_grid : 768;
    const int need = ((A.ell_blocks + 7) / 8) * 8;
    if (grid > need) grid = need < 8 ? 8 : need;
    const int pb = A.ell_pb, nt = 1;
    switch (A.ell_w) {
    case 1: launch_ellm_w<1>(A, a, epi, grid, pb, nt, s); break;
    case 2: launch_ellm_w<2>(A, a, epi, grid, pb, nt, s); break;
    case 3: launch_ellm_w<3>(A, a, epi, grid, pb, nt, s); break;
    case 4: launch_ellm_w<4>(A, a, epi, grid, pb, nt, s); break;
    case 5: launch_ellm_w<5>(A, a, epi, grid, pb, nt, s); break;
    case 6: launch_ellm_w<6>(A, a, epi, grid, pb, nt, s); break;
    case 7: launch_ellm_w<7>(A, a, epi, grid, pb, nt, s); break;
    default: launch_ellm_w<8>(A, a, epi, grid, pb, nt, s); break;
    }
}

}  // namespace its
