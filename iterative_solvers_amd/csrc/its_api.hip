// its_api.hip — the extern "C" boundary of libits_hip.so (declared in include/its_hip.h).
// Host-side state machine that mirrors Solver<VT> (reference solver.hpp) with every vector
// resident in HBM; the only per-trip host traffic is one 8-byte scalar (GMRES: one H column).
#include "its_handle.h"
#include "its_hostplan.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

// 0 = watch the pinned word (default), 1 = write the word the same way but wait with hipStreamSynchronize
// (ITS_READBACK_SYNC), 2 = device slot + 8-byte copy + synchronize (ITS_READBACK_COPY).  its_set_option "readback"
// overrides the environment (tests exercise all three against each other).
static int g_readback_mode = getenv("ITS_READBACK_COPY") ? 2 : getenv("ITS_READBACK_SYNC") ? 1 : 0;
static inline int readback_mode() { return g_readback_mode; }

// ITS_TIMING=1: wall time of the set-up phases on stderr (uploads, plans, schedules): what SURVEY.md §8f.2 asks to
// be kept off the serial host path
struct PhaseTimer {
    bool on; std::chrono::steady_clock::time_point t0; const char *ctx;
    explicit PhaseTimer(const char *c) : on(getenv("ITS_TIMING") != nullptr), t0(std::chrono::steady_clock::now()), ctx(c) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[its timing] %s / %s: %.1f ms\n", ctx, what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

int build_panel_image(its_handle *h, CsrDev &A, const int *host_rp);    // its_panel.hip

static void free_lines(its_handle *h)
{
    dfree(h->lines.img); dfree(h->lines.mimg); dfree(h->lines.rimg); dfree(h->lines.ximg); dfree(h->lines.xedge); dfree(h->lines.sbase); dfree(h->lines.tabs); dfree(h->lines.counter);
    h->lines = LinesPlan(); h->lines_ready = false;
}

static void free_level_ell(its_handle *h)
{
    for (auto &v : h->level_ell) { dfree(v.val); dfree(v.idx); }
    h->level_ell.clear();
}

static void free_csr(CsrDev &A)
{
    dfree(A.row_ptr); dfree(A.col); dfree(A.val); dfree(A.blk_desc); dfree(A.blk_desc_big); dfree(A.blk_desc_short); dfree(A.irr_long_rows);
    dfree(A.sell_chunk_ptr); dfree(A.sell_chunk_len); dfree(A.sell_col); dfree(A.sell_perm); dfree(A.sell_val); dfree(A.sell_grp_start); dfree(A.cidx); dfree(A.off_dict); dfree(A.vidx); dfree(A.val_dict);
    dfree(A.ell_val); dfree(A.ell_idx); dfree(A.ellg_desc); dfree(A.ellg_base);
    dfree(A.pan_key); dfree(A.pan_val); dfree(A.pan_desc); dfree(A.pan_tile_ptr); dfree(A.pan_tile_row); dfree(A.pan_long); dfree(A.pan_diag); dfree(A.pan_long_rows);
    dfree(A.mix_rp_d); dfree(A.mix_col_d); dfree(A.mix_val_d); dfree(A.mix_rp_s); dfree(A.mix_col_s); dfree(A.mix_val_s);
    dfree(A.mix_rp_h); dfree(A.mix_col_h); dfree(A.mix_val_h); dfree(A.ellm_img); dfree(A.ellm_tier); dfree(A.ellm_off);
    A = CsrDev();
}

// ---------------------------------------------------------------------------------------
// Scratch of the entry point that allocates nothing (its_dot_noalloc, below): the ITS_NPART per-workgroup partial
// sums of one reduction, 16 KB per GPU, held from module load until the process ends.  `scratch` is a host-side
// stand-in for the symbol, NOT a device address: host code only ever uses the per-device address that
// its_create() obtains with hipGetSymbolAddress (each GPU has its own copy of the array at its own address).
static __device__ __align__(256) double scratch[ITS_NPART];
constexpr int ITS_MAX_DEVICES = 64;
static double *g_scratch_ptr[ITS_MAX_DEVICES] = {};

// ---------------------------------------------------------------------------------------
extern "C" const char *its_version(void) { return "its_hip 0.1 (gfx950)"; }
extern "C" const char *its_last_error(const its_handle *h) { return h ? h->err.c_str() : "null handle"; }

// pinned read-back buffer with room for a restart length of m (its_handle.h: layout); grows, never shrinks
static int size_pinned(its_handle *h, int m)
{
    const size_t region = (size_t)m + 2;
    const size_t need = std::max<size_t>(4096, PIN_REGIONS + 3 * region);
    if (need > h->pinned_cap) {
        if (h->pinned) (void)hipHostFree(h->pinned);
        h->pinned = nullptr; h->pinned_cap = 0;
        HIPCHK(h, hipHostMalloc((void **)&h->pinned, need * sizeof(double), hipHostMallocDefault));
        h->pinned_cap = need;
    }
    h->pin_copy = PIN_REGIONS; h->pin_y = PIN_REGIONS + region; h->pin_h = PIN_REGIONS + 2 * region;
    return ITS_OK;
}

// ITS_DEBUG_POISON_LDS=1 (test aid): every compute entry point first fills the LDS of all CUs with a NaN pattern, so that
// a kernel reading LDS it has not written shows up as a wrong result instead of depending on what ran before it
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned long long *sink)
{
    __shared__ unsigned long long s_p[8192];                   // 64 KB: two workgroups per CU
    for (int i = threadIdx.x; i < 8192; i += 256) s_p[i] = 0x7ff8dead0badf11dull;     // (the line sweep's pending pattern, its_lines.hip)
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) { }                     // 20 us: long enough for the launch to cover every CU
    if (sink && s_p[(threadIdx.x * 37) & 8191] != 0x7ff8dead0badf11dull) *sink = 1;
}
static void launch_lines_poison_lds(hipStream_t s) { hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(256), 0, s, (unsigned long long *)nullptr); }
static void debug_poison_lds(its_handle *h)
{
    static const bool on = getenv("ITS_DEBUG_POISON_LDS") && atoi(getenv("ITS_DEBUG_POISON_LDS")) != 0;
    if (on && h) launch_lines_poison_lds(h->stream);
}

extern "C" int its_device_count(void)
{
    int ndev = 0;
    return hipGetDeviceCount(&ndev) == hipSuccess ? ndev : 0;
}

extern "C" int its_create(its_handle **out, int device)
{
    if (!out) return ITS_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ITS_EHIP;
    its_handle *h = new its_handle;
    if (const char *e = getenv("ITS_DIST_OVERLAP")) h->dist_overlap = atoi(e) != 0;
    // (for hosts that take no options of their own — its_solve: the same switches as its_set_option)
    if (const char *e = getenv("ITS_GMRES_PERSIST")) h->gmres_persist = atoi(e) != 0;
    if (const char *e = getenv("ITS_GMRES_PERSIST_MIN_N")) h->gmres_persist_min_n = std::max(1, atoi(e));
    if (device >= 0) { if (hipSetDevice(device) != hipSuccess) { delete h; return ITS_EHIP; } }
    (void)hipGetDevice(&h->device);
    // once per device, with that device current: the address of THIS device's copy of the no-allocation scratch
    if (h->device >= 0 && h->device < ITS_MAX_DEVICES && !g_scratch_ptr[h->device]) {
        double *ptr = nullptr;
        if (hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) { delete h; return ITS_EHIP; }
        g_scratch_ptr[h->device] = ptr;
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return ITS_EHIP; }
    (void)hipEventCreate(&h->ev0); (void)hipEventCreate(&h->ev1);
    for (int i = 0; i < 4; ++i) {
        if (dalloc(h, &h->part[i], ITS_NPART) != ITS_OK) { delete h; return ITS_EHIP; }
        (void)hipMemsetAsync(h->part[i], 0, ITS_NPART * sizeof(double), h->stream);
    }
    if (dalloc(h, &h->sc, SC_COUNT) != ITS_OK) { delete h; return ITS_EHIP; }
    (void)hipMemsetAsync(h->sc, 0, SC_COUNT * sizeof(double), h->stream);
    if (size_pinned(h, 0) != ITS_OK) { delete h; return ITS_EHIP; }
    (void)hipStreamSynchronize(h->stream);
    *out = h;
    return ITS_OK;
}

void its_dist_destroy(its_handle *h);   // its_dist.hip

extern "C" int its_destroy(its_handle *h)
{
    if (!h) return ITS_EINVAL;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    its_dist_destroy(h);
    for (auto &A : h->mat) free_csr(A);
    free_lines(h);
    dfree(h->D); dfree(h->b); dfree(h->r); dfree(h->tmp); dfree(h->x_a); dfree(h->x_b); dfree(h->x);
    dfree(h->p); dfree(h->V); dfree(h->hcol_dev); dfree(h->y_dev); dfree(h->gm_w2);
    dfree(h->pcg_w); dfree(h->pcg_s); dfree(h->pcg_z); dfree(h->pcg_pair[0]); dfree(h->pcg_pair[1]); dfree(h->gram_part); dfree(h->gram_h); dfree(h->mgs_slots);
    for (auto &e : h->prof_ev) (void)hipEventDestroy(e);
    for (auto &p : h->part) dfree(p);
    dfree(h->sc); dfree(h->res_dev); dfree(h->level_rows); dfree(h->level_ptr_dev); dfree(h->sweep_desc); free_level_ell(h);
    dfree(h->sw_col); dfree(h->sw_val); dfree(h->sw_D);
    if (h->pinned) (void)hipHostFree(h->pinned);
    (void)hipEventDestroy(h->ev0); (void)hipEventDestroy(h->ev1);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return ITS_OK;
}

// ---------------------------------------------------------------------------------------
// host-side plans
// ---------------------------------------------------------------------------------------
// (solo_len > 0: a row of more than solo_len entries is a block of its own)
static void build_stream_plan(int n, const int *rp, std::vector<int> &blk, int max_rows = ITS_BLOCK,
                              int max_nnz = SPMV_NNZ_MAX, int solo_len = 0)
{
    blk.clear(); blk.push_back(0);
    int rs = 0;
    while (rs < n) {
        int re = rs + 1;                                     // a block always takes one row
        const int64_t a0 = rp[rs];
        const bool solo = solo_len > 0 && rp[rs + 1] - rp[rs] > solo_len;
        while (!solo && re < n && re - rs < max_rows && (int64_t)rp[re + 1] - a0 <= max_nnz &&
               !(solo_len > 0 && rp[re + 1] - rp[re] > solo_len)) ++re;
        blk.push_back(re);
        rs = re;
    }
}

// lanes per row of the vector-CSR kernel, from the row-length statistics (measured on MI355X):
//   regular short rows (7-pt: mean 7, longest 7): the largest power of two not above the mean — 4 lanes take two
//   steps and beat 8 lanes' one step with half the lanes idle (0.458 vs 0.509 ms at 256^3);
//   otherwise the smallest power of two not below the mean (27-pt -> 32), and once more when the longest row is far
//   out in the tail (power-law, longest 8192 vs mean 8.4: 32 lanes 1.81 ms, 16 lanes 1.92 ms, 8 lanes 2.16 ms).
static int pick_lanes_per_row(int n, int64_t nnz, int max_len)
{
    if (const char *e = getenv("ITS_VEC_LANES")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return v; }
    const double mean = n > 0 ? (double)nnz / n : 1.0;
    int l = 2;
    while (l < 64 && l < mean) l <<= 1;
    // regular rows (stencils): two or more entries per lane beat one entry and a longer butterfly — 7-pt: 4 lanes 0.38 ms,
    // 8 lanes 0.46; 27-pt 128^3: 16 lanes 0.175 ms, 8 lanes 0.177, 32 lanes 0.242
    if (((max_len <= 16 && mean <= 8.0) || max_len <= 1.5 * mean + 2.0) && l > 2 && l > mean) l >>= 1;
    else if (max_len > 64.0 * mean && l < 64) l <<= 1;
    return l;
}

// SELL-64-sigma image, layout of CUDA_tests/helpers/utilities.cpp:183-361 (C = 64): rows sorted
// by descending length inside windows of sigma rows, chunk c stored column-major at
// chunk_ptr[c] + j*64 + r, padding = (col 0, val 0.0).
static int build_sell(its_handle *h, CsrDev &A, int n, const int *rp)
{
    // the order and the chunk lengths on the host cores in parallel (its_host::sell_order), chunk pointers in 64 bits, the
    // col / val arrays filled on the device from the CRS arrays that are already there (launch_sell_fill): nothing of the
    // size of the matrix is built or copied on the host any more
    const int C = 64;
    int sigma = 4096;
    bool sigma_fixed = false;
    if (const char *e = getenv("ITS_SELL_SIGMA")) { sigma = std::max(1, atoi(e)); sigma_fixed = true; }
    const int n_chunks = (n + C - 1) / C, n_pad = n_chunks * C;
    const int64_t nnz = n > 0 ? rp[n] : 0;
    std::vector<int> order((size_t)n_pad), clen((size_t)n_chunks);
    std::vector<long long> cptr((size_t)n_chunks + 1);
    int64_t tot = 0;
    // sigma grows (x8) while less than 60 % of the stored elements are real: power-law rows need a
    // wide sorting window before chunks of 64 rows have similar lengths
    for (;;) {
        tot = its_host::sell_order(n, n_pad, rp, sigma, C, order.data(), clen.data());
        if (sigma_fixed || sigma >= n_pad || tot == 0 || (double)nnz / (double)tot >= 0.6) break;
        sigma = (int)std::min<int64_t>((int64_t)sigma * 8, n_pad);
    }
    tot = 0;
    for (int c = 0; c < n_chunks; ++c) { cptr[c] = tot; tot += (int64_t)clen[c] * C; }
    cptr[n_chunks] = tot;
    // work-balanced static partition: groups of 4 chunks (one per wave) dealt to ITS_NPART workgroup
    // slots so that every slot gets about the same number of stored elements (+ a per-group constant)
    const int n_grp = (n_chunks + 3) / 4;
    std::vector<int> gstart(ITS_NPART + 1, n_grp);
    {
        std::vector<int64_t> pre((size_t)n_grp + 1, 0);
        for (int g = 0; g < n_grp; ++g) {
            int64_t wk = 16;
            for (int c = 4 * g; c < std::min(n_chunks, 4 * g + 4); ++c) wk += clen[c];
            pre[g + 1] = pre[g] + wk;
        }
        int g = 0;
        for (int sl = 0; sl <= ITS_NPART; ++sl) {
            const int64_t target = pre[n_grp] * sl / ITS_NPART;
            while (g < n_grp && pre[g] < target) ++g;
            gstart[sl] = g;
        }
        gstart[0] = 0; gstart[ITS_NPART] = n_grp;
    }
    for (int i = 0; i < n_pad; ++i) if (order[i] >= n) order[i] = -1;     // (order becomes the image's row permutation)
    A.sell_chunks = n_chunks; A.sell_sigma = sigma; A.sell_elems = tot;
    if (dalloc(h, &A.sell_grp_start, gstart.size())) return ITS_EHIP;
    HIPCHK(h, hipMemcpy(A.sell_grp_start, gstart.data(), gstart.size() * sizeof(int), hipMemcpyHostToDevice));
    if (dalloc(h, &A.sell_chunk_ptr, (size_t)n_chunks + 1) || dalloc(h, &A.sell_chunk_len, (size_t)n_chunks) ||
        dalloc(h, &A.sell_perm, (size_t)n_pad) || dalloc(h, &A.sell_col, (size_t)tot) || dalloc(h, &A.sell_val, (size_t)tot))
        return ITS_EHIP;
    HIPCHK(h, hipMemcpy(A.sell_chunk_ptr, cptr.data(), ((size_t)n_chunks + 1) * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(A.sell_chunk_len, clen.data(), (size_t)n_chunks * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(A.sell_perm, order.data(), (size_t)n_pad * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemsetAsync(A.sell_col, 0, (size_t)tot * sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(A.sell_val, 0, (size_t)tot * sizeof(double), h->stream));
    launch_sell_fill(A, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ITS_OK;
}

int its_upload_csr_local(its_handle *h, int which, int n_rows, int n_cols, int64_t nnz,
                         const int *row_ptr, const int *col, const double *val, int format)
{
    if (h) h->gm_ahead_iter = -1;
    if (!h || which < 0 || which > ITS_MAT_DU || n_rows < 0 || nnz < 0 || !row_ptr) return ITS_EINVAL;
    if (nnz > 0 && (!col || !val)) FAIL(h, ITS_EINVAL, "null col/val");
    if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) FAIL(h, ITS_EINVAL, "row_ptr[0] != 0 or row_ptr[n] != nnz");
    // (the walks of long rows prefetch up to 3 chunks — a few thousand entries — past the entry they work on with 32-bit
    // indices, its_panel.hip: the limit keeps those sums inside int32)
    if (nnz > INT32_MAX - 65536) FAIL(h, ITS_EUNSUPPORTED, "nnz exceeds int32 offsets (sparse_matrix.hpp:20-27; limit 2^31 - 65536 entries)");
    HIPCHK(h, hipSetDevice(h->device));
    PhaseTimer pt(which == ITS_MAT_A ? "upload A" : which == ITS_MAT_L ? "upload L" : which == ITS_MAT_U ? "upload U" : "image of D + U");
    // validate what the kernels assume BEFORE anything of the old matrix is given up: monotone row_ptr, columns
    // inside [0, n_cols).  A rejected upload leaves the handle as it was.
    {
        const int bad = its_host::validate_crs(n_rows, n_cols, nnz, row_ptr, col);
        if (bad == 1) FAIL(h, ITS_EINVAL, "row_ptr not monotone");
        if (bad == 2) FAIL(h, ITS_EINVAL, "column index out of range");
    }
    pt.lap("validate");
    CsrDev &A = h->mat[which];
    // from here on the old matrix is gone: whatever was derived from it goes with it, and a failure below leaves
    // the slot empty (have[which] false), never half-built
    h->have[which] = false;
    h->setup_done = false; h->ahead = false;
    if (which != ITS_MAT_DU) {                    // the image of D + U describes the old A / L / U
        h->gs_prefix_ready = false; h->have[ITS_MAT_DU] = false;
        free_csr(h->mat[ITS_MAT_DU]);
    }
    if (which == ITS_MAT_A) h->residual_done = false;
    if (which == ITS_MAT_L) {                     // the sweep schedule describes the old L
        dfree(h->level_rows); dfree(h->sweep_desc); free_level_ell(h); free_lines(h);
        dfree(h->sw_col); dfree(h->sw_val); dfree(h->sw_D); h->sw_w = 0;
        h->n_levels = 0; h->level_ptr.clear(); h->level_nshort.clear(); h->level_desc_first.clear(); h->level_desc_count.clear(); h->level_empty_lo.clear();
        h->L_rp.clear(); h->L_maxcol.clear();
    }
    free_csr(A);
    A.n_rows = n_rows; A.n_cols = n_cols; A.nnz = nnz;
    if (dalloc(h, &A.row_ptr, (size_t)n_rows + 1) || dalloc(h, &A.col, (size_t)nnz + CSR_PAD) || dalloc(h, &A.val, (size_t)nnz + CSR_PAD))
        return ITS_EHIP;
    HIPCHK(h, hipMemcpy(A.row_ptr, row_ptr, ((size_t)n_rows + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(A.col + nnz, 0, CSR_PAD * sizeof(int)));
    HIPCHK(h, hipMemset(A.val + nnz, 0, CSR_PAD * sizeof(double)));
    if (nnz) {
        HIPCHK(h, hipMemcpy(A.col, col, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(A.val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
    }
    pt.lap("copy CRS to the device");
    {
        // column-offset dictionary: at most 256 distinct (col - row) over the whole matrix; value dictionary: at most
        // 256 distinct fp64 bit patterns (constant-coefficient stencils).  Both in first-occurrence order, built by the
        // OpenMP passes of its_host.cpp.
        std::vector<unsigned char> cidx((size_t)nnz + CSR_PAD, 0);
        int dict[256]; int n_off = 0;
        const bool ok = nnz > 0 && getenv("ITS_NO_OFFSET_DICT") == nullptr &&
                        its_host::offset_dict(n_rows, row_ptr, col, cidx.data(), dict, &n_off);
        if (ok) {
            A.n_off = n_off;
            if (dalloc(h, &A.cidx, cidx.size()) || dalloc(h, &A.off_dict, 256)) return ITS_EHIP;
            HIPCHK(h, hipMemcpy(A.cidx, cidx.data(), cidx.size(), hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(A.off_dict, dict, 256 * sizeof(int), hipMemcpyHostToDevice));
            double vd[256]; int n_vals = 0;
            std::vector<unsigned char> &vidx = cidx;                   // reuse the host buffer
            if (its_host::value_dict(nnz, val, vidx.data(), vd, &n_vals)) {
                A.n_vals = n_vals;
                if (dalloc(h, &A.vidx, vidx.size()) || dalloc(h, &A.val_dict, 256)) return ITS_EHIP;
                HIPCHK(h, hipMemcpy(A.vidx, vidx.data(), vidx.size(), hipMemcpyHostToDevice));
                HIPCHK(h, hipMemcpy(A.val_dict, vd, 256 * sizeof(double), hipMemcpyHostToDevice));
            }
        }
    }
    pt.lap("offset / value dictionaries");
    std::vector<int> blk, longest;
    build_stream_plan(n_rows, row_ptr, blk);
    A.n_blocks = (int)blk.size() - 1;
    std::vector<int4> desc(std::max(A.n_blocks, 1));
    longest.assign(std::max(A.n_blocks, 1), 0);
    its_host::block_longest(A.n_blocks, blk.data(), row_ptr, longest.data());
    // .y = rows | (longest row of the block, capped) << 16: the kernel's batch count
    for (int i = 0; i < A.n_blocks; ++i)
        desc[i] = make_int4(blk[i], (blk[i + 1] - blk[i]) | (std::min(longest[i], 0x7fff) << 16), row_ptr[blk[i]], row_ptr[blk[i + 1]]);
    if (dalloc(h, &A.blk_desc, desc.size())) return ITS_EHIP;
    HIPCHK(h, hipMemcpy(A.blk_desc, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice));
    // hub rows: out of the plan, onto the wave-per-row launch (its_kernels.h, CsrDev::blk_desc_short).  One lane of the
    // irregular-row kernel walks a row at 16 ns per entry while its workgroup waits, a wave of spmv_long_kernel at 7-9 ns:
    // the cut is not the row block's capacity but SPMV_SOLO_LEN entries (200 000-row power-law matrix, the launch over the
    // plan: 34 us with the rows of up to 1536 entries in it)
    {
        std::vector<int4> shortd;
        std::vector<int> longr;
        static const int solo_len = []{ const char *e = getenv("ITS_SPMV_SOLO_LEN"); const int v = e ? atoi(e) : 0; return v > 0 ? v : SPMV_SOLO_LEN; }();
        bool any = false;
        for (int i = 0; i < A.n_blocks && !any; ++i) any = (longest[i] > solo_len);
        if (any) {
            std::vector<int> blk2, longest2;
            build_stream_plan(n_rows, row_ptr, blk2, ITS_BLOCK, SPMV_NNZ_MAX, solo_len);
            const int nb2 = (int)blk2.size() - 1;
            longest2.assign(std::max(nb2, 1), 0);
            its_host::block_longest(nb2, blk2.data(), row_ptr, longest2.data());
            for (int i = 0; i < nb2; ++i) {
                if (longest2[i] > solo_len) longr.push_back(blk2[i]);      // (a block of one row by construction)
                else shortd.push_back(make_int4(blk2[i], (blk2[i + 1] - blk2[i]) | (std::min(longest2[i], 0x7fff) << 16),
                                                row_ptr[blk2[i]], row_ptr[blk2[i + 1]]));
            }
        }
        if (!longr.empty() && !shortd.empty()) {
            std::sort(longr.begin(), longr.end(), [&](int r0, int r1) {
                const int l0 = row_ptr[r0 + 1] - row_ptr[r0], l1 = row_ptr[r1 + 1] - row_ptr[r1];
                return l0 != l1 ? l0 > l1 : r0 < r1;
            });
            A.n_blocks_short = (int)shortd.size(); A.irr_n_long = (int)longr.size();
            if (dalloc(h, &A.blk_desc_short, shortd.size()) || dalloc(h, &A.irr_long_rows, longr.size())) return ITS_EHIP;
            HIPCHK(h, hipMemcpy(A.blk_desc_short, shortd.data(), shortd.size() * sizeof(int4), hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(A.irr_long_rows, longr.data(), longr.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    // longest row; and what trip_gs compares between A and crs_U before it lets one pass over A stand in for the pass
    // over U: count, column sum and an order-independent hash of the value bits of the entries right of the diagonal
    its_host::row_stats(n_rows, row_ptr, col, val, which != ITS_MAT_L, &A.max_row_len, &A.n_upper, &A.upper_col_sum, &A.upper_val_xor);
    if (A.max_row_len <= 8 && n_rows > 0) {
        // rows this short may run on spmv_stream4_kernel, which takes larger row blocks (its_kernels.h)
        build_stream_plan(n_rows, row_ptr, blk, ITS_BLOCK, SPMV_NNZ_BIG);
        A.n_blocks_big = (int)blk.size() - 1;
        desc.assign(std::max(A.n_blocks_big, 1), make_int4(0, 0, 0, 0));
        longest.assign(std::max(A.n_blocks_big, 1), 0);
        its_host::block_longest(A.n_blocks_big, blk.data(), row_ptr, longest.data());
        for (int i = 0; i < A.n_blocks_big; ++i)
            desc[i] = make_int4(blk[i], (blk[i + 1] - blk[i]) | (longest[i] << 16), row_ptr[blk[i]], row_ptr[blk[i + 1]]);
        if (dalloc(h, &A.blk_desc_big, desc.size())) return ITS_EHIP;
        HIPCHK(h, hipMemcpy(A.blk_desc_big, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    A.lanes_per_row = pick_lanes_per_row(n_rows, nnz, A.max_row_len);
    pt.lap("stream plans + statistics");
    if (format == ITS_FMT_SELL) { int rc = build_sell(h, A, n_rows, row_ptr); if (rc) return rc; pt.lap("SELL image"); }
    if (format == ITS_FMT_MIXED) {
        const size_t ne = (size_t)std::max<int64_t>(nnz, 1);
        std::vector<int> rpd((size_t)n_rows + 1), rps((size_t)n_rows + 1), rph((size_t)n_rows + 1), cd(ne), cs(ne), ch(ne);
        std::vector<double> vd(ne);
        std::vector<float> vs(ne);
        std::vector<uint16_t> vh(ne);
        its_host::split_by_tiers(n_rows, row_ptr, col, val, h->mixed_mode, h->mixed_threshold, h->mixed_threshold2,
                                 rpd.data(), cd.data(), vd.data(), rps.data(), cs.data(), vs.data(), rph.data(), ch.data(), vh.data());
        A.mix_nnz_d = rpd[n_rows]; A.mix_nnz_s = rps[n_rows]; A.mix_nnz_h = rph[n_rows];
        if (dalloc(h, &A.mix_rp_d, (size_t)n_rows + 1) || dalloc(h, &A.mix_rp_s, (size_t)n_rows + 1) || dalloc(h, &A.mix_rp_h, (size_t)n_rows + 1) ||
            dalloc(h, &A.mix_col_d, (size_t)A.mix_nnz_d) || dalloc(h, &A.mix_val_d, (size_t)A.mix_nnz_d) ||
            dalloc(h, &A.mix_col_s, (size_t)A.mix_nnz_s) || dalloc(h, &A.mix_val_s, (size_t)A.mix_nnz_s) ||
            dalloc(h, &A.mix_col_h, (size_t)A.mix_nnz_h) || dalloc(h, &A.mix_val_h, (size_t)A.mix_nnz_h)) return ITS_EHIP;
        HIPCHK(h, hipMemcpy(A.mix_rp_d, rpd.data(), ((size_t)n_rows + 1) * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(A.mix_rp_s, rps.data(), ((size_t)n_rows + 1) * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(A.mix_rp_h, rph.data(), ((size_t)n_rows + 1) * sizeof(int), hipMemcpyHostToDevice));
        if (A.mix_nnz_d) {
            HIPCHK(h, hipMemcpy(A.mix_col_d, cd.data(), (size_t)A.mix_nnz_d * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(A.mix_val_d, vd.data(), (size_t)A.mix_nnz_d * sizeof(double), hipMemcpyHostToDevice));
        }
        if (A.mix_nnz_s) {
            HIPCHK(h, hipMemcpy(A.mix_col_s, cs.data(), (size_t)A.mix_nnz_s * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(A.mix_val_s, vs.data(), (size_t)A.mix_nnz_s * sizeof(float), hipMemcpyHostToDevice));
        }
        if (A.mix_nnz_h) {
            HIPCHK(h, hipMemcpy(A.mix_col_h, ch.data(), (size_t)A.mix_nnz_h * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(A.mix_val_h, vh.data(), (size_t)A.mix_nnz_h * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        pt.lap("mixed-precision image");
    }
    // sliced ELL image for short regular rows (its_ell.hip), built on the device from the arrays just uploaded
    {
        const bool dict_forms = ell_dict_forms_enabled();
        const bool narrow = dict_forms && A.n_off >= 1 && A.n_off <= ELL_MAX_OFF && A.max_row_len >= 1 && A.max_row_len <= ELL_MAX_W;
        const bool wide = dict_forms && !narrow && A.n_off >= 1 && A.n_off <= ELLW_MAX_OFF && A.max_row_len >= 1 && A.max_row_len <= ELLW_MAX_W;
        // explicit columns (its_ellg.hip): any matrix with rows of <= 8 entries that has no dictionary image
        const bool generic = !narrow && !wide && A.max_row_len >= 1 && A.max_row_len <= ELL_MAX_W && getenv("ITS_NO_ELLG") == nullptr;
        if ((narrow || wide || generic) && n_rows > 0 && getenv("ITS_NO_ELL") == nullptr) {
            const int64_t n_blk = ((int64_t)n_rows + ELL_RB - 1) / ELL_RB;
            const int64_t stored = n_blk * ELL_RB * A.max_row_len;
            if (stored <= nnz + nnz / 4 + 8 * ELL_RB && stored < ((int64_t)1 << 40) && n_blk < (1 << 24)) {
                A.ell_w = A.max_row_len; A.ell_blocks = (int)n_blk; A.ell_kind = narrow ? 1 : wide ? 2 : 3;
                if (generic) {
                    // period of the work map from the most frequent col - row of a sample of rows (its_host.cpp)
                    int offs[16]; const int n_offs = its_host::frequent_offsets(n_rows, row_ptr, col, offs, 15);
                    A.ell_pb = ell_choose_period(offs, n_offs);
                    const size_t slices = (size_t)n_blk * (ELL_RB / ELL_S);
                    if (dalloc(h, &A.ell_val, (size_t)stored) || dalloc(h, &A.ellg_desc, slices) || dalloc(h, &A.ellg_base, slices * A.ell_w)) return ITS_EHIP;
                    HIPCHK(h, ellg_build(A, h->stream, &A.ellg_idx_words, A.ellg_hist));
                    if (getenv("ITS_DEBUG_ELLG"))
                        fprintf(stderr, "its: explicit-column ELL image: W = %d, %zu slices (8-bit %d, 16-bit %d, 32-bit %d), %.1f index bytes per row, period %d row blocks\n",
                                A.ell_w, slices, A.ellg_hist[0], A.ellg_hist[1], A.ellg_hist[2], 4.0 * A.ellg_idx_words / std::max(n_rows, 1), A.ell_pb);
                } else {
                {
                    std::vector<int> dict(256);
                    HIPCHK(h, hipMemcpy(dict.data(), A.off_dict, 256 * sizeof(int), hipMemcpyDeviceToHost));
                    A.ell_pb = ell_choose_period(dict.data(), A.n_off);
                }
                const size_t idx_words = narrow ? (size_t)n_blk * ELL_RB : (size_t)n_blk * ELL_RB * ((A.ell_w + 3) / 4);
                if (dalloc(h, &A.ell_val, (size_t)stored) || dalloc(h, &A.ell_idx, idx_words)) return ITS_EHIP;
                launch_ell_build(A, h->stream);
                HIPCHK(h, hipGetLastError());
                HIPCHK(h, hipStreamSynchronize(h->stream));
                }
            }
        }
    }
    pt.lap("ELL image");
    if (format == ITS_FMT_MIXED) {
        // tiered sliced-ELL form of the split (its_ellm.hip) where the narrow ELL image exists; an extra — without it the
        // products run on the CRS split
        if (ellm_build(A, h->mixed_mode, h->mixed_threshold, h->mixed_threshold2, h->stream) != hipSuccess || !A.ellm_img) {
            dfree(A.ellm_img); dfree(A.ellm_tier); dfree(A.ellm_off); A.ellm_bytes = 0;
            (void)hipGetLastError();
        }
        pt.lap("tiered ELL image");
    }
    if (which == ITS_MAT_A || which == ITS_MAT_U || which == ITS_MAT_DU) {
        // column-panel blocked image for large irregular matrices (its_panel.hip); crs_U for the b - U x pass of GS
        // (the image is an extra: when it cannot be built — its 12 bytes per entry or the sort's 24 do not fit — the matrix
        // stays on the irregular-row kernel)
        if (build_panel_image(h, A, row_ptr) != ITS_OK) {
            dfree(A.pan_key); dfree(A.pan_val); dfree(A.pan_desc); dfree(A.pan_tile_ptr); dfree(A.pan_tile_row);
            dfree(A.pan_long); dfree(A.pan_diag); dfree(A.pan_long_rows);
            A.pan_tiles = A.pan_n_long = A.pan_chunks = 0; A.pan_nnz = 0;
            (void)hipGetLastError();
        }
        pt.lap("panel image");
    }
    if (which == ITS_MAT_L) {
        h->L_rp.assign(row_ptr, row_ptr + n_rows + 1);
        h->L_maxcol.assign((size_t)std::max(n_rows, 1), -1);
        its_host::row_maxcol(n_rows, row_ptr, col, n_rows, h->L_maxcol.data());
        // columns at or beyond the rows (halo columns of a row partition) are left out of L_maxcol; remember whether
        // there are any: outside a distributed run they make L not strictly lower
        h->L_has_halo_cols = false;
        if (n_cols > n_rows) {
            std::vector<int> any((size_t)std::max(n_rows, 1), -1);
            its_host::row_maxcol(n_rows, row_ptr, col, n_cols, any.data());
            for (int i = 0; i < n_rows && !h->L_has_halo_cols; ++i) h->L_has_halo_cols = any[i] >= n_rows;
        }
        pt.lap("row structure of L for the sweep schedule");
    }
    h->have[which] = true;
    if (which == ITS_MAT_A) { h->fmt = format; h->setup_done = false; h->residual_done = false; }
    return ITS_OK;
}

extern "C" int its_upload_crs(its_handle *h, int which, int n_rows, int n_cols, int64_t nnz,
                              const int *row_ptr, const int *col, const double *val, int format)
{
    if (!h) return ITS_EINVAL;
    if (format < ITS_FMT_CSR_EXACT || format > ITS_FMT_MIXED) FAIL(h, ITS_EINVAL, "unknown format");
    if (which < ITS_MAT_A || which > ITS_MAT_U) return ITS_EINVAL;
    int rc = its_upload_csr_local(h, which, n_rows, n_cols, nnz, row_ptr, col, val, format);
    if (rc == ITS_OK && which == ITS_MAT_A) { h->n = n_rows; h->n_ext = n_cols; }
    return rc;
}

extern "C" int its_set_mixed_threshold(its_handle *h, double threshold)
{
    if (!h) return ITS_EINVAL;
    if (threshold != threshold) FAIL(h, ITS_EINVAL, "threshold is NaN");
    h->mixed_mode = ITS_AP_DP_SP; h->mixed_threshold = threshold; h->mixed_threshold2 = 0.0;
    return ITS_OK;
}

extern "C" int its_set_mixed_mode(its_handle *h, int ap_value_type, double threshold_1, double threshold_2)
{
    if (!h) return ITS_EINVAL;
    if (ap_value_type < ITS_AP_DP_SP || ap_value_type > ITS_AP_DP_SP_HP) FAIL(h, ITS_EINVAL, "ap_value_type: ITS_AP_DP_SP, _DP_HP, _SP_HP or _DP_SP_HP (config.mk:57)");
    if (threshold_1 != threshold_1 || threshold_2 != threshold_2) FAIL(h, ITS_EINVAL, "threshold is NaN");
    if (ap_value_type == ITS_AP_DP_SP_HP && threshold_2 > threshold_1) FAIL(h, ITS_EINVAL, "AP_THRESHOLD_2 must not exceed AP_THRESHOLD_1");
    h->mixed_mode = ap_value_type; h->mixed_threshold = threshold_1; h->mixed_threshold2 = threshold_2;
    return ITS_OK;
}

extern "C" int its_mixed_tiers(its_handle *h, int which, int64_t *nnz_by_tier, int64_t *value_bytes, int *ell_form)
{
    if (!h || which < 0 || which > 2) return ITS_EINVAL;
    if (!h->have[which] || !h->mat[which].mix_rp_d) FAIL(h, ITS_ESTATE, "matrix was not uploaded with ITS_FMT_MIXED");
    const CsrDev &A = h->mat[which];
    if (nnz_by_tier) { nnz_by_tier[0] = A.mix_nnz_d; nnz_by_tier[1] = A.mix_nnz_s; nnz_by_tier[2] = A.mix_nnz_h; }
    if (ell_form) *ell_form = A.ellm_img ? 1 : 0;
    if (value_bytes) *value_bytes = A.ellm_img ? A.ellm_bytes : 8 * A.mix_nnz_d + 4 * A.mix_nnz_s + 2 * A.mix_nnz_h;
    return ITS_OK;
}

extern "C" int its_mixed_info(its_handle *h, int which, int64_t *nnz_fp64, int64_t *nnz_fp32)
{
    if (!h || which < 0 || which > 2) return ITS_EINVAL;
    if (!h->have[which] || !h->mat[which].mix_rp_d) FAIL(h, ITS_ESTATE, "matrix was not uploaded with ITS_FMT_MIXED");
    if (nnz_fp64) *nnz_fp64 = h->mat[which].mix_nnz_d;
    if (nnz_fp32) *nnz_fp32 = h->mat[which].mix_nnz_s;
    return ITS_OK;
}

extern "C" int its_set_option(its_handle *h, const char *name, int value)
{
    if (!h || !name) return ITS_EINVAL;
    if (!strcmp(name, "gmres_givens")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "gmres_givens: 0 (rotations) or 1 (reference dense products)");
        if (h->setup_done && h->solver == ITS_GMRES) FAIL(h, ITS_ESTATE, "set gmres_givens before its_solver_setup");
        h->gmres_givens = value;
        return ITS_OK;
    }
    if (!strcmp(name, "dist_overlap")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "dist_overlap: 0 or 1");
        h->dist_overlap = value;
        return ITS_OK;
    }
    if (!strcmp(name, "sweep_stream")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "sweep_stream: 0 or 1");
        h->sweep_stream = value;
        return ITS_OK;
    }
    if (!strcmp(name, "cg_variant")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "cg_variant: 0 (the reference's CG) or 1 (pipelined CG)");
        if (h->setup_done && h->solver == ITS_CG) FAIL(h, ITS_ESTATE, "set cg_variant before its_solver_setup");
        h->cg_variant = value;
        return ITS_OK;
    }
    if (!strcmp(name, "gmres_ortho")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "gmres_ortho: 0 (modified Gram-Schmidt, the reference) or 1 (CGS-2)");
        if (h->setup_done && h->solver == ITS_GMRES) FAIL(h, ITS_ESTATE, "set gmres_ortho before its_solver_setup");
        h->gmres_ortho = value;
        return ITS_OK;
    }
    if (!strcmp(name, "gmres_persist")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "gmres_persist: 0 (one launch per basis vector) or 1 (one persistent launch per trip)");
        if (h->setup_done && h->solver == ITS_GMRES) FAIL(h, ITS_ESTATE, "set gmres_persist before its_solver_setup");
        h->gmres_persist = value;
        return ITS_OK;
    }
    if (!strcmp(name, "gmres_persist_inject_fail")) {       // tests of the fallback: that many launches give up on purpose
        if (value < 0) FAIL(h, ITS_EINVAL, "gmres_persist_inject_fail >= 0");
        h->mgs_inject_fail = value;
        return ITS_OK;
    }
    if (!strcmp(name, "gmres_persist_inject_kind")) {       // 1: workgroup 0 never publishes; 2: a follower gives up at the last step
        if (value < 1 || value > 2) FAIL(h, ITS_EINVAL, "gmres_persist_inject_kind: 1 or 2");
        h->mgs_inject_kind = value;
        return ITS_OK;
    }
    if (!strcmp(name, "gmres_persist_min_n")) {
        if (value < 1) FAIL(h, ITS_EINVAL, "gmres_persist_min_n >= 1");
        if (h->setup_done && h->solver == ITS_GMRES) FAIL(h, ITS_ESTATE, "set gmres_persist_min_n before its_solver_setup");
        h->gmres_persist_min_n = value;
        return ITS_OK;
    }
    if (!strcmp(name, "readback")) {
        if (value < 0 || value > 2) FAIL(h, ITS_EINVAL, "readback: 0 (watch the pinned word), 1 (synchronize), 2 (copy + synchronize)");
        g_readback_mode = value;
        return ITS_OK;
    }
    if (!strcmp(name, "watch_window_ms")) {
        if (value < 0) FAIL(h, ITS_EINVAL, "watch_window_ms >= 0");
        h->watch_window_ms = value; h->watch_misses = 0;
        return ITS_OK;
    }
    if (!strcmp(name, "sweep_ell_levels")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "sweep_ell_levels: 0 or 1");
        h->sweep_ell_levels = value;
        return ITS_OK;
    }
    if (!strcmp(name, "sweep_lines")) {
        if (value < 0 || value > 2) FAIL(h, ITS_EINVAL, "sweep_lines: 0 (level schedule) or 1 (line-pipelined sweep wherever the matrix qualifies; 2 means the same)");
        h->sweep_lines = value;
        return ITS_OK;
    }
    if (!strcmp(name, "sweep_lines_fault")) {
        // test hook: the first tile of the next line sweeps keeps its edge lines to itself, so its neighbours' polls run
        // into their time limit — what a hung or time-sliced GPU looks like to the sweep (tests/test_gpu_sweep_lines.py)
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "sweep_lines_fault: 0 or 1");
        h->lines.mute_first_tile = value != 0;
        return ITS_OK;
    }
    if (!strcmp(name, "debug_poison_lds")) {
        // test hook: fills the LDS of every CU with a NaN pattern (what a kernel finds there is whatever the last one left)
        launch_lines_poison_lds(h->stream);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return ITS_OK;
    }
    if (!strcmp(name, "gmres_ahead")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "gmres_ahead: 0 or 1");
        h->gmres_ahead = value; h->gm_ahead_iter = -1;
        return ITS_OK;
    }
    if (!strcmp(name, "sweep_run")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "sweep_run: 0 or 1");
        h->sweep_run = value;
        return ITS_OK;
    }
    if (!strcmp(name, "sweep_ell")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "sweep_ell: 0 or 1");
        h->sweep_ell = value;
        return ITS_OK;
    }
    if (!strcmp(name, "gs_first_level_fused")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "gs_first_level_fused: 0 or 1");
        h->gs_first_level_fused = value;
        h->ahead = false;
        return ITS_OK;
    }
    if (!strcmp(name, "gs_prefix")) {
        if (value < 0 || value > 2) FAIL(h, ITS_EINVAL, "gs_prefix: 0 (off), 1 (where it pays: irregular matrices) or 2 (wherever it applies)");
        if (h->setup_done && h->solver == ITS_GAUSS_SEIDEL && value != h->gs_prefix) FAIL(h, ITS_ESTATE, "set gs_prefix before its_solver_setup");
        h->gs_prefix = value;
        return ITS_OK;
    }
    if (!strcmp(name, "fused_residual")) {
        if (value != 0 && value != 1) FAIL(h, ITS_EINVAL, "fused_residual: 0 or 1");
        h->fused_residual = value;
        h->ahead = false;
        return ITS_OK;
    }
    {
        int rc = spmv_tune(name, value);            // process-wide launch tunables of the exact SpMV
        if (rc == -2) rc = ell_tune(name, value);
        if (rc == -2) rc = panel_tune(name, value);
        if (rc == 0) return ITS_OK;
        if (rc == -1) FAIL(h, ITS_EINVAL, std::string("bad value for option ") + name);
    }
    FAIL(h, ITS_EINVAL, std::string("unknown option ") + name);
}

extern "C" int its_plan_info(its_handle *h, int which, int format, int *lanes_per_row, double *sell_occ, int *n_row_blocks)
{
    if (!h || which < 0 || which > 2 || !h->have[which]) return ITS_EINVAL;
    (void)format;
    const CsrDev &A = h->mat[which];
    if (lanes_per_row) *lanes_per_row = A.lanes_per_row;
    if (sell_occ) *sell_occ = A.sell_elems ? (double)A.nnz / (double)A.sell_elems : 0.0;
    if (n_row_blocks) *n_row_blocks = A.n_blocks;
    return ITS_OK;
}

extern "C" int its_ell_info(its_handle *h, int which, int *kind, int *width, int64_t *index_words, int *slices_by_delta_width)
{
    if (!h || which < 0 || which > 2 || !h->have[which]) return ITS_EINVAL;
    const CsrDev &A = h->mat[which];
    if (kind) *kind = A.ell_val ? A.ell_kind : 0;
    if (width) *width = A.ell_val ? A.ell_w : 0;
    const int64_t rows_pad = (int64_t)A.ell_blocks * ELL_RB;
    if (index_words) *index_words = !A.ell_val ? 0 : A.ell_kind == 3 ? A.ellg_idx_words : A.ell_kind == 2 ? rows_pad * ((A.ell_w + 3) / 4) : rows_pad;
    if (slices_by_delta_width) for (int i = 0; i < 3; ++i) slices_by_delta_width[i] = (A.ell_val && A.ell_kind == 3) ? A.ellg_hist[i] : 0;
    return ITS_OK;
}

extern "C" int its_gmres_info(its_handle *h, int *persist_grid, int *fallbacks)
{
    if (!h) return ITS_EINVAL;
    if (persist_grid) *persist_grid = h->mgs_grid;
    if (fallbacks) *fallbacks = h->mgs_fallbacks;
    return ITS_OK;
}

extern "C" int its_gs_info(its_handle *h, int *prefix_ready, int64_t *nnz_du, char *kernel, int kernel_cap)
{
    if (!h) return ITS_EINVAL;
    const bool on = h->gs_prefix_ready && h->have[ITS_MAT_DU];
    if (prefix_ready) *prefix_ready = on ? 1 : 0;
    if (nnz_du) *nnz_du = on ? h->mat[ITS_MAT_DU].nnz : 0;
    if (kernel && kernel_cap > 0) {
        SpmvArgs a;
        const char *nm = on ? spmv_exact_kernel_name(h->mat[ITS_MAT_DU], a, EPI_GS_FUSED, nullptr) : "";
        snprintf(kernel, (size_t)kernel_cap, "%s", nm);
    }
    return ITS_OK;
}

extern "C" int its_sweep_info(its_handle *h, int *lines_workgroups, int *line_length, int *rows_per_chunk, int *n_levels)
{
    if (!h) return ITS_EINVAL;
    if (n_levels) *n_levels = h->n_levels;
    const bool on = h->lines_ready && h->sweep_lines && !h->lines_failed;
    if (lines_workgroups) *lines_workgroups = on ? h->lines.nby * h->lines.nbz : 0;
    if (line_length) *line_length = on ? h->lines.a : 0;
    if (rows_per_chunk) *rows_per_chunk = on ? h->lines.ch : 0;
    return ITS_OK;
}

// ---------------------------------------------------------------------------------------
static int spmv_dispatch(its_handle *h, const CsrDev &A, const SpmvArgs &a, int epi, int format)
{
    // the stream kernel gathers x[0] for idle lanes: x must have at least one element
    if (A.n_cols < 1 || A.n_rows < 1) {
        if (A.n_rows > 0 && epi == EPI_STORE) HIPCHK(h, hipMemsetAsync(a.y, 0, (size_t)A.n_rows * sizeof(double), h->stream));
        else if (A.n_rows > 0) FAIL(h, ITS_EUNSUPPORTED, "matrix without columns");
        if (a.part_sum) HIPCHK(h, hipMemsetAsync(a.part_sum, 0, ITS_NPART * sizeof(double), h->stream));
        if (a.part_max) HIPCHK(h, hipMemsetAsync(a.part_max, 0, ITS_NPART * sizeof(double), h->stream));
        return ITS_OK;
    }
    if (epi >= EPI_JACOBI_SHIPPED || format == ITS_FMT_CSR_EXACT) launch_spmv_exact(A, a, epi, h->stream);
    else if (format == ITS_FMT_CSR_VECTOR) launch_spmv_vector(A, a, epi, h->stream);
    else if (format == ITS_FMT_SELL) {
        if (!A.sell_val && A.nnz) FAIL(h, ITS_ESTATE, "matrix was not uploaded with ITS_FMT_SELL");
        launch_spmv_sell(A, a, epi, h->stream);
    } else if (format == ITS_FMT_MIXED) {
        if (!A.mix_rp_d) FAIL(h, ITS_ESTATE, "matrix was not uploaded with ITS_FMT_MIXED");
        launch_spmv_mixed(A, a, epi, h->stream);
    } else FAIL(h, ITS_EINVAL, "unknown format");
    HIPCHK(h, hipGetLastError());
    return ITS_OK;
}

// multi-GPU hooks (its_dist.hip); no-ops / identity on one GPU
int  its_dist_active(const its_handle *h);
int  its_dist_spmv(its_handle *h, const SpmvArgs &a, int epi);           // halo exchange + SpMV
int  its_dist_allreduce(its_handle *h, double *dev, int count, int op);  // 0 sum, 1 max (in place)
int  its_dist_exchange(its_handle *h, double *x, int color);             // boundary values of one colour (or all: -1)
int  its_dist_n_colors(const its_handle *h);                             // 0: no colour structure given
const int *its_dist_color_ptr(const its_handle *h);

// y = A x (+ epilogue) for the solver loops: halo exchange first when row-partitioned.
static int solver_spmv(its_handle *h, const SpmvArgs &a, int epi)
{
    if (its_dist_active(h)) return its_dist_spmv(h, a, epi);
    return spmv_dispatch(h, h->mat[ITS_MAT_A], a, epi, h->fmt);
}

// Distributed reductions: the ITS_NPART partials of every rank are summed (or maxed) element-wise across ranks,
// in place, by ONE all-reduce of 16 KB; whoever consumes them then re-sums the array exactly as on one GPU.  No
// finalize launch and no scalar hand-off in front of the collective, the same code path behind it, and with one
// rank the values are the single-GPU ones bit for bit.
static int allreduce_partials(its_handle *h, double *part, int op)
{
    return its_dist_active(h) ? its_dist_allreduce(h, part, ITS_NPART, op) : ITS_OK;
}

// partials -> one device scalar
static int finish_scalar(its_handle *h, double *part, double *out, int mode)
{
    int rc = allreduce_partials(h, part, mode == FIN_MAX ? 1 : 0);
    if (rc) return rc;
    launch_reduce_finalize(part, out, mode, h->stream);
    return ITS_OK;
}

extern "C" int its_upload_vectors(its_handle *h, const double *D, const double *b, const double *x0, int n)
{
    if (h) h->gm_ahead_iter = -1;
    if (!h || !D || !b || !x0) return ITS_EINVAL;
    if (!h->have[ITS_MAT_A] || n != h->n) FAIL(h, ITS_ESTATE, "upload the matrix first; n must equal its row count");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t ne = (size_t)std::max(h->n_ext, h->n);
    dfree(h->D); dfree(h->b); dfree(h->r); dfree(h->tmp); dfree(h->x_a); dfree(h->x_b); dfree(h->x);
    if (dalloc(h, &h->D, n) || dalloc(h, &h->b, n) || dalloc(h, &h->r, ne) || dalloc(h, &h->tmp, ne) ||
        dalloc(h, &h->x_a, ne) || dalloc(h, &h->x_b, ne) || dalloc(h, &h->x, ne)) return ITS_EHIP;
    // allocate_cpu_solver_structs zero-fills everything (solver.hpp:220-280)
    for (double *v : {h->r, h->tmp, h->x_a, h->x_b, h->x}) HIPCHK(h, hipMemsetAsync(v, 0, ne * sizeof(double), h->stream));
    HIPCHK(h, hipMemcpyAsync(h->D, D, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->b, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->x_a, x0, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));   // x_old
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->setup_done = false; h->residual_done = false; h->ahead = false;
    // the level-ordered copy of L carries the OLD diagonal: drop it (the next its_solver_setup rebuilds it; a bare
    // its_spltsv falls back to the level-list kernel, which reads D itself)
    dfree(h->sw_col); dfree(h->sw_val); dfree(h->sw_D); h->sw_w = 0;
    free_lines(h);                                    // (so does the image of the line-pipelined sweep)
    return ITS_OK;
}

extern "C" int its_init_residual(its_handle *h, double *r0_inf, double *r0_two)
{
    if (h) h->gm_ahead_iter = -1;
    debug_poison_lds(h);
    if (!h) return ITS_EINVAL;
    if (!h->x_a) FAIL(h, ITS_ESTATE, "its_upload_vectors first");
    HIPCHK(h, hipSetDevice(h->device));
    SpmvArgs a; a.x = h->x_a; a.y = h->r; a.b = h->b; a.part_sum = h->part[0]; a.part_max = h->part[1];
    int rc = solver_spmv(h, a, EPI_RESIDUAL);                       // solver.hpp:852-854
    if (rc) return rc;
    if ((rc = finish_scalar(h, h->part[0], h->sc + SC_TMP, FIN_SQRT))) return rc;
    if ((rc = finish_scalar(h, h->part[1], h->sc + SC_TMP + 1, FIN_MAX))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->pinned, h->sc + SC_TMP, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (r0_two) *r0_two = h->pinned[0];
    if (r0_inf) *r0_inf = h->pinned[1];
    h->residual_done = true;
    return ITS_OK;
}

// ---------------------------------------------------------------------------------------
// triangular-sweep schedule: level(i) = 1 + max level(j) over strictly-lower entries of row i
// ---------------------------------------------------------------------------------------
// Sweep schedule of the uploaded crs_L from its host-side row structure (its_handle::L_rp, L_maxcol).
// given_ptr != nullptr: the levels are given (the colours of a row-block partition in global multicolour order,
// rows [given_ptr[c], given_ptr[c+1]) = colour c; columns >= n are halo entries that the exchange after each colour
// supplies) instead of being derived from the dependencies.
static int build_levels(its_handle *h, int n, const int *given_ptr = nullptr, int n_given = 0)
{
    const int *Lp = h->L_rp.data();
    const int *maxcol = h->L_maxcol.data();
    std::vector<int> lev(n, 0);
    int nl = 0;
    if (given_ptr) {
        nl = n_given;
        for (int c = 0; c < n_given; ++c) for (int i = given_ptr[c]; i < given_ptr[c + 1]; ++i) lev[i] = c;
    } else {
        // Schedule 1: cut the rows, in order, into contiguous ranges without internal dependencies (a new range
        // starts at the first row that reads a row of the current one).  For a colour-by-colour ordering this
        // recovers the colours (dependency levels would scatter a colour's rows over earlier levels); for a natural
        // ordering it degenerates to about one range per row.  Needs only the largest column of every row.
        std::vector<int> rng(n, 0);
        int nr = n ? 1 : 0, start = 0;
        for (int i = 0; i < n; ++i) {
            if (maxcol[i] >= start) { start = i; ++nr; }
            rng[i] = nr - 1;
        }
        if (n == 0) nr = 0;
        if (nr <= 64) { lev.swap(rng); nl = nr; }            // a colour ordering: no need to look at the columns at all
        else {
            // Schedule 2: dependency levels, level(i) = 1 + max level(j) over the entries of row i (needs the columns:
            // copied back from the device, this is the natural-order case).  Either schedule gives every row the same
            // operands (everything it reads was finished by an earlier launch), so the choice is free: ranges win when
            // they are not many more than the levels, because a range is swept like an SpMV (below).
            const CsrDev &L = h->mat[ITS_MAT_L];
            std::vector<int> Lc((size_t)L.nnz);
            if (L.nnz) HIPCHK(h, hipMemcpy(Lc.data(), L.col, (size_t)L.nnz * sizeof(int), hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i) {
                int l = 0;
                for (int k = Lp[i]; k < Lp[i + 1]; ++k) l = std::max(l, lev[Lc[k]] + 1);
                lev[i] = l; nl = std::max(nl, l + 1);
            }
            if (nr <= 2 * nl + 16) { lev.swap(rng); nl = nr; }
        }
    }
    if (n == 0) nl = 0;
    h->n_levels = nl;
    h->level_ptr.assign(nl + 1, 0);
    for (int i = 0; i < n; ++i) ++h->level_ptr[lev[i] + 1];
    for (int l = 0; l < nl; ++l) h->level_ptr[l + 1] += h->level_ptr[l];
    std::vector<int> rows(n), fill(h->level_ptr.begin(), h->level_ptr.end() - (nl ? 1 : 0));
    // inside a level: short rows first (one lane each), then the long ones (one wave each), both ascending
    h->level_nshort.assign(nl, 0);
    for (int i = 0; i < n; ++i) if (Lp[i + 1] - Lp[i] <= SWEEP_SHORT) { rows[fill[lev[i]]++] = i; ++h->level_nshort[lev[i]]; }
    for (int i = 0; i < n; ++i) if (Lp[i + 1] - Lp[i] > SWEEP_SHORT) rows[fill[lev[i]]++] = i;
    dfree(h->level_rows);
    if (dalloc(h, &h->level_rows, n)) return ITS_EHIP;
    HIPCHK(h, hipMemcpy(h->level_rows, rows.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    // A level that is one contiguous range of many rows (a colour of the multicolour ordering) is an SpMV over
    // those rows of L with the sweep's epilogue: it gets row-block descriptors of its own (blocks end at the
    // level's end) and runs on the exact stream kernels.  Levels of the natural order are scattered rows: they
    // keep the level-list kernel.
    h->level_desc_first.assign(nl, 0); h->level_desc_count.assign(nl, 0); h->level_empty_lo.assign(nl, -1);
    free_level_ell(h); h->level_ell.assign(nl, SweepEll());
    int L_pb = -1;                                           // period of the ELL work map from L's offsets (first use)
    std::vector<int4> desc;
    std::vector<int> blk;
    for (int l = 0; l < nl; ++l) {
        const int b0 = h->level_ptr[l], cnt = h->level_ptr[l + 1] - b0;
        if (cnt < SWEEP_STREAM_MIN) continue;
        int lo = rows[b0], hi = rows[b0];
        for (int k = b0; k < b0 + cnt; ++k) { lo = std::min(lo, rows[k]); hi = std::max(hi, rows[k]); }
        if (hi - lo + 1 != cnt) continue;
        // (the first colour of a multicolour ordering: no row reads anything, x = rhs / D is all there is to do)
        if (Lp[lo + cnt] == Lp[lo]) { h->level_empty_lo[l] = lo; continue; }
        // The level as an ELL image of its own (its_ell.hip) when its rows are short and regular: values slot-major, 4- or
        // 8-bit dictionary indices, no row_ptr stream — the colours of a stencil in multicolour order (7-pt: 7 distinct
        // offsets, rows of 6; 27-pt: 52 offsets, rows of <= 26).  Needs the offset dictionary of L and 16-byte aligned slices.
        {
            const CsrDev &L = h->mat[ITS_MAT_L];
            int wv = 0;
            for (int i = lo; i < lo + cnt; ++i) wv = std::max(wv, Lp[i + 1] - Lp[i]);
            const int64_t nnz_v = (int64_t)Lp[lo + cnt] - Lp[lo];
            const int64_t n_blk = ((int64_t)cnt + ELL_RB - 1) / ELL_RB;
            const bool narrow = L.n_off >= 1 && L.n_off <= ELL_MAX_OFF && wv >= 1 && wv <= ELL_MAX_W;
            const bool wide = !narrow && L.n_off >= 1 && L.n_off <= ELLW_MAX_OFF && wv >= 1 && wv <= ELLW_MAX_W;
            if (h->sweep_ell_levels && h->D && (narrow || wide) && (lo & 1) == 0 && L.cidx && getenv("ITS_NO_ELL") == nullptr &&
                n_blk * ELL_RB * wv <= nnz_v + nnz_v / 4 + 8 * ELL_RB) {
                SweepEll &v = h->level_ell[l];
                if (L_pb < 0) {
                    std::vector<int> dict(256);
                    HIPCHK(h, hipMemcpy(dict.data(), L.off_dict, 256 * sizeof(int), hipMemcpyDeviceToHost));
                    L_pb = ell_choose_period(dict.data(), L.n_off);
                }
                v.w = wv; v.blocks = (int)n_blk; v.kind = narrow ? 1 : 2; v.lo = lo; v.cnt = cnt; v.pb = L_pb;
                const size_t idx_words = narrow ? (size_t)n_blk * ELL_RB : (size_t)n_blk * ELL_RB * ((wv + 3) / 4);
                if (dalloc(h, &v.val, (size_t)(n_blk * ELL_RB * wv)) || dalloc(h, &v.idx, idx_words)) return ITS_EHIP;
                CsrDev V;
                V.n_rows = cnt; V.row_ptr = L.row_ptr + lo; V.col = L.col; V.val = L.val; V.cidx = L.cidx; V.off_dict = L.off_dict;
                V.n_off = L.n_off; V.ell_w = wv; V.ell_blocks = (int)n_blk; V.ell_kind = v.kind; V.ell_val = v.val; V.ell_idx = v.idx;
                launch_ell_build(V, h->stream, lo);
                HIPCHK(h, hipGetLastError());
            }
        }
        build_stream_plan(cnt, Lp + lo, blk);              // row indices relative to lo
        h->level_desc_first[l] = (int)desc.size();
        const int nbl = h->level_desc_count[l] = (int)blk.size() - 1;
        std::vector<int4> hubs, rest;
        std::vector<int> longest_of(blk.size(), 0);
        its_host::block_longest(nbl, blk.data(), Lp + lo, longest_of.data());
        for (size_t i = 0; i + 1 < blk.size(); ++i) {
            const int r0 = lo + blk[i], r1 = lo + blk[i + 1];
            const int longest = longest_of[i];
            const int4 d4 = make_int4(r0, (r1 - r0) | (std::min(longest, 0x7fff) << 16), Lp[r0], Lp[r1]);
            (longest >= SWEEP_HUB ? hubs : rest).push_back(d4);
        }
        // A level lasts at least as long as the serial walk of its longest row.  Blocks are independent, so the
        // ones holding such rows go first: under the kernels' XCD-interleaved map (workgroup j of XCD e walks
        // blocks X0(e) + j, X0(e) + j + G/8, ...) the front of each eighth of the list is what the workgroups
        // start with, so every long walk begins at once instead of behind a queue of short rows.
        if (!hubs.empty() && nbl >= 64 && (int)hubs.size() <= nbl / 2) {
            std::sort(hubs.begin(), hubs.end(), [](const int4 &a, const int4 &b) { return (a.y >> 16) > (b.y >> 16); });
            size_t ir = 0;
            for (int e = 0; e < 8; ++e) {
                const int X0 = (int)(((int64_t)e * nbl) / 8), X1 = (int)(((int64_t)(e + 1) * nbl) / 8);
                int filled = 0;
                for (size_t k = e; k < hubs.size() && filled < X1 - X0; k += 8, ++filled) desc.push_back(hubs[k]);
                for (; filled < X1 - X0; ++filled) desc.push_back(rest[ir++]);
            }
        } else {
            // (original order)
            std::vector<int4> all(hubs.size() + rest.size());
            std::merge(hubs.begin(), hubs.end(), rest.begin(), rest.end(), all.begin(),
                       [](const int4 &a, const int4 &b) { return a.x < b.x; });
            desc.insert(desc.end(), all.begin(), all.end());
        }
    }
    dfree(h->sweep_desc);
    if (!desc.empty()) {
        if (dalloc(h, &h->sweep_desc, desc.size())) return ITS_EHIP;
        HIPCHK(h, hipMemcpy(h->sweep_desc, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    // Many scattered levels of short rows (a natural-order sweep of a stencil): level-ordered copy of L, so that a
    // level costs two dependent memory round trips instead of four (its_kernels.hip, sweep_ell_kernel)
    dfree(h->sw_col); dfree(h->sw_val); dfree(h->sw_D); h->sw_w = 0;
    {
        const CsrDev &L = h->mat[ITS_MAT_L];
        int n_list = 0;
        for (int l = 0; l < nl; ++l) n_list += h->level_desc_count[l] == 0 && h->level_empty_lo[l] < 0;
        if (h->sweep_ell && h->D && n > 0 && nl >= 32 && n_list * 2 >= nl && L.max_row_len >= 1 && L.max_row_len <= SWEEP_ELL_W) {
            const int W = L.max_row_len;
            if (dalloc(h, &h->sw_col, (size_t)W * n) || dalloc(h, &h->sw_val, (size_t)W * n) || dalloc(h, &h->sw_D, (size_t)n)) return ITS_EHIP;
            launch_sweep_ell_build(n, W, h->level_rows, L, h->D, h->sw_col, h->sw_val, h->sw_D, h->stream);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipStreamSynchronize(h->stream));
            h->sw_w = W;
        }
    }
    // runs of consecutive levels that the level-ordered copy serves and that fit one workgroup
    dfree(h->level_ptr_dev); h->level_run_end.assign(nl, 0);
    if (h->sw_w > 0 && nl > 0) {
        if (dalloc(h, &h->level_ptr_dev, (size_t)nl + 1)) return ITS_EHIP;
        HIPCHK(h, hipMemcpy(h->level_ptr_dev, h->level_ptr.data(), ((size_t)nl + 1) * sizeof(int), hipMemcpyHostToDevice));
        auto small = [&](int l) {
            return h->level_desc_count[l] == 0 && h->level_empty_lo[l] < 0 && !h->level_ell[l].val &&
                   h->level_ptr[l + 1] - h->level_ptr[l] <= sweep_run_rows(h->sw_w);
        };
        for (int l = 0; l < nl;) {
            if (!small(l)) { ++l; continue; }
            int e = l;
            while (e < nl && small(e)) ++e;
            if (e - l >= 2) h->level_run_end[l] = e;
            l = e;
        }
    }
    return ITS_OK;
}

// ---------------------------------------------------------------------------------------
// Gauss-Seidel, option "gs_prefix" (its_handle.h): is every row of A its row of crs_L, then ONE diagonal entry, then its
// row of crs_U — entry for entry, bitwise, in that order?  (gauss_seidel.hpp:249-630 builds crs_L / crs_U that way from a
// column-sorted A; nothing is assumed, everything compared.)  Then the sweep's chain over L is the residual chain's
// prefix.  One lane per row; flags[0] |= 1 where a row differs; diag[row] = A's diagonal entry.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(ITS_BLOCK) void gs_prefix_check_kernel(
    int n, const int *__restrict__ a_rp, const int *__restrict__ a_col, const double *__restrict__ a_val,
    const int *__restrict__ l_rp, const int *__restrict__ l_col, const double *__restrict__ l_val,
    const int *__restrict__ u_rp, const int *__restrict__ u_col, const double *__restrict__ u_val,
    double *__restrict__ diag, int *__restrict__ flags)
{
    const int row = blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (row >= n) return;
    const int a0 = a_rp[row], a1 = a_rp[row + 1], l0 = l_rp[row], l1 = l_rp[row + 1], u0 = u_rp[row], u1 = u_rp[row + 1];
    bool ok = a1 - a0 == (l1 - l0) + 1 + (u1 - u0);
    double dg = 0.0;
    if (ok) {
        int p = a0;
        for (int k = l0; k < l1 && ok; ++k, ++p)
            ok = a_col[p] == l_col[k] && a_col[p] < row && __double_as_longlong(a_val[p]) == __double_as_longlong(l_val[k]);
        ok = ok && a_col[p] == row;
        dg = a_val[p]; ++p;
        for (int k = u0; k < u1 && ok; ++k, ++p)
            ok = a_col[p] == u_col[k] && a_col[p] > row && __double_as_longlong(a_val[p]) == __double_as_longlong(u_val[k]);
    }
    diag[row] = dg;
    if (!ok) atomicOr(flags, 1);
}

int its_upload_csr_local(its_handle *h, int which, int n_rows, int n_cols, int64_t nnz,
                         const int *row_ptr, const int *col, const double *val, int format);

// Builds mat[ITS_MAT_DU] when the check above passes; leaves gs_prefix_ready false (and the trips on the pass over A)
// in every other case.  Never fails the set-up: the image is an extra.
static int build_gs_prefix(its_handle *h)
{
    h->gs_prefix_ready = false; h->have[ITS_MAT_DU] = false;
    free_csr(h->mat[ITS_MAT_DU]);
    static const bool env_off = getenv("ITS_GS_PREFIX") && atoi(getenv("ITS_GS_PREFIX")) == 0;
    if (!h->gs_prefix || env_off || !h->fused_residual || its_dist_active(h)) return ITS_OK;
    if (!h->have[ITS_MAT_A] || !h->have[ITS_MAT_L] || !h->have[ITS_MAT_U] || !h->D) return ITS_OK;
    const CsrDev &A = h->mat[ITS_MAT_A], &L = h->mat[ITS_MAT_L], &U = h->mat[ITS_MAT_U];
    const int n = h->n;
    if (n < 1 || A.n_rows != n || A.n_cols != n || L.n_rows != n || U.n_rows != n || A.nnz != L.nnz + U.nnz + n) return ITS_OK;
    // Where it pays (measured, ms per trip, off -> on).  Irregular matrices, whose passes are bound by their gathers: power-law
    // 10 M rows, multicolour, 2.48 -> 1.83; 200 000 rows 0.270 -> 0.257 (natural order 0.382 -> 0.374).  Stencils: NOT — in
    // red-black order D + U has rows of 7 entries and rows of 1, no ELL image, and the pass lands on the CRS stream kernel
    // (7-pt 256^3 0.444 -> 0.518, 27-pt 128^3 0.217 -> 0.221); in natural order the sweep's levels are latency-bound
    // launches and the second store per row costs them more than the shorter pass saves (256^3: 3.62 -> 3.87, 2-D 256^2:
    // 0.391 -> 0.422).  Option value 1 therefore takes the irregular class only; 2 forces the form wherever the check passes.
    if (h->gs_prefix == 1 && A.max_row_len <= IRR_ROW_LEN) return ITS_OK;
    PhaseTimer pt("gs_prefix");
    double *diag_dev = nullptr; int *flags = nullptr;
    auto give_up = [&]() { dfree(diag_dev); dfree(flags); h->err.clear(); (void)hipGetLastError(); return ITS_OK; };
    if (dalloc(h, &diag_dev, (size_t)n) || dalloc(h, &flags, 4)) return give_up();
    if (hipMemsetAsync(flags, 0, 4 * sizeof(int), h->stream) != hipSuccess) return give_up();
    hipLaunchKernelGGL(gs_prefix_check_kernel, dim3((n + ITS_BLOCK - 1) / ITS_BLOCK), dim3(ITS_BLOCK), 0, h->stream, n,
                       A.row_ptr, A.col, A.val, L.row_ptr, L.col, L.val, U.row_ptr, U.col, U.val, diag_dev, flags);
    int hflags[4] = {1, 0, 0, 0};
    if (hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return give_up();
    pt.lap("rows of A == [L | diagonal | U] checked on the device");
    if (hflags[0] & 1) return give_up();
    const int64_t nnz_u = U.nnz, nnz_du = nnz_u + n;
    std::vector<int> u_rp((size_t)n + 1), u_col((size_t)std::max<int64_t>(nnz_u, 1)), rp((size_t)n + 1), col((size_t)nnz_du);
    std::vector<double> u_val((size_t)std::max<int64_t>(nnz_u, 1)), dg((size_t)n), val((size_t)nnz_du);
    if (hipMemcpy(u_rp.data(), U.row_ptr, ((size_t)n + 1) * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(dg.data(), diag_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return give_up();
    if (nnz_u && (hipMemcpy(u_col.data(), U.col, (size_t)nnz_u * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
                  hipMemcpy(u_val.data(), U.val, (size_t)nnz_u * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)) return give_up();
    dfree(diag_dev); dfree(flags);
    its_host::assemble_diag_upper(n, u_rp.data(), u_col.data(), u_val.data(), dg.data(), rp.data(), col.data(), val.data());
    pt.lap("D + U assembled on the host");
    const bool setup_was = h->setup_done;
    const int rc = its_upload_csr_local(h, ITS_MAT_DU, n, n, nnz_du, rp.data(), col.data(), val.data(), ITS_FMT_CSR_EXACT);
    h->setup_done = setup_was;
    if (rc != ITS_OK) { free_csr(h->mat[ITS_MAT_DU]); h->have[ITS_MAT_DU] = false; h->err.clear(); (void)hipGetLastError(); return ITS_OK; }
    h->gs_prefix_ready = true;
    pt.lap("plans and images of D + U");
    return ITS_OK;
}

// Plan of the line-pipelined natural-order sweep (its_lines.hip) when crs_L is a stencil's: offsets {-1, -a} or {-1, -a, -b}
// with a | b | n, a a multiple of 4, and every row verified on the device.  An extra: leaves lines_ready false otherwise.
static int build_lines_plan(its_handle *h)
{
    free_lines(h);
    static const bool env_off = getenv("ITS_SWEEP_LINES") && atoi(getenv("ITS_SWEEP_LINES")) == 0;
    if (env_off || its_dist_active(h) || !h->have[ITS_MAT_L]) return ITS_OK;
    const CsrDev &L = h->mat[ITS_MAT_L];
    const int n = L.n_rows;
    if (n < 64 || L.n_cols != n || L.n_off < 2 || L.n_off > 3 || !L.off_dict) return ITS_OK;
    int dict[256];
    if (hipMemcpy(dict, L.off_dict, sizeof dict, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return ITS_OK; }
    std::sort(dict, dict + L.n_off);                          // most negative first
    if (dict[L.n_off - 1] != -1) return ITS_OK;
    const int64_t a = -(int64_t)dict[L.n_off - 2], b = L.n_off == 3 ? -(int64_t)dict[0] : 0;
    if (a < 4 || (a & 3) || n % a || a > ((int64_t)1 << 22)) return ITS_OK;     // (lines of more than 4 M rows: the step tables would be the size of the matrix)
    if (b && (b <= a || b % a || n % b)) return ITS_OK;
    LinesPlan P;
    P.n = n; P.a = (int)a; P.b = (int)b;
    P.ny = b ? (int)(b / a) : (int)(n / a);
    P.nz = b ? (int)(n / b) : 1;
    lines_plan_geometry(P);
    // measured (tools/lines_probe.py, tools/sweep_lines_ab.py): faster than the level schedule from 16 x 16 rows to 256^3
    // (its_lines.hip), so taken wherever the matrix qualifies (option values 1 and 2 alike; 0 = level schedule)
    int *flags = nullptr;
    auto give_up = [&]() { dfree(P.img); dfree(P.mimg); dfree(P.rimg); dfree(P.ximg); dfree(P.xedge); dfree(P.sbase); dfree(P.tabs); dfree(P.counter); dfree(flags);
                           h->err.clear(); (void)hipGetLastError(); return ITS_OK; };
    if (!h->D) return ITS_OK;
    if (lines_image_doubles(P) > ((size_t)1 << 36) || (int64_t)P.nby * P.nbz * P.tslots > (int64_t)1 << 30) return ITS_OK;
    if (dalloc(h, &P.img, lines_image_doubles(P)) || dalloc(h, &P.mimg, lines_image_masks(P)) || dalloc(h, &P.rimg, 4 * lines_image_masks(P)) ||
        dalloc(h, &P.ximg, 4 * lines_image_masks(P)) || dalloc(h, &P.xedge, lines_edge_doubles(P)) || dalloc(h, &P.counter, 8) || dalloc(h, &flags, 4)) return give_up();
    if (hipMemsetAsync(flags, 0, 4 * sizeof(int), h->stream) != hipSuccess || hipMemsetAsync(P.counter, 0, 8 * sizeof(int), h->stream) != hipSuccess ||
        hipMemsetAsync(P.xedge, 0, lines_edge_doubles(P) * sizeof(double), h->stream) != hipSuccess) return give_up();
    std::vector<int> sbase((size_t)P.nsteps + 1), tabs((size_t)4 * P.nsteps);
    lines_step_tables(P, sbase.data(), tabs.data());
    if (dalloc(h, &P.sbase, sbase.size()) || hipMemcpy(P.sbase, sbase.data(), sbase.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        dalloc(h, &P.tabs, tabs.size()) || hipMemcpy(P.tabs, tabs.data(), tabs.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return give_up();
    launch_lines_build(P, L, h->D, flags, h->stream);
    int hflags[4] = {1, 0, 0, 0};
    if (hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return give_up();
    if (hflags[0] & 1) return give_up();
    static const bool env_masked = getenv("ITS_LINES_MASKED") && atoi(getenv("ITS_LINES_MASKED")) != 0;
    P.masked = (hflags[0] & 2) != 0 || env_masked;
    dfree(flags);
    P.fail = P.counter + 4;
    P.fail_host = reinterpret_cast<unsigned long long *>(h->pinned + PIN_LINES);
    *reinterpret_cast<volatile unsigned long long *>(P.fail_host) = 0ull;
    P.mute_first_tile = h->lines.mute_first_tile;
    h->lines = P; h->lines_ready = true; h->lines_failed = false;
    if (getenv("ITS_DEBUG_LINES"))
        fprintf(stderr, "its: line-pipelined sweep: lines of %d rows, %d x %d lines, tiles of %d x %d lines (%d workgroups), chunks of %d rows\n",
                P.a, P.ny, P.nz, P.ty, P.tz, P.nby * P.nbz, P.ch);
    return ITS_OK;
}

// the sweep's "gave up" word (pinned memory, written by sweep_lines_kernel): checked wherever the host has just waited
static int check_lines_failure(its_handle *h)
{
    if (!h->lines_ready || !h->lines.fail_host) return ITS_OK;
    if (*reinterpret_cast<volatile unsigned long long *>(h->lines.fail_host) == 0ull) return ITS_OK;
    h->lines_failed = true;                                   // this handle stays on the level schedule from here on
    *reinterpret_cast<volatile unsigned long long *>(h->lines.fail_host) = 0ull;
    (void)hipMemsetAsync(h->lines.fail, 0, sizeof(int), h->stream);
    FAIL(h, ITS_ESTATE, "the line-pipelined triangular sweep gave up waiting for a neighbouring tile (is the GPU shared or hung?): "
                        "the iterate of this trip is invalid; option sweep_lines = 0 selects the level schedule");
}

// forward solve (D+L) x = rhs, level by level (kernels.hpp:451-473); x may alias rhs.  acc_out (may alias rhs): every row's
// chain sum is stored there (SpmvArgs::acc_out)
static int sweep(its_handle *h, double *x, const double *rhs, int first_level = 0, double *acc_out = nullptr)
{
    const CsrDev &L = h->mat[ITS_MAT_L];
    const bool dist = its_dist_active(h);
    // stencil matrices in natural order: the whole sweep as one launch pipelined over grid lines (its_lines.hip).  The
    // right-hand side is packed before the sweep starts and x unpacked after it ends, so x may be rhs (the in-place sweeps
    // of the preconditioner) and acc_out may be rhs
    if (h->lines_ready && h->sweep_lines && !h->lines_failed && !dist && first_level == 0 && h->D) {
        launch_sweep_lines(h->lines, x, rhs, acc_out, h->stream);
        HIPCHK(h, hipGetLastError());
        return ITS_OK;
    }
    for (int l = first_level; l < h->n_levels; ++l) {
        const int cnt = h->level_ptr[l + 1] - h->level_ptr[l];
        if (h->level_empty_lo[l] >= 0 && h->sweep_stream) {
            const int lo = h->level_empty_lo[l];
            launch_sweep_empty(x + lo, rhs + lo, h->D + lo, cnt, h->stream, acc_out ? acc_out + lo : nullptr);
        } else if (h->level_ell[l].val && h->sweep_ell_levels && h->sweep_stream) {
            launch_sweep_ell_view(L, h->level_ell[l], x, rhs, h->D, h->stream, acc_out);
        } else if (h->level_desc_count[l] > 0 && h->sweep_stream) {
            SpmvArgs a; a.x = x; a.y = x; a.b = rhs; a.D = h->D;   // rows of a level never read each other's x
            a.acc_out = acc_out;
            launch_spmv_exact_desc(L, h->sweep_desc + h->level_desc_first[l], h->level_desc_count[l], a, EPI_SWEEP, 0, h->stream);
        } else if (h->sw_w > 0 && h->sweep_ell && h->sweep_run && !dist && h->level_run_end[l] > l) {
            const int e = h->level_run_end[l];
            launch_sweep_ell_run(L.n_rows, h->sw_w, l, e, h->level_ptr_dev, h->level_rows, h->sw_col, h->sw_val, h->sw_D, x, rhs, h->stream, acc_out);
            l = e - 1;
            continue;
        } else if (h->sw_w > 0 && h->sweep_ell) {
            launch_sweep_ell_level(L.n_rows, h->sw_w, h->level_ptr[l], h->level_ptr[l + 1], h->level_rows, h->sw_col, h->sw_val,
                                   h->sw_D, x, rhs, h->stream, acc_out);
        } else {
            launch_sweep_level(L, h->level_rows + h->level_ptr[l], cnt, h->level_nshort[l], x, rhs, h->D, h->stream, acc_out);
        }
        // row-partitioned: colour l is done on every rank, its boundary values go to the neighbours before any rank
        // starts colour l + 1 (stream order on both sides of the exchange)
        if (dist) { int rc = its_dist_exchange(h, x, l); if (rc) return rc; }
    }
    HIPCHK(h, hipGetLastError());
    return ITS_OK;
}

extern "C" int its_solver_setup(its_handle *h, int solver_kind, int restart_len, int precond_kind,
                                int jacobi_mode, int gs_order)
{
    if (h) h->gm_ahead_iter = -1;
    if (!h) return ITS_EINVAL;
    if (solver_kind < ITS_JACOBI || solver_kind > ITS_GMRES) FAIL(h, ITS_EINVAL, "unknown solver kind");
    if (!h->residual_done) FAIL(h, ITS_ESTATE, "its_init_residual first (main.cpp:307 precedes :317)");
    HIPCHK(h, hipSetDevice(h->device));
    h->solver = solver_kind; h->precond = precond_kind; h->jacobi_mode = jacobi_mode; h->gs_order = gs_order;
    const int n = h->n;
    const size_t ne = (size_t)std::max(h->n_ext, n);
    const bool need_L = solver_kind == ITS_GAUSS_SEIDEL || (solver_kind == ITS_GMRES && precond_kind == ITS_PRECOND_GS);
    PhaseTimer pt("solver_setup");
    if (need_L) {
        const bool dist = its_dist_active(h);
        if (dist && its_dist_n_colors(h) == 0)
            FAIL(h, ITS_EUNSUPPORTED, "row-partitioned triangular sweeps need the global colour structure (its_dist_set_colors): "
                                      "the natural-order sweep is sequential across ranks");
        if (!h->have[ITS_MAT_L] || (solver_kind == ITS_GAUSS_SEIDEL && !h->have[ITS_MAT_U]))
            FAIL(h, ITS_ESTATE, "upload crs_L / crs_U first (gauss_seidel.hpp:433-450)");
        const CsrDev &L = h->mat[ITS_MAT_L];
        if (L.n_rows != n || (int)h->L_rp.size() != n + 1) FAIL(h, ITS_EINVAL, "crs_L must have the rows of A");
        int rc;
        if (dist) {
            // owned columns of a row must belong to earlier colours (strictly lower in the global colour-major order);
            // halo columns (>= n) are earlier colours of other ranks by the caller's construction
            const int *cp = its_dist_color_ptr(h);
            const int nc = its_dist_n_colors(h);
            for (int c = 0; c < nc; ++c)
                for (int i = cp[c]; i < cp[c + 1]; ++i)
                    if (h->L_maxcol[i] >= cp[c]) FAIL(h, ITS_EINVAL, "crs_L: a row reads an owned row of its own or a later colour");
            rc = build_levels(h, n, cp, nc);
        } else {
            if (h->L_has_halo_cols) FAIL(h, ITS_EINVAL, "crs_L must be strictly lower triangular (it has columns beyond its rows and no row partition is set up)");
            for (int i = 0; i < n; ++i)
                if (h->L_maxcol[i] >= i) FAIL(h, ITS_EINVAL, "crs_L must be strictly lower triangular");
            rc = build_levels(h, n);
        }
        if (rc) return rc;
        pt.lap("sweep schedule");
        if ((rc = build_lines_plan(h))) return rc;
        pt.lap("line-pipelined sweep plan");
        if (solver_kind == ITS_GAUSS_SEIDEL && (rc = build_gs_prefix(h))) return rc;
    }
    pt.lap("image of D + U (gs_prefix)");
    if (solver_kind == ITS_JACOBI && jacobi_mode == ITS_JACOBI_SHIPPED) {
        // The shipped sweep gathers from x_new, which is still all zeros on the first trip
        // (solver.hpp:225); x_old = x0 is overwritten by the swapped SpMV (jacobi.hpp:191)
        // before anything reads it.  x_a plays x_new here.
        HIPCHK(h, hipMemsetAsync(h->x_a, 0, ne * sizeof(double), h->stream));
    }
    if (solver_kind == ITS_CG) {                                     // cg.hpp:326-352
        dfree(h->p);
        if (dalloc(h, &h->p, ne)) return ITS_EHIP;
        HIPCHK(h, hipMemsetAsync(h->p, 0, ne * sizeof(double), h->stream));
        HIPCHK(h, hipMemcpyAsync(h->p, h->r, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        // rr = r0 . r0 (cg.hpp:148 computes it at the top of the first trip)
        launch_dot_partials(h->r, h->r, n, h->part[0], h->stream);
        h->rr_slot = 0;
        int rc = finish_scalar(h, h->part[0], h->sc + SC_RR0, FIN_SUM);
        if (rc) return rc;
        if (h->cg_variant == 1) {
            // pipelined CG: w = A r, s = z = p = 0, and the first pair of sums (r,r), (w,r)
            dfree(h->pcg_w); dfree(h->pcg_s); dfree(h->pcg_z); dfree(h->pcg_pair[0]); dfree(h->pcg_pair[1]);
            if (dalloc(h, &h->pcg_w, ne) || dalloc(h, &h->pcg_s, ne) || dalloc(h, &h->pcg_z, ne) ||
                dalloc(h, &h->pcg_pair[0], 2 * ITS_NPART) || dalloc(h, &h->pcg_pair[1], 2 * ITS_NPART)) return ITS_EHIP;
            for (double *v : {h->pcg_w, h->pcg_s, h->pcg_z, h->p}) HIPCHK(h, hipMemsetAsync(v, 0, ne * sizeof(double), h->stream));
            SpmvArgs a; a.x = h->r; a.y = h->pcg_w;
            if ((rc = solver_spmv(h, a, EPI_STORE))) return rc;
            launch_pcg_dots(h->r, h->pcg_w, n, h->pcg_pair[0], h->stream);
            if (its_dist_active(h) && (rc = its_dist_allreduce(h, h->pcg_pair[0], 2 * ITS_NPART, 0))) return rc;
            h->pcg_trip = 0;
        }
    } else if (solver_kind == ITS_GMRES) {                           // gmres.hpp:672-786
        if (restart_len < 1) FAIL(h, ITS_EINVAL, "restart_len < 1");
        if (restart_len > (1 << 24)) FAIL(h, ITS_EINVAL, "restart_len beyond 2^24");
        const int m = h->m = restart_len;
        { int rcp = size_pinned(h, m); if (rcp) return rcp; }
        dfree(h->V); dfree(h->hcol_dev); dfree(h->y_dev); dfree(h->gm_w2);
        if (dalloc(h, &h->V, (((size_t)ne + 31) & ~(size_t)31) * (size_t)(m + 1)) || dalloc(h, &h->hcol_dev, m + 2) || dalloc(h, &h->y_dev, m + 1) ||
            dalloc(h, &h->gm_w2, ne)) return ITS_EHIP;
        HIPCHK(h, hipMemsetAsync(h->gm_w2, 0, (size_t)ne * sizeof(double), h->stream));
        h->gm_ahead_iter = -1;
        h->Rm.assign((size_t)(m + 1) * m, 0.0); h->g.assign(m + 1, 0.0);
        h->cs.assign(m, 0.0); h->sn.assign(m, 0.0); h->hcol.assign(m + 2, 0.0);
        if (const char *e = getenv("ITS_GMRES_GIVENS")) h->gmres_givens = !strcmp(e, "dense") || !strcmp(e, "1");
        if (h->gmres_givens == 1) {
            h->Hd.assign((size_t)(m + 1) * m, 0.0); h->Ht = h->Hd; h->Rd = h->Hd; h->dense_full = false;
            h->Qd.assign((size_t)(m + 1) * (m + 1), 0.0); h->Jd = h->Qd; h->Qc = h->Qd;
            for (int i = 0; i <= m; ++i) h->Qd[(size_t)i * (m + 1) + i] = 1.0;
        }
        if (h->gmres_ortho == 1) {
            if (m > 4000) FAIL(h, ITS_EUNSUPPORTED, "gmres_ortho = 1 (CGS-2) keeps the projection coefficients in LDS: restart_len <= 4000");
            dfree(h->gram_part); dfree(h->gram_h);
            if (dalloc(h, &h->gram_part, (size_t)(m + 2) * gram_blocks()) || dalloc(h, &h->gram_h, (size_t)2 * (m + 2))) return ITS_EHIP;
        }
        // persistent Gram-Schmidt launch (its_mgs.hip): one GPU, modified Gram-Schmidt, vectors that fit one resident grid
        dfree(h->mgs_slots); h->mgs_grid = 0; h->mgs_fallbacks = 0;
        if (h->gmres_persist && h->gmres_ortho == 0 && !its_dist_active(h) && n >= h->gmres_persist_min_n) {
            h->mgs_grid = mgs_persist_grid(n, h->device);
            // (its slots grow with restart length x workgroups: beyond 256 MB, or if they cannot be had, the per-step kernels)
            if (h->mgs_grid > 0 && mgs_persist_slot_words(m, h->mgs_grid) > ((size_t)256 << 20) / sizeof(unsigned long long)) h->mgs_grid = 0;
            if (h->mgs_grid > 0 && dalloc(h, &h->mgs_slots, mgs_persist_slot_words(m, h->mgs_grid))) { h->mgs_grid = 0; h->err.clear(); (void)hipGetLastError(); }
            if (h->mgs_grid > 0) HIPCHK(h, mgs_persist_reset(h->mgs_slots, m, h->mgs_grid, h->stream));
        }
        if (getenv("ITS_DEBUG_GMRES")) fprintf(stderr, "its_solver_setup(GMRES): Gram-Schmidt %s (%d workgroups), n = %d, restart %d\n", h->mgs_grid > 0 ? "as one persistent launch per trip" : "as one launch per basis vector", h->mgs_grid, n, m);
        h->restart_count = 0;
        // beta = ||r||, V[0] = r * (1/beta)   (gmres.hpp:734-736,768-771)
        launch_dot_partials(h->r, h->r, n, h->part[0], h->stream);
        int rc = finish_scalar(h, h->part[0], h->sc + SC_TMP, FIN_SUM);
        if (rc) return rc;
        launch_mgs_normalize(h->V, h->r, h->sc + SC_TMP, 1, h->hcol_dev, n, h->stream);
        HIPCHK(h, hipMemcpyAsync(h->pinned, h->hcol_dev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->beta = h->pinned[0];
        std::fill(h->g.begin(), h->g.end(), 0.0); h->g[0] = h->beta;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    h->setup_done = true;
    h->last_residual = 0.0;
    h->ahead = false;
    h->watch_misses = 0;
    return ITS_OK;
}

// ---------------------------------------------------------------------------------------
// Scalars the host needs every trip are written by the trip's last kernel straight into pinned host memory
// (GPU-visible on ROCm) and the host watches the word instead of asking the runtime for the stream's completion
// signal: the word is first set to a NaN pattern no kernel here can produce.  hipStreamSynchronize remains the
// way out if the value does not arrive (a failed launch is reported there).
// ---------------------------------------------------------------------------------------
constexpr uint64_t kPendingWord = 0x7ff8dead0badf00dull;
static inline void arm_pinned_word(double *slot) { *reinterpret_cast<volatile uint64_t *>(slot) = kPendingWord; }
// Returns once the watched word has been written by the trip's last kernel.  What that does and does not mean: the
// kernels of the trip that ran BEFORE the last one are complete (stream order), and so are their writes to pinned
// memory (the GMRES H-column entries above the watched one); OTHER workgroups of the last kernel may still be
// running.  That is safe because the host touches nothing those workgroups write: every later access to device
// vectors is queued on the same stream behind that kernel (its_download_x, the next trip, its_synchronize).  The
// acquire fence orders the host's reads of the other pinned words after the read that saw the word arrive.
static int wait_pinned_word(its_handle *h, const double *slot_d)
{
    const volatile uint64_t *slot = reinterpret_cast<const volatile uint64_t *>(slot_d);
    bool arrived = false;
    if (g_readback_mode == 0 && h->watch_misses < 2) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0; !arrived; ++spins) {
            arrived = __atomic_load_n(const_cast<const uint64_t *>(reinterpret_cast<const volatile uint64_t *>(slot)), __ATOMIC_ACQUIRE) != kPendingWord;
            if (!arrived && (spins & 1023) == 1023 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(h->watch_window_ms)) break;
        }
        // a trip longer than the watch window (huge systems, a busy GPU): after two misses this handle synchronizes
        // straight away instead of paying the window every trip
        if (!arrived) ++h->watch_misses;
    }
    if (!arrived) HIPCHK(h, hipStreamSynchronize(h->stream));
    std::atomic_thread_fence(std::memory_order_acquire);
    return ITS_OK;
}

// leading dimension of the Krylov basis V: the vector length (owned + halo entries) rounded up to 32 doubles, so
// that every basis vector starts 256-byte aligned and the vector kernels keep their two-elements-per-lane path
static inline size_t krylov_ld(const its_handle *h) { return ((size_t)std::max(h->n_ext, h->n) + 31) & ~(size_t)31; }

// ---------------------------------------------------------------------------------------
// one trip of each method, enqueued on h->stream; the residual scalar lands in res_out (device)
// ---------------------------------------------------------------------------------------
static int trip_cg(its_handle *h, double *res_out)
{
    const int n = h->n;
    double *rr_in = h->sc + (h->rr_slot ? SC_RR1 : SC_RR0);
    double *rr_out = h->sc + (h->rr_slot ? SC_RR0 : SC_RR1);
    SpmvArgs a; a.x = h->p; a.y = h->tmp; a.dotv = h->p; a.part_sum = h->part[0];
    const bool prof = h->prof_on && 2 * h->prof_n + 1 < (int)h->prof_ev.size();
    if (prof) HIPCHK(h, hipEventRecord(h->prof_ev[2 * h->prof_n], h->stream));
    int rc = solver_spmv(h, a, EPI_STORE);                            // cg.hpp:128,160
    if (rc) return rc;
    if (prof) { HIPCHK(h, hipEventRecord(h->prof_ev[2 * h->prof_n + 1], h->stream)); ++h->prof_n; }
    if ((rc = allreduce_partials(h, h->part[0], 0))) return rc;       // p.Ap over all ranks
    launch_cg_update_r(h->r, h->tmp, n, h->part[0], 0, rr_in, h->sc, h->part[1], h->stream);
    if ((rc = allreduce_partials(h, h->part[1], 0))) return rc;       // r.r over all ranks
    launch_cg_update_xp(h->x_a, h->p, h->r, n, h->part[1], 0, rr_in, rr_out, h->sc, res_out, h->stream);
    h->rr_slot ^= 1;
    HIPCHK(h, hipGetLastError());
    return ITS_OK;
}

// One trip of pipelined CG ("cg_variant" 1; its_krylov_cr.hip): q = A w, then ONE fused vector pass, then ONE
// reduction point (the pair of partial arrays is all-reduced together when row-partitioned).  The recorded value is
// ||r_new||_2 as in the reference loop (utility_funcs.hpp:483).
static int trip_cg_pipelined(its_handle *h, double *res_out)
{
    const int n = h->n;
    const int par = h->pcg_trip & 1;
    SpmvArgs a; a.x = h->pcg_w; a.y = h->tmp;
    int rc = solver_spmv(h, a, EPI_STORE);
    if (rc) return rc;
    double *sc_in = h->sc + (par ? SC_PCG1 : SC_PCG0), *sc_out = h->sc + (par ? SC_PCG0 : SC_PCG1);
    launch_pcg_update(h->x_a, h->r, h->pcg_w, h->p, h->pcg_s, h->pcg_z, h->tmp, n, h->pcg_pair[par], h->pcg_pair[par ^ 1],
                      sc_in, sc_out, h->pcg_trip == 0, h->stream);
    if (its_dist_active(h) && (rc = its_dist_allreduce(h, h->pcg_pair[par ^ 1], 2 * ITS_NPART, 0))) return rc;
    launch_reduce_finalize(h->pcg_pair[par ^ 1], res_out, FIN_SQRT, h->stream);
    ++h->pcg_trip;
    HIPCHK(h, hipGetLastError());
    return ITS_OK;
}

static int trip_jacobi(its_handle *h, double *res_out)
{
    // x_a = x_old (and, in SHIPPED mode, the reference's x_new before the trip: after
    // exchange_arrays both hold the same values, solver.hpp:796-798); x_b receives the new iterate.
    SpmvArgs a; a.y = h->x_b; a.b = h->b; a.D = h->D;
    int rc;
    if (h->jacobi_mode != ITS_JACOBI_SHIPPED && h->fused_residual) {
        // One pass over A per trip.  The pass that records ||b - A x_new|| walks the same rows of A that the
        // next trip's sweep of x_new needs, so it produces that sweep too (second in-order chain, same bits as
        // a launch of its own) into the spare vector; the next trip starts with its iterate already there.
        if (!h->ahead) {
            a.x = h->x_a;
            if ((rc = solver_spmv(h, a, EPI_JACOBI_REF))) return rc;
        }
        SpmvArgs f; f.x = h->x_b; f.y = h->x; f.b = h->b; f.part_sum = h->part[0];
        if ((rc = solver_spmv(h, f, EPI_JACOBI_FUSED))) return rc;
        if ((rc = finish_scalar(h, h->part[0], res_out, FIN_SQRT))) return rc;
        double *spare = h->x_a;
        h->x_a = h->x_b;                                              // exchange_arrays: the new iterate
        h->x_b = h->x;                                                // the iterate after it
        h->x = spare;
        h->ahead = true;
        return ITS_OK;
    }
    if (h->jacobi_mode == ITS_JACOBI_SHIPPED && h->fused_residual) {
        // The shipped sweep's chain IS the full row (jacobi.hpp:41-62 works on A x): the pass that sweeps x_new into the
        // iterate after it walks exactly the chain of ||b - A x_new||, so it records that norm as well — one pass over A
        // and one chain per trip, same bits as the sweep and the residual pass apart.
        if (!h->ahead) {
            a.x = h->x_a;
            if ((rc = solver_spmv(h, a, EPI_JACOBI_SHIPPED))) return rc;
        }
        SpmvArgs f; f.x = h->x_b; f.y = h->x; f.b = h->b; f.D = h->D; f.part_sum = h->part[0];
        if ((rc = solver_spmv(h, f, EPI_JACOBI_SHIPPED))) return rc;
        if ((rc = finish_scalar(h, h->part[0], res_out, FIN_SQRT))) return rc;
        double *spare = h->x_a;
        h->x_a = h->x_b;                                              // exchange_arrays: the new iterate
        h->x_b = h->x;                                                // the iterate after it
        h->x = spare;
        h->ahead = true;
        return ITS_OK;
    }
    if (h->jacobi_mode == ITS_JACOBI_SHIPPED) {
        a.x = h->x_a;
        rc = solver_spmv(h, a, EPI_JACOBI_SHIPPED);
    } else {
        a.x = h->x_a;
        rc = solver_spmv(h, a, EPI_JACOBI_REF);
    }
    if (rc) return rc;
    // record_residual_norm: r = b - A x_new, 2-norm (utility_funcs.hpp:423-426)
    // (only the norm is recorded: the residual vector itself is not stored, y = nullptr)
    SpmvArgs q; q.x = h->x_b; q.y = nullptr; q.b = h->b; q.part_sum = h->part[0];
    if ((rc = solver_spmv(h, q, EPI_RESIDUAL))) return rc;
    if ((rc = finish_scalar(h, h->part[0], res_out, FIN_SQRT))) return rc;
    std::swap(h->x_a, h->x_b);                                        // exchange_arrays
    return ITS_OK;
}

static int trip_gs(its_handle *h, double *res_out)
{
    SpmvArgs u; u.x = h->x; u.y = h->tmp; u.b = h->b;
    int rc;
    // crs_U is the strictly upper part of A (gauss_seidel.hpp:433-450 builds it so; the entry counts are
    // compared): the residual pass over A then also yields b - U x_new for the next trip (EPI_GS_FUSED)
    // (not for matrices on the irregular-row kernel: its one-lane-per-row walk out of LDS is issue-bound, and the
    // per-entry column test of the second chain costs more there than the separate pass over U: 2.7 vs 2.3 ms on
    // the power-law matrix in multicolour order; with the panel image of A the fused pass wins again, and so it does where
    // the hub rows are on the wave-per-row launch, whose second chain is free: its_solve -gs, 200 000-row power-law matrix,
    // 0.63 -> 0.56 ms per trip)
    const CsrDev &Am = h->mat[ITS_MAT_A], &Um = h->mat[ITS_MAT_U];
    // option "gs_prefix" (its_handle.h): the fused pass walks D + U only, its full chain starting from the sums the sweep left
    // in tmp — the entries of L are read once per trip, by the sweep (power-law matrix, 10 M rows: the pass over A was 1.42 ms
    // of the 2.47 ms trip)
    const bool prefix = h->fused_residual && h->gs_prefix_ready && h->have[ITS_MAT_DU] && !its_dist_active(h);
    const CsrDev &Fm = prefix ? h->mat[ITS_MAT_DU] : Am;
    const bool fused = h->fused_residual && Um.nnz == Am.n_upper && Um.n_upper == Um.nnz &&
                       Um.upper_col_sum == Am.upper_col_sum && Um.upper_val_xor == Am.upper_val_xor &&
                       (Fm.max_row_len <= IRR_ROW_LEN || spmv_panel_eligible(Fm, u, EPI_GS_FUSED) || Fm.irr_n_long > 0) && !its_dist_active(h);
    double *acc_io = (prefix && fused) ? h->tmp : nullptr;
    // first sweep level without entries in L (the first colour): its rows' next iterate is (b - U x) / D, which the fused
    // pass can store itself (ELL kernels) — into x_b, because other rows of that pass still read the current x
    const int lo0 = h->n_levels >= 2 ? h->level_empty_lo[0] : -1, hi0 = lo0 >= 0 ? lo0 + (h->level_ptr[1] - h->level_ptr[0]) : -1;
    SpmvArgs f; f.x = h->x; f.y = h->tmp; f.b = h->b; f.part_sum = h->part[0];
    const bool pre = fused && h->gs_first_level_fused && h->sweep_stream && lo0 >= 0 && (lo0 & 1) == 0 &&
                     ((hi0 & 1) == 0 || hi0 == Am.n_rows) && spmv_ell_eligible(Fm, f, EPI_GS_FUSED) && Fm.ell_kind != 3;
    if (!(fused && h->ahead)) {
        if ((rc = spmv_dispatch(h, h->mat[ITS_MAT_U], u, EPI_GS_RHS, ITS_FMT_CSR_EXACT))) return rc;   // gauss_seidel.hpp:144,177
        h->ahead_pre = false;
    }
    if (h->ahead_pre) {
        if ((rc = sweep(h, h->x_b, h->tmp, 1, acc_io))) return rc;                     // level 0 is already there
        std::swap(h->x, h->x_b);
        h->ahead_pre = false;
    } else if ((rc = sweep(h, h->x, h->tmp, 0, acc_io))) return rc;                    // :227
    if (fused) {
        f.x = h->x;
        if (pre) { f.y2 = h->x_b; f.y2_lo = lo0; f.y2_hi = hi0; f.D = h->D; }
        f.acc0 = acc_io;                                                               // (read and overwritten row by row: f.y is tmp too)
        if ((rc = spmv_dispatch(h, Fm, f, EPI_GS_FUSED, ITS_FMT_CSR_EXACT))) return rc;
        h->ahead = true; h->ahead_pre = pre;
        return finish_scalar(h, h->part[0], res_out, FIN_SQRT);
    }
    SpmvArgs q; q.x = h->x; q.y = nullptr; q.b = h->b; q.part_sum = h->part[0];         // norm only
    if ((rc = solver_spmv(h, q, EPI_RESIDUAL))) return rc;                             // utility_funcs.hpp:444-447
    return finish_scalar(h, h->part[0], res_out, FIN_SQRT);
}

// GMRES trip: device part (SpMV + MGS) then the host Givens update.  Needs a host sync.
static int trip_gmres(its_handle *h, int iter_count, double *res_host)
{
    const int n = h->n, m = h->m;
    const size_t ld = krylov_ld(h);
    const int k = iter_count - h->restart_count * m;                  // gmres.hpp:113
    if (k < 0 || k >= m) FAIL(h, ITS_ESTATE, "GMRES inner index out of range: restart missing?");
    double *w = h->tmp;
    int rc;
    if (h->gm_ahead_iter == iter_count && h->gm_ahead_restart == h->restart_count && h->gm_ahead_w) {
        w = h->gm_ahead_w;                                            // queued behind the last trip's sweep (below): same kernels, same operands
    } else {
        SpmvArgs a; a.x = h->V + (size_t)k * ld; a.y = w;
        if ((rc = solver_spmv(h, a, EPI_STORE))) return rc;           // gmres.hpp:234
        if (h->precond == ITS_PRECOND_GS) { if ((rc = sweep(h, w, w))) return rc; }   // gmres.hpp:251
    }
    h->gm_ahead_iter = -1;
    // the next trip's product, to be queued as soon as this trip's sweep is: v_{k+1} is the sweep's last write, nothing the
    // host does with the Hessenberg column changes it.  Not at the end of a cycle (the restart rebuilds V), not across ranks
    // (the halo exchange is a collective every rank would have to queue alike: kept out of a speculation)
    auto queue_ahead = [&]() -> int {
        if (!h->gmres_ahead || k + 1 >= m || its_dist_active(h) || !h->gm_w2) return ITS_OK;
        double *wn = (w == h->tmp) ? h->gm_w2 : h->tmp;
        SpmvArgs a2; a2.x = h->V + (size_t)(k + 1) * ld; a2.y = wn;
        int rc2 = solver_spmv(h, a2, EPI_STORE);
        if (rc2) return rc2;
        if (h->precond == ITS_PRECOND_GS && (rc2 = sweep(h, wn, wn))) return rc2;
        h->gm_ahead_w = wn; h->gm_ahead_iter = iter_count + 1; h->gm_ahead_restart = h->restart_count;
        return ITS_OK;
    };
    // the Hessenberg column is written straight into pinned host memory (region pin_h, m + 2 doubles), or copied
    // (the persistent Gram-Schmidt launch always writes the column to pinned memory — a launch that gave up is recognised
    // by the word it left armed; in read-back mode 2 the host then synchronizes before it reads, which is what that mode
    // is for, so every mode runs the same kernels and returns the same bits)
    const bool use_copy = readback_mode() == 2 && h->mgs_grid <= 0;
    const bool hcol_pinned = !use_copy;
    double *hcol_out = hcol_pinned ? h->pinned + h->pin_h : h->hcol_dev;
    // Every entry of the column is armed and waited for, not only the last one.  The entries come from k + 2 different
    // kernels, and the host does NOT see their writes to pinned memory in launch order: one its_solve -gm run on lap3d_16
    // printed 2.38e-6 for the reference's 1.75e-6 at trip 36, and tools/gmres_stale_entry_probe.py reproduces exactly that
    // number on the CPU by giving entry k - 1 of column k (written by the third-last kernel of the trip) the value the
    // slot held in column k - 1 — no other single substitution does (profiles/r03_gmres_stale_probe.txt).
    // Stores of different kernels to host memory are posted writes that may pass each other on the way; an arrival word
    // per entry needs no ordering at all.
    if (hcol_pinned) for (int i = 0; i <= k + 1; ++i) arm_pinned_word(hcol_out + i);
    // the whole sweep as one persistent launch with w in registers (its_mgs.hip) when the set-up found it applicable;
    // w in memory is not modified by it, so a launch that gave up (grid not co-resident) is repeated by the per-step kernels
    bool persist = h->mgs_grid > 0 && hcol_pinned;
    if (persist) {
        HIPCHK(h, launch_mgs_persist(w, h->V, (int64_t)ld, k + 1, h->V + (size_t)(k + 1) * ld, hcol_out, h->mgs_slots, h->mgs_parity++, m, h->mgs_grid, n, h->stream,
                                     h->mgs_inject_fail > 0 ? (h->mgs_inject_fail--, h->mgs_inject_kind) : 0));
        if ((rc = queue_ahead())) return rc;
        if ((rc = wait_pinned_word(h, hcol_out + (k + 1)))) return rc;
        if (*reinterpret_cast<const volatile uint64_t *>(hcol_out + (k + 1)) == kPendingWord) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (*reinterpret_cast<const volatile uint64_t *>(hcol_out + (k + 1)) == kPendingWord) {
                persist = false;
                h->gm_ahead_iter = -1;                                 // (what was queued read a v_{k+1} that was never written)
                HIPCHK(h, mgs_persist_reset(h->mgs_slots, m, h->mgs_grid, h->stream));
                if (h->watch_misses > 0) --h->watch_misses;            // the window that ran out was this launch's, not a slow trip's
                if (++h->mgs_fallbacks >= 2) h->mgs_grid = 0;          // twice: this handle stays on the per-step kernels
                for (int i = 0; i <= k + 1; ++i) arm_pinned_word(hcol_out + i);
            }
        }
    }
    const double *prev_part = nullptr; int prev_scalar = 0;
    if (persist) {
        // everything below already happened inside the launch
    } else if (h->gmres_ortho == 1) {
        // CGS-2 ("gmres_ortho" 1): h1 = V^T w, w -= V h1, h2 = V^T w, w -= V h2, H[:,k] = h1 + h2 — every product
        // batched over j, one reduction per pass and one for ||w||: 3 per trip instead of k + 2
        const int nv = k + 1;
        double *h1 = h->gram_h, *h2 = h->gram_h + (m + 2);
        for (int pass = 0; pass < 2; ++pass) {
            double *hp = pass ? h2 : h1;
            launch_gram_dots(w, h->V, (int64_t)ld, nv, n, h->gram_part, h->stream);
            launch_gram_finalize(h->gram_part, nv, hp, h->stream);
            if (its_dist_active(h) && (rc = its_dist_allreduce(h, hp, nv, 0))) return rc;
            launch_gram_axpy(w, h->V, (int64_t)ld, nv, hp, n, pass ? h->part[0] : nullptr, h->stream);
        }
        launch_gram_combine(h1, h2, nv, hcol_out, h->stream);
        if ((rc = allreduce_partials(h, h->part[0], 0))) return rc;
        prev_part = h->part[0];
    } else
    // MGS (gmres.hpp:269-329): step j finishes w -= h_{j-1} v_{j-1} and starts h_j = w . v_j
    for (int j = 0; j <= k + 1; ++j) {
        double *next_part = h->part[j & 1];
        const double *v_prev = j > 0 ? h->V + (size_t)(j - 1) * ld : nullptr;
        const double *v_next = j <= k ? h->V + (size_t)j * ld : nullptr;   // j == k+1: w . w
        launch_mgs_step(w, v_prev, prev_part, prev_scalar, v_next, next_part,
                        j > 0 ? hcol_out + (j - 1) : nullptr, n, h->stream);
        prev_part = next_part; prev_scalar = 0;
        if ((rc = allreduce_partials(h, next_part, 0))) return rc;
    }
    // H[k+1,k] = ||w||, V[k+1] = w / H[k+1,k]   (gmres.hpp:359,389)
    if (!persist) {
        launch_mgs_normalize(h->V + (size_t)(k + 1) * ld, w, prev_part, prev_scalar, hcol_out + (k + 1), n, h->stream);
        if ((rc = queue_ahead())) return rc;
    }
    HIPCHK(h, hipGetLastError());
    double *hc = h->hcol.data();
    if (hcol_pinned) {
        // H[k+1,k] comes from the last kernel of the trip; the entries above it from kernels that finished before
        if ((rc = wait_pinned_word(h, hcol_out + (k + 1)))) return rc;
        for (int i = 0; i <= k; ++i)                                       // (arrived long ago in all but the rare case)
            if (*reinterpret_cast<const volatile uint64_t *>(hcol_out + i) == kPendingWord && (rc = wait_pinned_word(h, hcol_out + i))) return rc;
        std::atomic_thread_fence(std::memory_order_acquire);
        // wait_pinned_word ends in a stream synchronize when a word does not arrive: an entry that is still armed after
        // that was never written — an error of this library, reported instead of entering H as a NaN
        for (int i = 0; i <= k + 1; ++i)
            if (*reinterpret_cast<const volatile uint64_t *>(hcol_out + i) == kPendingWord) {
                HIPCHK(h, hipStreamSynchronize(h->stream));
                if (*reinterpret_cast<const volatile uint64_t *>(hcol_out + i) == kPendingWord)
                    FAIL(h, ITS_ESTATE, "GMRES: an entry of the Hessenberg column was never written to the read-back buffer");
            }
        for (int i = 0; i <= k + 1; ++i) hc[i] = hcol_out[i];
    } else {
        HIPCHK(h, hipMemcpyAsync(h->pinned + h->pin_copy, h->hcol_dev, (size_t)(k + 2) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i <= k + 1; ++i) hc[i] = h->pinned[h->pin_copy + i];
    }
    if (h->gmres_givens == 1) {
        // ---- the reference's formulation, product for product (gmres.hpp:427-658): J = I with
        // one 2x2 block, H_tmp = Q H, Q <- J Q, R = Q H, g = Q (beta e1), every entry a fused
        // chain over the full (m+1) range.  O(m^3) per trip on the host, but its rounding — in
        // particular the stagnation of |g[k+1]| at ~eps*beta once converged — is the reference's,
        // so iteration counts at TOL below the attainable accuracy match the shipped binary.
        const int m1 = m + 1;
        auto mm = [](const double *A, const double *B, double *C, int nrA, int ncA, int ncB) {
            for (int i = 0; i < nrA; ++i)
                for (int j = 0; j < ncB; ++j) {
                    double t = 0.0;
                    for (int q = 0; q < ncA; ++q) t = std::fma(A[(size_t)i * ncA + q], B[(size_t)q * ncB + j], t);
                    C[(size_t)i * ncB + j] = t;
                }
        };
        for (int i = 0; i <= k + 1; ++i) h->Hd[(size_t)i * m + k] = hc[i];          // H row-major (m+1) x m, column k
        // The products as the reference writes them are O(m^3) per trip; almost all of their chains repeat last trip's
        // operands or multiply by rows of the identity.  What can change is computed, with the reference's own chains
        // (every term, in order), the rest is kept: J differs from I in rows k, k+1 only, so Q <- J Q rewrites those two
        // rows (the chain of any other row is fma(0, ., t) ... fma(1, Q_ij, +0) ... = Q_ij); of H_tmp = Q H only entries
        // (k, k) and (k+1, k) are read; R = Q H changes in rows k, k+1 (new rows of Q) and in column k (new column of H).
        // O(m^2) per trip, the same bits (tests/test_gpu_state.py runs both; ITS_GMRES_DENSE_FULL=1 keeps the full
        // products).  A non-finite entry (fma(0, inf) = NaN would spread through the full products) switches the rest of
        // the cycle to them.
        static const bool always_full = getenv("ITS_GMRES_DENSE_FULL") != nullptr;
        bool finite = true;
        for (int i = 0; i <= k + 1; ++i) finite = finite && std::isfinite(hc[i]);
        if (!finite) h->dense_full = true;
        auto chain = [&](const double *Arow, const double *Bcol, int stride) {      // sum_q A[q] * B[q * stride], q = 0 .. m
            double t = 0.0;
            for (int q = 0; q < m1; ++q) t = std::fma(Arow[q], Bcol[(size_t)q * stride], t);
            return t;
        };
        if (always_full || h->dense_full) {
            std::fill(h->Jd.begin(), h->Jd.end(), 0.0);
            for (int i = 0; i < m1; ++i) h->Jd[(size_t)i * m1 + i] = 1.0;
            if (k == 0) h->Ht = h->Hd; else mm(h->Qd.data(), h->Hd.data(), h->Ht.data(), m1, m1, m);
            const double hkk = h->Ht[(size_t)k * m + k], hk1 = h->Ht[(size_t)(k + 1) * m + k];
            const double denom = std::sqrt(std::fma(hkk, hkk, hk1 * hk1));
            const double c = hkk / denom, s = hk1 / denom;
            h->Jd[(size_t)k * m1 + k] = c;            h->Jd[(size_t)k * m1 + k + 1] = s;
            h->Jd[(size_t)(k + 1) * m1 + k] = -1.0 * s; h->Jd[(size_t)(k + 1) * m1 + k + 1] = c;
            mm(h->Jd.data(), h->Qd.data(), h->Qc.data(), m1, m1, m1);
            h->Qd = h->Qc;
            mm(h->Qd.data(), h->Hd.data(), h->Rd.data(), m1, m1, m);
        } else {
            const double hkk = k == 0 ? h->Hd[(size_t)k * m + k] : chain(&h->Qd[(size_t)k * m1], &h->Hd[k], m);
            const double hk1 = k == 0 ? h->Hd[(size_t)(k + 1) * m + k] : chain(&h->Qd[(size_t)(k + 1) * m1], &h->Hd[k], m);
            const double denom = std::sqrt(std::fma(hkk, hkk, hk1 * hk1));
            const double c = hkk / denom, s = hk1 / denom;
            if (!std::isfinite(c) || !std::isfinite(s)) h->dense_full = true;   // (this trip is still what the full products give: NaN rows k, k+1; from the next one on it spreads)
            // rows k, k+1 of J Q: terms q = k, k+1 of the chain, in that order; every other term adds an exact zero
            double *Qk = &h->Qd[(size_t)k * m1], *Qk1 = &h->Qd[(size_t)(k + 1) * m1];
            const double ms = -1.0 * s;
            for (int j = 0; j < m1; ++j) {
                const double a0 = Qk[j], a1 = Qk1[j];
                Qk[j] = std::fma(s, a1, std::fma(c, a0, 0.0));
                Qk1[j] = std::fma(c, a1, std::fma(ms, a0, 0.0));
            }
            // R = Q H: rows k, k+1 for the columns filled so far, column k for every row
            for (int j = 0; j <= k; ++j) {
                h->Rd[(size_t)k * m + j] = chain(Qk, &h->Hd[j], m);
                h->Rd[(size_t)(k + 1) * m + j] = chain(Qk1, &h->Hd[j], m);
            }
            for (int i = 0; i < m1; ++i) h->Rd[(size_t)i * m + k] = chain(&h->Qd[(size_t)i * m1], &h->Hd[k], m);
        }
        for (int i = 0; i < m1; ++i) h->g[i] = h->Qd[(size_t)i * m1] * h->beta;   // Q (beta e1): the chain's other terms are exact zeros
        for (int c2 = 0; c2 <= k; ++c2)                                           // keep Rm (col-major) for gmres_get_x
            for (int i = 0; i <= c2; ++i) h->Rm[(size_t)c2 * m1 + i] = h->Rd[(size_t)i * m + c2];
        *res_host = std::fabs(h->g[k + 1]);
        return ITS_OK;
    }
    // ---- default: Givens least squares (gmres.hpp:427-658) as O(k) rotation updates.  The
    // reference forms Q = J_k...J_0 and R = Q H with dense products; applying the stored
    // rotations to the new column gives the same R[:,k] and g up to round-off.
    for (int i = 0; i < k; ++i) {
        const double t0 = h->cs[i] * hc[i] + h->sn[i] * hc[i + 1];
        const double t1 = -h->sn[i] * hc[i] + h->cs[i] * hc[i + 1];
        hc[i] = t0; hc[i + 1] = t1;
    }
    const double denom = std::sqrt(hc[k] * hc[k] + hc[k + 1] * hc[k + 1]);   // gmres.hpp:484-485
    const double c = hc[k] / denom, s = hc[k + 1] / denom;                   // :487-488
    h->cs[k] = c; h->sn[k] = s;
    hc[k] = c * hc[k] + s * hc[k + 1]; hc[k + 1] = 0.0;
    for (int i = 0; i <= k; ++i) h->Rm[(size_t)k * (m + 1) + i] = hc[i];
    h->g[k + 1] = -s * h->g[k];
    h->g[k] = c * h->g[k];
    *res_host = std::fabs(h->g[k + 1]);                                      // :658
    return ITS_OK;
}

// gmres_get_x (gmres.hpp:845-978): back-substitute on the host, x = x0 + V^T y on the device
static int gmres_get_x(its_handle *h, double *x_out, const double *x0, int iter_count)
{
    const int n = h->n, m = h->m;
    const size_t ld = krylov_ld(h);
    const int k = iter_count - h->restart_count * m;                  // :880
    std::vector<double> y(m + 1, 0.0);
    for (int row = k; row >= 0; --row) {                              // :901-916
        double sum = 0.0;
        for (int c = row + 1; c <= k; ++c) sum = std::fma(h->Rm[(size_t)c * (m + 1) + row], y[c], sum);
        y[row] = (h->g[row] - sum) / h->Rm[(size_t)row * (m + 1) + row];
    }
    const int terms = std::max(0, k + 1);
    for (int i = 0; i < m + 1; ++i) h->pinned[h->pin_y + i] = y[i];
    HIPCHK(h, hipMemcpyAsync(h->y_dev, h->pinned + h->pin_y, (size_t)(m + 1) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_gmres_update_x(x_out, x0, h->V, (int64_t)ld, h->y_dev, terms, n, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));   // pinned y buffer is reused
    return ITS_OK;
}

static int check_ready(its_handle *h)
{
    if (!h) return ITS_EINVAL;
    if (!h->setup_done) FAIL(h, ITS_ESTATE, "its_solver_setup first");
    if (hipSetDevice(h->device) != hipSuccess) FAIL(h, ITS_EHIP, "hipSetDevice");
    return ITS_OK;
}

extern "C" int its_iterate(its_handle *h, int iter_count, double *residual_norm)
{
    debug_poison_lds(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (!residual_norm) return ITS_EINVAL;
    if (h->solver == ITS_GMRES) {
        rc = trip_gmres(h, iter_count, residual_norm);
        if (rc == ITS_OK) h->last_residual = *residual_norm;
        return rc;
    }
    // the recorded value goes straight into pinned host memory and the host watches it (wait_pinned_word): this
    // returns ~6 us earlier per trip than a stream synchronize, and at 256^3 the next trip is queued while the tail
    // of this one drains.  ITS_READBACK_COPY=1 / ITS_READBACK_SYNC=1 select the device-slot copy / the plain
    // synchronize for A/B runs.
    const bool use_copy = readback_mode() == 2;
    double *slot = h->pinned + PIN_RES;
    double *res = use_copy ? h->sc + SC_RES : slot;
    if (!use_copy) arm_pinned_word(slot);
    if (h->solver == ITS_CG) rc = h->cg_variant == 1 ? trip_cg_pipelined(h, res) : trip_cg(h, res);
    else if (h->solver == ITS_JACOBI) rc = trip_jacobi(h, res);
    else rc = trip_gs(h, res);
    if (rc) return rc;
    if (use_copy) {
        HIPCHK(h, hipMemcpyAsync(slot, res, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else if ((rc = wait_pinned_word(h, slot))) return rc;
    *residual_norm = h->last_residual = *slot;
    return check_lines_failure(h);
}

extern "C" int its_iterate_n(its_handle *h, int iter_count0, int n_trips, double *residuals)
{
    debug_poison_lds(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (n_trips < 1 || !residuals) return ITS_EINVAL;
    if (h->solver == ITS_GMRES) {
        // GMRES needs its H column on the host every trip; time the whole region all the same
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        for (int i = 0; i < n_trips; ++i) {
            const int it = iter_count0 + i;
            if ((rc = trip_gmres(h, it, &residuals[i]))) return rc;
            h->last_residual = residuals[i];
            if ((it + 1) % h->m == 0 && i + 1 < n_trips) { if ((rc = its_gmres_restart(h, it))) return rc; }
        }
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipEventSynchronize(h->ev1));
        HIPCHK(h, hipEventElapsedTime(&h->last_region_ms, h->ev0, h->ev1));
        return ITS_OK;
    }
    if (n_trips > h->res_cap) {
        dfree(h->res_dev);
        if (dalloc(h, &h->res_dev, n_trips)) return ITS_EHIP;
        h->res_cap = n_trips;
    }
    if (h->solver == ITS_CG) {
        if (h->prof_ev.empty()) {
            h->prof_ev.resize(512);
            for (auto &e : h->prof_ev) HIPCHK(h, hipEventCreate(&e));
        }
        h->prof_n = 0; h->prof_on = true;
    }
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < n_trips; ++i) {
        if (h->solver == ITS_CG) rc = h->cg_variant == 1 ? trip_cg_pipelined(h, h->res_dev + i) : trip_cg(h, h->res_dev + i);
        else if (h->solver == ITS_JACOBI) rc = trip_jacobi(h, h->res_dev + i);
        else rc = trip_gs(h, h->res_dev + i);
        if (rc) return rc;
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipMemcpyAsync(residuals, h->res_dev, (size_t)n_trips * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipEventElapsedTime(&h->last_region_ms, h->ev0, h->ev1));
    h->prof_on = false;
    if (h->prof_n > 0) {
        double tot = 0.0;
        for (int i = 0; i < h->prof_n; ++i) {
            float ms = 0.f;
            HIPCHK(h, hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
            tot += ms;
        }
        h->prof_spmv_ms = (float)(tot / h->prof_n);
    }
    h->last_residual = residuals[n_trips - 1];
    return check_lines_failure(h);
}

extern "C" int its_gmres_restart(its_handle *h, int iter_count)
{
    if (h) h->gm_ahead_iter = -1;
    debug_poison_lds(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->solver != ITS_GMRES) FAIL(h, ITS_ESTATE, "not a GMRES handle");
    const int n = h->n;
    // x = x_old + V^T y (solver.hpp:563-590); r = b - A x (:609); precondition (:610)
    if ((rc = gmres_get_x(h, h->x, h->x_a, iter_count))) return rc;
    SpmvArgs a; a.x = h->x; a.y = h->r; a.b = h->b; a.part_sum = h->part[0];
    if ((rc = solver_spmv(h, a, EPI_RESIDUAL))) return rc;
    double *nrm_part = h->part[0];
    if (h->precond == ITS_PRECOND_GS) {
        if ((rc = sweep(h, h->r, h->r))) return rc;
        launch_dot_partials(h->r, h->r, n, h->part[1], h->stream);
        nrm_part = h->part[1];
    }
    // beta, V[0] = r / beta, fresh rotations (solver.hpp:621-641 -> gmres.hpp:726-786)
    if ((rc = finish_scalar(h, nrm_part, h->sc + SC_TMP, FIN_SUM))) return rc;
    launch_mgs_normalize(h->V, h->r, h->sc + SC_TMP, 1, h->hcol_dev, n, h->stream);
    HIPCHK(h, hipMemcpyAsync(h->pinned, h->hcol_dev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    // x_old <- x (solver.hpp:652-655)
    HIPCHK(h, hipMemcpyAsync(h->x_a, h->x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->beta = h->pinned[0];
    std::fill(h->g.begin(), h->g.end(), 0.0); h->g[0] = h->beta;
    std::fill(h->Rm.begin(), h->Rm.end(), 0.0);
    if (h->gmres_givens == 1) {                                        // init_gmres_structs (gmres.hpp:773-779)
        std::fill(h->Hd.begin(), h->Hd.end(), 0.0);
        std::fill(h->Rd.begin(), h->Rd.end(), 0.0);
        h->dense_full = false;
        std::fill(h->Qd.begin(), h->Qd.end(), 0.0);
        for (int i = 0; i <= h->m; ++i) h->Qd[(size_t)i * (h->m + 1) + i] = 1.0;
    }
    ++h->restart_count;
    return ITS_OK;
}

extern "C" int its_download_x(its_handle *h, double *x_star, int n, int iter_count)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!x_star || n != h->n) return ITS_EINVAL;
    const double *src = nullptr;
    if (h->solver == ITS_JACOBI || h->solver == ITS_CG) src = h->x_a;          // solver.hpp:663-698 (x_old)
    else if (h->solver == ITS_GAUSS_SEIDEL) src = h->x;                          // :699-708
    else {                                                                        // :709-725
        if ((rc = gmres_get_x(h, h->x_b, h->x_a, iter_count - 1))) return rc;
        src = h->x_b;
    }
    HIPCHK(h, hipMemcpyAsync(x_star, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ITS_OK;
}

extern "C" int its_final_residual(its_handle *h, double *residual_norm)
{
    debug_poison_lds(h);
    int rc = check_ready(h);
    if (rc) return rc;
    if (!residual_norm) return ITS_EINVAL;
    if (h->solver == ITS_GMRES) { *residual_norm = h->last_residual; return ITS_OK; }
    double *res = h->sc + SC_RES;
    if (h->solver == ITS_CG && h->cg_variant == 1) {
        // the same quirk as below, kept: the norm of the residual BEFORE the last trip = sqrt(gamma) that trip read
        launch_sqrt_scalar(h->sc + ((h->pcg_trip & 1) ? SC_PCG1 : SC_PCG0), res, h->stream);
    } else if (h->solver == ITS_CG) {
        // Solver::r after iterate's swap and exchange_arrays' swap (solver.hpp:545,833) is the
        // buffer that held r_old during the last trip, i.e. the PREVIOUS trip's residual.  The
        // device keeps one r; the previous norm is sqrt(rr_old) which still sits in the other slot.
        launch_sqrt_scalar(h->sc + (h->rr_slot ? SC_RR0 : SC_RR1), res, h->stream);
    } else {
        const double *xs = h->solver == ITS_JACOBI ? h->x_a : h->x;
        SpmvArgs q; q.x = xs; q.y = h->r; q.b = h->b; q.part_sum = h->part[0];
        if ((rc = solver_spmv(h, q, EPI_RESIDUAL))) return rc;
        if ((rc = finish_scalar(h, h->part[0], res, FIN_SQRT))) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->pinned, res, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *residual_norm = h->pinned[0];
    return ITS_OK;
}

// ---------------------------------------------------------------------------------------
// kernels alone
// ---------------------------------------------------------------------------------------
extern "C" int its_spmv(its_handle *h, int which, const double *x, double *y, int where, int format)
{
    debug_poison_lds(h);
    if (!h || which < 0 || which > 2 || !x || !y) return ITS_EINVAL;
    if (!h->have[which]) FAIL(h, ITS_ESTATE, "matrix not uploaded");
    HIPCHK(h, hipSetDevice(h->device));
    const CsrDev &A = h->mat[which];
    if (where == 1) {
        SpmvArgs a; a.x = x; a.y = y;
        int rc = spmv_dispatch(h, A, a, EPI_STORE, format);
        if (rc) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return ITS_OK;
    }
    double *dx = nullptr, *dy = nullptr;
    if (dalloc(h, &dx, A.n_cols) || dalloc(h, &dy, A.n_rows)) { dfree(dx); dfree(dy); return ITS_EHIP; }
    int rc = ITS_OK;
    hipError_t e = hipMemcpyAsync(dx, x, (size_t)A.n_cols * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        SpmvArgs a; a.x = dx; a.y = dy;
        rc = spmv_dispatch(h, A, a, EPI_STORE, format);
        if (rc == ITS_OK) e = hipMemcpyAsync(y, dy, (size_t)A.n_rows * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    }
    // the staging vectors are released on every path, and only once nothing queued on the stream still reads them
    const hipError_t es = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = es;
    if (rc == ITS_OK && e != hipSuccess) { h->err = hipGetErrorString(e); rc = ITS_EHIP; }
    dfree(dx); dfree(dy);
    return rc;
}

extern "C" int its_spmv_timed(its_handle *h, int which, const double *x_dev, double *y_dev, int format, int with_dot, int reps, float *avg_ms)
{
    if (!h || which < 0 || which > 2 || !x_dev || !y_dev || reps < 1 || !avg_ms) return ITS_EINVAL;
    if (!h->have[which]) FAIL(h, ITS_ESTATE, "matrix not uploaded");
    HIPCHK(h, hipSetDevice(h->device));
    SpmvArgs a; a.x = x_dev; a.y = y_dev;
    int epi = EPI_STORE;
    if (with_dot == 1) { a.dotv = x_dev; a.part_sum = h->part[3]; }
    // measurement forms of the residual epilogue (r = b - A x with its norm): 2 = r stored, 3 = norm only
    if (with_dot == 2 || with_dot == 3) { epi = EPI_RESIDUAL; a.b = x_dev; a.part_sum = h->part[3]; if (with_dot == 3) a.y = nullptr; }
    // 4..7 = the Jacobi / GS epilogues as the trips launch them (b := x): EPI_JACOBI_REF, EPI_GS_RHS, and their
    // fused forms EPI_JACOBI_FUSED, EPI_GS_FUSED
    if (with_dot >= 4 && with_dot <= 7) {
        const int e[4] = {EPI_JACOBI_REF, EPI_GS_RHS, EPI_JACOBI_FUSED, EPI_GS_FUSED};
        epi = e[with_dot - 4]; a.b = x_dev; a.part_sum = h->part[3];
    }
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) { int rc = spmv_dispatch(h, h->mat[which], a, epi, format); if (rc) return rc; }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_region_ms = ms;
    *avg_ms = ms / reps;
    return ITS_OK;
}

extern "C" int its_spmv_kernel_info(its_handle *h, int which, int with_dot, char *name, int cap, int64_t *streamed_bytes)
{
    if (!h || which < 0 || which > 2 || !name || cap < 1) return ITS_EINVAL;
    if (!h->have[which]) FAIL(h, ITS_ESTATE, "matrix not uploaded");
    SpmvArgs a; a.x = h->D; a.y = h->D;                       // any 16-byte aligned pointers: only alignment is looked at
    if (with_dot) a.dotv = h->D;
    const char *nm = spmv_exact_kernel_name(h->mat[which], a, EPI_STORE, streamed_bytes);
    snprintf(name, (size_t)cap, "%s", nm);
    return ITS_OK;
}

extern "C" int its_last_region_ms(its_handle *h, float *ms)
{
    if (!h || !ms) return ITS_EINVAL;
    *ms = h->last_region_ms;
    return ITS_OK;
}

extern "C" int its_last_spmv_ms(its_handle *h, float *ms)
{
    if (!h || !ms) return ITS_EINVAL;
    *ms = h->prof_spmv_ms;
    return ITS_OK;
}

static int scalar_out(its_handle *h, double *out)
{
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->pinned, h->sc + SC_TMP, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *out = h->pinned[0];
    return ITS_OK;
}

extern "C" int its_dot(its_handle *h, const double *a, const double *b, int64_t n, double *out)
{
    debug_poison_lds(h);
    if (!h || !a || !b || !out || n < 0) return ITS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    launch_dot_partials(a, b, n, h->part[2], h->stream);
    launch_reduce_finalize(h->part[2], h->sc + SC_TMP, FIN_SUM, h->stream);
    return scalar_out(h, out);
}

// its_dot without any allocation and without a host round trip: the partial sums go to this GPU's static scratch
// array (through the address taken at its_create, never through the symbol's name), the result to caller memory on
// the device.  Same two kernels, same fixed order as its_dot: bit-identical value.  Stream-ordered, no
// synchronisation: the value is in out_dev once the handle's stream has reached this point (its_synchronize).
extern "C" int its_dot_noalloc(its_handle *h, const double *a_dev, const double *b_dev, int64_t n, double *out_dev)
{
    if (!h || !a_dev || !b_dev || !out_dev || n < 0) return ITS_EINVAL;
    if (n > ITS_NOALLOC_MAX_N) FAIL(h, ITS_EINVAL, "its_dot_noalloc: n exceeds ITS_NOALLOC_MAX_N (2^33 elements)");
    if (h->device < 0 || h->device >= ITS_MAX_DEVICES || !g_scratch_ptr[h->device])
        FAIL(h, ITS_ESTATE, "its_dot_noalloc: no scratch address for this handle's device");
    HIPCHK(h, hipSetDevice(h->device));
    double *part = g_scratch_ptr[h->device];
    // every launch rewrites all ITS_NPART slots (its own partials, zeros in the slots a smaller grid leaves
    // unowned: store_partial), so nothing of an earlier, larger call survives
    launch_dot_partials(a_dev, b_dev, n, part, h->stream);
    launch_reduce_finalize(part, out_dev, FIN_SUM, h->stream);
    HIPCHK(h, hipGetLastError());
    return ITS_OK;
}

extern "C" int its_nrm2(its_handle *h, const double *v, int64_t n, double *out)
{
    debug_poison_lds(h);
    if (!h || !v || !out || n < 0) return ITS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    launch_dot_partials(v, v, n, h->part[2], h->stream);
    launch_reduce_finalize(h->part[2], h->sc + SC_TMP, FIN_SQRT, h->stream);
    return scalar_out(h, out);
}

extern "C" int its_norm_inf(its_handle *h, const double *v, int64_t n, double *out)
{
    debug_poison_lds(h);
    if (!h || !v || !out || n < 0) return ITS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    launch_absmax_partials(v, n, h->part[2], h->stream);
    launch_reduce_finalize(h->part[2], h->sc + SC_TMP, FIN_MAX, h->stream);
    return scalar_out(h, out);
}

extern "C" int its_axpby(its_handle *h, double *out, const double *a, const double *b, int64_t n, double s)
{
    debug_poison_lds(h);
    if (!h || !out || !a || !b || n < 0) return ITS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    launch_axpby(out, a, b, n, s, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ITS_OK;
}

extern "C" int its_scale(its_handle *h, double *out, const double *v, double s, int64_t n)
{
    debug_poison_lds(h);
    if (!h || !out || !v || n < 0) return ITS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    launch_scale(out, v, s, n, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ITS_OK;
}

extern "C" int its_spltsv(its_handle *h, double *x_dev, const double *rhs_dev)
{
    debug_poison_lds(h);
    if (!h || !x_dev || !rhs_dev) return ITS_EINVAL;
    if (!h->have[ITS_MAT_L] || !h->D) FAIL(h, ITS_ESTATE, "needs crs_L and D");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->level_rows) {
        const int n = h->mat[ITS_MAT_L].n_rows;
        if (h->L_has_halo_cols && !its_dist_active(h)) FAIL(h, ITS_EINVAL, "crs_L must be strictly lower triangular (it has columns beyond its rows and no row partition is set up)");
        for (int i = 0; i < n; ++i)
            if (h->L_maxcol[i] >= i) FAIL(h, ITS_EINVAL, "crs_L must be strictly lower triangular");
        int rc = build_levels(h, n);
        if (rc) return rc;
        if ((rc = build_lines_plan(h))) return rc;
    }
    int rc = sweep(h, x_dev, rhs_dev);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_lines_failure(h);
}

extern "C" int its_dev_alloc(its_handle *h, void **p, size_t bytes)
{
    if (!h || !p) return ITS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMalloc(p, std::max<size_t>(bytes, 16)));
    return ITS_OK;
}
extern "C" int its_dev_free(its_handle *h, void *p) { if (!h) return ITS_EINVAL; HIPCHK(h, hipFree(p)); return ITS_OK; }
extern "C" int its_memcpy_h2d(its_handle *h, void *d, const void *s, size_t bytes)
{ if (!h) return ITS_EINVAL; HIPCHK(h, hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream)); return ITS_OK; }
extern "C" int its_memcpy_d2h(its_handle *h, void *d, const void *s, size_t bytes)
{ if (!h) return ITS_EINVAL; HIPCHK(h, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream)); return ITS_OK; }
extern "C" int its_synchronize(its_handle *h) { if (!h) return ITS_EINVAL; HIPCHK(h, hipStreamSynchronize(h->stream)); return ITS_OK; }
