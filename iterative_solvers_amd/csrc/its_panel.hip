// its_panel.hip — the exact SpMV for large IRREGULAR matrices whose x does not fit an XCD's L2 (the 10 M-row power-law
// matrix: x = 80 MB), column-panel blocked so that most gathers hit L2 instead of the Infinity Cache.
//
// Why.  With one gather per nonzero into an 80 MB vector every exact, vector and SELL kernel sits on the chip's rate
// for isolated 8-byte gathers from the Infinity Cache (56.5 G/s, tools/gatherlab.hip: 1.49 ms for 84 M entries); an
// L2-resident table serves the same loop at 200 G/s.
// How.  The bit-exactness contract (spmv_crs_cpu, kernels.hpp:380-409) fixes the ORDER of a row's multiply-adds, not
// where its entries live.  If every row's columns ascend, walking x in column PANELS keeps each row's chain order: the
// entries of a row that fall into panel p all precede those of panel p + 1.  So
//   * a workgroup owns a TILE of up to PAN_TR = 4096 consecutive rows (fewer where rows are long: tiles hold about the same
//     number of entries) whose running sums stay in LDS for the whole walk;
//   * the tile's entries are stored panel-major — sorted by (panel, row, position in the row) — as 12 bytes each
//     (row-in-tile << 19 | column-in-panel, value), cut into chunks of <= PAN_CH entries that never span two panels;
//   * per chunk: all lanes load their entries (coalesced), gather x from the panel, park (value, x, row) in LDS; then
//     the lane holding the FIRST entry of a row's run walks the run in order, starting from the row's running sum;
//   * tiles are dealt round-robin, all resident workgroups start at panel 0 and walk at about the same pace, so the
//     panel they gather from (2^16 doubles = 512 KB by default) stays in each XCD's 4 MB L2.
// Rows longer than PAN_LCUT entries are left out of the image: their runs inside one panel would be walked by one
// lane while the workgroup waits (measured: slower than no blocking at all).  They get a wave each in a second launch
// (spmv_long_kernel), whose partial sums are added to this kernel's.
// Measured in the lab (tools/panellab.hip, profiles/r02_panellab.txt): 100 G entries/s for the rows of <= 128 entries.
//
// Epilogues: EPI_STORE (with or without the fused dot), EPI_RESIDUAL, EPI_JACOBI_REF, EPI_GS_RHS (b - U x, over the image
// of crs_U) — one running sum per row — and
// EPI_JACOBI_FUSED / EPI_GS_FUSED, which carry two chains per row: two running sums, 64 KB of LDS, half the rows in
// flight per CU.  The sweeps stay on the irregular-row kernel (a sweep touches one colour of L at a time: no reuse).
// The image is built on the device at upload: 64-bit keys (tile, panel, row) per entry, one stable radix sort
// (rocPRIM), one gather pass; chunk descriptors from the per-(tile, panel) counts on the host.
#include "its_handle.h"
#include "its_epilogue.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace its {

static int pan_env(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static int g_pan_on = pan_env("ITS_SPMV_PANEL", 1);
static int g_pan_min_cols = pan_env("ITS_SPMV_PANEL_MIN_COLS", 1 << 21);   // x of >= 16 MB: far beyond one XCD's L2
static int g_pan_log2pw = pan_env("ITS_SPMV_PANEL_LOG2PW", 16);
static int g_pan_lcut = pan_env("ITS_SPMV_PANEL_LCUT", PAN_LCUT);          // rows longer than this get a wave each (read at upload)
static int g_pan_grid = pan_env("ITS_SPMV_PANEL_GRID", 1024);              // 4 workgroups per CU (37 KB of LDS each)

int panel_tune(const char *name, int value)
{
    if (!strcmp(name, "spmv_panel")) { if (value < 0 || value > 1) return -1; g_pan_on = value; return 0; }
    if (!strcmp(name, "spmv_panel_min_cols")) { if (value < 1) return -1; g_pan_min_cols = value; return 0; }
    if (!strcmp(name, "spmv_panel_log2pw")) { if (value < 4 || value > PAN_COLBITS) return -1; g_pan_log2pw = value; return 0; }
    if (!strcmp(name, "spmv_panel_lcut")) { if (value < 1 || value > 4096) return -1; g_pan_lcut = value; return 0; }
    if (!strcmp(name, "spmv_panel_grid")) { if (value < 8 || value > ITS_NPART || (value & 7)) return -1; g_pan_grid = value; return 0; }
    return -2;
}

// ---------------------------------------------------------------------------------------------------------------
// image build
// ---------------------------------------------------------------------------------------------------------------
// one lane per row: sort key (tile, panel, row in tile) of every entry of a short row, ~0 for the entries of long rows
// (they sort to the end and are dropped); entries per (tile, panel); flags[0] |= 1 when a row's columns do not ascend
__global__ __launch_bounds__(ITS_BLOCK) void panel_mark_kernel(
    int n_rows, const int *__restrict__ rp, const int *__restrict__ col, int log2pw, int P, int lcut,
    const int *__restrict__ tile_row, int n_tiles,
    unsigned long long *__restrict__ key, unsigned int *__restrict__ src, int *__restrict__ piece_cnt,
    unsigned char *__restrict__ is_long, int *__restrict__ flags, const double *__restrict__ val, double *__restrict__ pdiag)
{
    const int row = blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (row >= n_rows) return;
    double dgv = 0.0;                                   // what the chain's `diag = dg ? v : diag` ends with (its_epilogue.h)
    const int j0 = rp[row], j1 = rp[row + 1];
    const bool lng = j1 - j0 > lcut;
    is_long[row] = lng;
    int tile = 0;                                        // last tile whose first row is <= row
    for (int lo = 0, hi = n_tiles - 1; lo <= hi;) { const int mid = (lo + hi) >> 1; if (tile_row[mid] <= row) { tile = mid; lo = mid + 1; } else hi = mid - 1; }
    const int r = row - tile_row[tile];
    int prev = -1; bool bad = false;
    for (int j = j0; j < j1; ++j) {
        const int c = col[j];
        bad |= c < prev; prev = c;
        if (c == row) dgv = val[j];
        src[j] = (unsigned int)j;
        if (lng) { key[j] = ~0ull; continue; }
        const int p = c >> log2pw;
        key[j] = ((unsigned long long)tile << 27) | ((unsigned long long)p << 12) | (unsigned long long)r;
        atomicAdd(&piece_cnt[(size_t)tile * P + p], 1);
    }
    pdiag[row] = dgv;
    if (bad) atomicOr(flags, 1);
}

__global__ __launch_bounds__(ITS_BLOCK) void panel_gather_kernel(
    int64_t n_short, const unsigned long long *__restrict__ key, const unsigned int *__restrict__ src,
    const int *__restrict__ col, const double *__restrict__ val, int log2pw, uint32_t *__restrict__ pkey, double *__restrict__ pval)
{
    const int64_t i = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (i >= n_short) return;
    const unsigned int j = src[i];
    pkey[i] = ((uint32_t)(key[i] & 4095ull) << PAN_COLBITS) | ((uint32_t)col[j] & ((1u << log2pw) - 1u));
    pval[i] = val[j];
}

}  // namespace its

using namespace its;

// Builds the panel image of A when it is the kind of matrix the blocking is for; leaves A.pan_* empty otherwise.
int build_panel_image(its_handle *h, CsrDev &A, const int *host_rp)
{
    const int n = A.n_rows;
    const int64_t nnz = A.nnz;
    if (!g_pan_on || n < 1 || nnz < 1 || A.max_row_len <= IRR_ROW_LEN || A.n_cols < g_pan_min_cols) return ITS_OK;
    const int log2pw = g_pan_log2pw, lcut = g_pan_lcut;
    const int64_t P64 = ((int64_t)A.n_cols + ((int64_t)1 << log2pw) - 1) >> log2pw;
    if (P64 >= (1 << 15)) return ITS_OK;
    const int P = (int)P64;
    // long rows and what the image would hold, from the host's row_ptr
    std::vector<unsigned char> is_long((size_t)n);
    std::vector<int> long_rows;
    int64_t nnz_long = 0;
    for (int i = 0; i < n; ++i) {
        const int len = host_rp[i + 1] - host_rp[i];
        is_long[i] = len > lcut;
        if (is_long[i]) { long_rows.push_back(i); nnz_long += len; }
    }
    // longest first (ties: ascending row): waves take the list round-robin, so the long walks start at once and the
    // work per wave evens out
    std::stable_sort(long_rows.begin(), long_rows.end(), [&](int x, int y) { return host_rp[x + 1] - host_rp[x] > host_rp[y + 1] - host_rp[y]; });
    if (nnz_long * 2 > nnz) return ITS_OK;                       // mostly long rows: nothing to block
    const int64_t n_short = nnz - nnz_long;
    // Tiles: consecutive rows, at most PAN_TR of them (the running sums in LDS) and at most 1.25x the mean tile's entries,
    // so that a stretch of longer rows (the late colours of a multicolour ordering: 130 K entries per 4096 rows against a
    // mean of 29 K on the power-law matrix) becomes more, shorter tiles instead of a few workgroups' long tail
    std::vector<int> tile_row(1, 0);
    {
        const int64_t full_tiles = ((int64_t)n + PAN_TR - 1) / PAN_TR;
        const int64_t mean = (n_short + full_tiles - 1) / full_tiles;
        const int64_t cap = std::max<int64_t>(mean + mean / 4, 4 * lcut);
        int64_t cur = 0; int rows_in = 0;
        for (int i = 0; i < n; ++i) {
            const int len = is_long[i] ? 0 : host_rp[i + 1] - host_rp[i];
            if (rows_in == PAN_TR || (rows_in > 0 && cur + len > cap)) { tile_row.push_back(i); cur = 0; rows_in = 0; }
            cur += len; ++rows_in;
        }
        tile_row.push_back(n);
    }
    const int n_tiles = (int)tile_row.size() - 1;
    if (n_tiles >= (1 << 19) || (int64_t)n_tiles * P64 > ((int64_t)1 << 27)) return ITS_OK;
    unsigned long long *key_in = nullptr, *key_out = nullptr;
    unsigned int *src_in = nullptr, *src_out = nullptr;
    int *piece_cnt = nullptr, *flags = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&]() { dfree(key_in); dfree(key_out); dfree(src_in); dfree(src_out); dfree(piece_cnt); dfree(flags); if (tmp) { (void)hipFree(tmp); tmp = nullptr; } };
#define PAN_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return ITS_EHIP; } } while (0)
    const size_t n_pieces = (size_t)n_tiles * P;
    if (dalloc(h, &key_in, (size_t)nnz) || dalloc(h, &key_out, (size_t)nnz) || dalloc(h, &src_in, (size_t)nnz) ||
        dalloc(h, &src_out, (size_t)nnz) || dalloc(h, &piece_cnt, n_pieces) || dalloc(h, &flags, 4) || dalloc(h, &A.pan_long, (size_t)n) ||
        dalloc(h, &A.pan_diag, (size_t)n) || dalloc(h, &A.pan_tile_row, tile_row.size())) { cleanup(); return ITS_EHIP; }
    PAN_TRY(hipMemcpyAsync(A.pan_tile_row, tile_row.data(), tile_row.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    PAN_TRY(hipMemsetAsync(piece_cnt, 0, n_pieces * sizeof(int), h->stream));
    PAN_TRY(hipMemsetAsync(flags, 0, 4 * sizeof(int), h->stream));
    hipLaunchKernelGGL(panel_mark_kernel, dim3((n + ITS_BLOCK - 1) / ITS_BLOCK), dim3(ITS_BLOCK), 0, h->stream, n, A.row_ptr, A.col,
                       log2pw, P, lcut, A.pan_tile_row, n_tiles, key_in, src_in, piece_cnt, A.pan_long, flags, A.val, A.pan_diag);
    PAN_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    PAN_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key_in, key_out, src_in, src_out, (size_t)nnz, 0u, 46u, h->stream));
    PAN_TRY(hipMalloc(&tmp, tmp_bytes + 256));
    PAN_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, key_in, key_out, src_in, src_out, (size_t)nnz, 0u, 46u, h->stream));
    std::vector<int> cnt(n_pieces);
    int hflags[4] = {0, 0, 0, 0};
    PAN_TRY(hipMemcpyAsync(cnt.data(), piece_cnt, n_pieces * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PAN_TRY(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, h->stream));
    PAN_TRY(hipStreamSynchronize(h->stream));
    if (hflags[0] & 1) {                                         // a row whose columns do not ascend: panel order != CRS order
        cleanup(); dfree(A.pan_long); dfree(A.pan_diag); dfree(A.pan_tile_row);
        return ITS_OK;
    }
    // chunk descriptors: {first entry, entries | panel << 9}, pieces in (tile, panel) order = the sorted order
    std::vector<int2> desc;
    std::vector<int> tile_ptr((size_t)n_tiles + 1, 0);
    desc.reserve(n_pieces + (size_t)(n_short / PAN_CH) + 16);
    int64_t off = 0;
    for (int t = 0; t < n_tiles; ++t) {
        tile_ptr[t] = (int)desc.size();
        for (int p = 0; p < P; ++p) {
            const int c = cnt[(size_t)t * P + p];
            for (int q = 0; q < c; q += PAN_CH) desc.push_back(make_int2((int)(off + q), std::min(PAN_CH, c - q) | (p << 9)));
            off += c;
        }
    }
    tile_ptr[n_tiles] = (int)desc.size();
    if (off != n_short) { cleanup(); dfree(A.pan_long); dfree(A.pan_diag); dfree(A.pan_tile_row); FAIL(h, ITS_EHIP, "panel image: entry counts do not add up"); }
    if (dalloc(h, &A.pan_key, (size_t)n_short) || dalloc(h, &A.pan_val, (size_t)n_short) || dalloc(h, &A.pan_desc, desc.size() + 2) ||
        dalloc(h, &A.pan_tile_ptr, tile_ptr.size()) || dalloc(h, &A.pan_long_rows, long_rows.size())) { cleanup(); return ITS_EHIP; }
    hipLaunchKernelGGL(panel_gather_kernel, dim3((unsigned)((n_short + ITS_BLOCK - 1) / ITS_BLOCK)), dim3(ITS_BLOCK), 0, h->stream,
                       n_short, key_out, src_out, A.col, A.val, log2pw, A.pan_key, A.pan_val);
    PAN_TRY(hipGetLastError());
    // two descriptors of padding behind the list: the kernel's look-ahead reads them (entries = 0)
    desc.push_back(make_int2(0, 0)); desc.push_back(make_int2(0, 0));
    PAN_TRY(hipMemcpyAsync(A.pan_desc, desc.data(), desc.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    PAN_TRY(hipMemcpyAsync(A.pan_tile_ptr, tile_ptr.data(), tile_ptr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (!long_rows.empty())
        PAN_TRY(hipMemcpyAsync(A.pan_long_rows, long_rows.data(), long_rows.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    PAN_TRY(hipStreamSynchronize(h->stream));
#undef PAN_TRY
    cleanup();
    A.pan_tiles = n_tiles; A.pan_n_long = (int)long_rows.size(); A.pan_log2pw = log2pw; A.pan_nnz = n_short;
    A.pan_chunks = (int)desc.size() - 2;
    if (getenv("ITS_DEBUG_PANEL"))
        fprintf(stderr, "panel image: %d tiles, %d panels, %d chunks for %lld entries (%.1f per chunk of %d), %d long rows with %lld entries\n",
                n_tiles, P, A.pan_chunks, (long long)n_short, A.pan_chunks ? (double)n_short / A.pan_chunks : 0.0, PAN_CH,
                (int)long_rows.size(), (long long)nnz_long);
    return ITS_OK;
}

namespace its {

// ---------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------
template <int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_panel_kernel(
    int n_rows, int n_tiles, const int *__restrict__ tile_row, const int *__restrict__ tile_ptr, const int2 *__restrict__ desc,
    const uint32_t *__restrict__ key, const double *__restrict__ val, const unsigned char *__restrict__ is_long,
    const double *__restrict__ pdiag, int log2pw, SpmvArgs a)
{
    constexpr int U = PAN_CH / ITS_BLOCK;
    static_assert(U == 1, "one entry per lane per chunk");
    // the fused forms carry two chains per row (its_epilogue.h): two running sums, half as many workgroups per CU
    constexpr int NACC = EpiKind<EPI>::fused ? 2 : 1;
    constexpr bool COLS = EpiKind<EPI>::needs_col;
    __shared__ double s_acc[NACC * PAN_TR];
    __shared__ double s_v[PAN_CH], s_x[PAN_CH];
    __shared__ int s_col[COLS ? PAN_CH : 1];
    __shared__ unsigned short s_row[PAN_CH];
    __shared__ double s_red[4];
    const int t = threadIdx.x;
    const int G = gridDim.x;
    double lane_sum = 0.0, lane_max = 0.0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += G) {
        const int d0 = tile_ptr[tile], d1 = tile_ptr[tile + 1];
        const int trow0 = tile_row[tile], trows = tile_row[tile + 1] - trow0;
        for (int i = t; i < PAN_TR; i += ITS_BLOCK) s_acc[i] = 0.0;
        // the full chain of the fused Gauss-Seidel pass over D + U starts from the sweep's chain over L (SpmvArgs::acc0)
        if (NACC == 2) for (int i = t; i < PAN_TR; i += ITS_BLOCK) s_acc[PAN_TR + i] = i < trows ? chain2_start<EPI>(a, trow0 + i) : 0.0;
        // software pipeline over the tile's chunks: entries two chunks ahead, gathers one chunk ahead of the chains
        uint32_t k0 = 0, k1 = 0; double v0 = 0.0, v1 = 0.0, x0 = 0.0;
        int m0 = 0, m1 = 0, p0 = 0, p1 = 0;
        if (d0 < d1) {
            const int2 ds = desc[d0];
            m0 = ds.y & 511; p0 = ds.y >> 9;
            if (t < m0) { k0 = __builtin_nontemporal_load(key + ds.x + t); v0 = __builtin_nontemporal_load(val + ds.x + t); }
            if (t < m0) x0 = a.x[((int64_t)p0 << log2pw) + (k0 & PAN_COLMASK)];
            if (d0 + 1 < d1) {
                const int2 dn = desc[d0 + 1];
                m1 = dn.y & 511; p1 = dn.y >> 9;
                if (t < m1) { k1 = __builtin_nontemporal_load(key + dn.x + t); v1 = __builtin_nontemporal_load(val + dn.x + t); }
            }
        }
        __syncthreads();                                      // running sums cleared (and the previous tile's epilogue done)
        for (int d = d0; d < d1; ++d) {
            if (t < m0) {
                s_v[t] = v0; s_x[t] = x0; s_row[t] = (unsigned short)(k0 >> PAN_COLBITS);
                if (COLS) s_col[t] = (int)(((int64_t)p0 << log2pw) + (k0 & PAN_COLMASK));
            }
            const int m = m0;
            __syncthreads();
            // gathers of chunk d + 1 and loads of chunk d + 2 are in flight during the chains
            if (t < m1) x0 = a.x[((int64_t)p1 << log2pw) + (k1 & PAN_COLMASK)];
            k0 = k1; v0 = v1; m0 = m1; p0 = p1;
            m1 = 0;
            if (d + 2 < d1) {
                const int2 dn = desc[d + 2];
                m1 = dn.y & 511; p1 = dn.y >> 9;
                if (t < m1) { k1 = __builtin_nontemporal_load(key + dn.x + t); v1 = __builtin_nontemporal_load(val + dn.x + t); }
            }
            // chains: the lane that holds the first entry of a row's run walks it, in order, from the row's running sums
            if (t < m) {
                const int r = s_row[t];
                if (t == 0 || s_row[t - 1] != r) {
                    const int row = trow0 + r;
                    // (parking the entries an epilogue leaves out as (0, 0) while staging, as spmv_long_kernel does, instead of
                    // the selects of chain_step in the walk: no gain here — Jacobi trip 1.263 vs 1.277 ms — the walk waits for LDS)
                    double acc = s_acc[r], acc2 = NACC == 2 ? s_acc[PAN_TR + r] : 0.0, dg = 0.0;
                    int j = t;
                    do { chain_step<EPI>(COLS ? s_col[j] : 0, row, s_v[j], s_x[j], acc, acc2, dg); ++j; } while (j < m && s_row[j] == r);
                    s_acc[r] = acc;
                    if (NACC == 2) s_acc[PAN_TR + r] = acc2;
                }
            }
            __syncthreads();
        }
        // epilogue of the tile's rows (long rows belong to the launch that follows)
        for (int i = t; i < trows; i += ITS_BLOCK) {
            const int row = trow0 + i;
            if (!is_long[row]) {
                double e0 = 0.0, e1 = 0.0, e2 = 0.0;
                epi_operands<EPI, DOT>(a, row, e0, e1, e2);
                // (the diagonal the Jacobi forms divide by: the row's last entry on the diagonal, noted at the image build)
                const double dg = EpiKind<EPI>::skip_diag ? pdiag[row] : 0.0;
                const double outv = row_finish<EPI, DOT>(s_acc[i], NACC == 2 ? s_acc[PAN_TR + i] : 0.0, dg, e0, e1, e2, lane_sum, lane_max);
                if (a.y) a.y[row] = outv;
            }
        }
        __syncthreads();
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

// ---------------------------------------------------------------------------------------------------------------
// The rows left out of the image (> PAN_LCUT entries): one WAVE per row, rows dealt to the waves from a list sorted by
// descending length.  The chain is serial by definition, so everything else is kept off it (as in sweep_long_kernel):
// col/val of chunk c + 2 and the x gathers of chunk c + 1 are in flight while chunk c is walked out of a wave-private LDS
// strip, whose reads run one batch of 8 ahead of the multiply-adds.  Partial sums are ADDED to the panel kernel's.
// (One-row blocks on the irregular-row kernel, the first version: 574 us for the 12.7 M entries of the 36 K long rows of
// the power-law matrix — one lane of 256 chains while a workgroup holds a row — and 883 us in multicolour order, where the
// hub rows sit next to each other in the list.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int LONG_LPL = 8;                         // entries per lane per chunk
constexpr int LONG_CHUNK = LONG_LPL * ITS_WAVE;     // 512 entries staged per trip
#ifndef ITS_LONG_WALK
#define ITS_LONG_WALK 16
#endif
constexpr int LONG_WALK = ITS_LONG_WALK;            // entries per batch of the walk
#ifndef ITS_LONG_PIPE
#define ITS_LONG_PIPE 1
#endif
#ifndef ITS_LONG_LANEWALK
#define ITS_LONG_LANEWALK 16                        // entries per lane of the register walk (0: every epilogue walks out of LDS)
#endif
#ifndef ITS_LONG_LANEWALK_MIN
#define ITS_LONG_LANEWALK_MIN 2048
#endif
constexpr int LW_MIN_LEN = ITS_LONG_LANEWALK_MIN;   // rows at least this long take the register walk: its loads are one lane, one
                                                    // cache line, and a short row leaves most lanes of its 1024-entry chunk idle
                                                    // (all rows on it: SpMV of the 10 M-row matrix 0.944 -> 0.990 ms)
#ifndef ITS_LONG_PIPE_W
#define ITS_LONG_PIPE_W 12
#endif
constexpr int LONG_PIPE_W = ITS_LONG_PIPE_W;        // entries per register set of the pipelined walk: two sets in flight must stay
                                                    // countable by s_waitcnt's 4-bit LDS counter (16 per set: the chain waits for
                                                    // reads just issued).  (Walking with only lanes 0 and 1 active: no faster — the LDS read costs the wave its 8 issue
                                                    // cycles whatever the mask, and that, not latency, is what stands beside the chain's 8.)
constexpr int LONG_PAD = 16;                        // the walk reads up to a batch past the chunk (never uses what it finds there)
static_assert(LONG_PIPE_W <= LONG_PAD && LONG_WALK <= LONG_PAD, "strip pad");

// The walk is one serial chain executed by a whole wave, so every instruction on it costs the SIMD a full issue: what an
// epilogue leaves out of its chain (the diagonal, or everything left of it) is therefore decided while STAGING, in
// parallel — an excluded entry is parked as (0, 0), which the chain passes as fma(0, 0, acc) = acc, the same operation
// chain_step performs — and the two chains of the fused forms run side by side in two lanes: lane 0 walks the masked
// copy of the strip, lane 1 the full one, one multiply-add per entry for the wave instead of two plus four selects
// (the fused passes over the long rows of the power-law matrix: Jacobi trip 1.36 -> 1.26 ms, multicolour GS 2.56 -> 2.49 ms).
template <int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_long_kernel(
    const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
    const int *__restrict__ rows, int n_long, SpmvArgs a)
{
    constexpr bool COLS = EpiKind<EPI>::needs_col;
    constexpr bool FUSED = EpiKind<EPI>::fused;
    __shared__ double2 s_vx[ITS_BLOCK / ITS_WAVE][LONG_CHUNK + LONG_PAD];                        // (value, x): the full chain
    __shared__ double2 s_m[FUSED ? ITS_BLOCK / ITS_WAVE : 1][FUSED ? LONG_CHUNK + LONG_PAD : 1]; // the same with excluded entries as (0, 0)
    __shared__ double s_red[4];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t wstride = (int64_t)gridDim.x * (ITS_BLOCK / ITS_WAVE);
    double lane_sum = 0.0, lane_max = 0.0;
    // rank k of the list (longest first) goes to wave k % 4 of workgroup k / 4: the four waves of a workgroup sit on the
    // four SIMDs of a CU, so the longest rows each get a SIMD of their own to start with and share it only with rows a
    // thousand ranks further down (dealt the other way — wave 0 of every workgroup takes the top ranks — the longest walks
    // all land on SIMD 0 of their CUs and time-slice each other: 275 us instead of the walk of one 8192-entry row)
#if ITS_LONG_LANEWALK
    // Epilogues that leave no entry out: the chain walks through the LANES' OWN REGISTERS.  A wave gets one LDS read through
    // every 12 cycles whatever its width (tools/ldsrate.hip), so a chain fed from LDS cannot go below 5 ns per entry; here
    // lane j holds LW_M CONSECUTIVE entries of the chunk, every lane runs the LW_M multiply-adds on its own entries from
    // the running sum, and the sum of lane j — the one lane whose input was the true sum — is handed to all lanes for step
    // j + 1 (two v_readlane): LW_M dependent multiply-adds and one hand-over per step, no LDS.  Same entries, same order,
    // same operation per entry: the same bits.  Entries past the row's end are staged as (0, 0): fma(0, 0, s) = s.
    if constexpr (!COLS && !FUSED) {
        constexpr int LW_M = ITS_LONG_LANEWALK, LW_CH = LW_M * ITS_WAVE;
        for (int64_t k = (int64_t)blockIdx.x * (ITS_BLOCK / ITS_WAVE) + w; k < n_long; k += wstride) {
            const int row = rows[k];
            const int rj0 = rp[row], rj1 = rp[row + 1];
            if (rj1 - rj0 < LW_MIN_LEN) continue;        // (the loop below takes it)
            int cn[LW_M]; double vn[LW_M], v1[LW_M], x1[LW_M], vv[LW_M], xg[LW_M];
#define LW_LOAD_CV(c0v)                                                                   \
            _Pragma("unroll") for (int u = 0; u < LW_M; ++u) {                            \
                const int e_ = (c0v) + l * LW_M + u;                                      \
                cn[u] = e_ < rj1 ? col[e_] : 0;                                           \
                vn[u] = e_ < rj1 ? val[e_] : 0.0;                                         \
            }
#define LW_GATHER(c0v)                                                                    \
            _Pragma("unroll") for (int u = 0; u < LW_M; ++u) {                            \
                const int e_ = (c0v) + l * LW_M + u;                                      \
                v1[u] = vn[u]; x1[u] = e_ < rj1 ? a.x[cn[u]] : 0.0;                       \
            }
            LW_LOAD_CV(rj0)
            LW_GATHER(rj0)
            LW_LOAD_CV(rj0 + LW_CH)
            double acc = 0.0;                            // the running sum: the same value in every lane
            for (int c0 = rj0; c0 < rj1; c0 += LW_CH) {
#pragma unroll
                for (int u = 0; u < LW_M; ++u) { vv[u] = v1[u]; xg[u] = x1[u]; }
                LW_GATHER(c0 + LW_CH)
                LW_LOAD_CV(c0 + 2 * LW_CH)
                const int nsteps = (min(LW_CH, rj1 - c0) + LW_M - 1) / LW_M;
                for (int j = 0; j < nsteps; ++j) {
                    double s_ = acc;
#pragma unroll
                    for (int u = 0; u < LW_M; ++u) s_ = fma(vv[u], xg[u], s_);
                    acc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s_), j),
                                           __builtin_amdgcn_readlane(__double2loint(s_), j));
                }
            }
#undef LW_LOAD_CV
#undef LW_GATHER
            if (l == 0) {
                double e0 = 0.0, e1 = 0.0, e2 = 0.0;
                epi_operands<EPI, DOT>(a, row, e0, e1, e2);
                const double outv = row_finish<EPI, DOT>(acc, 0.0, 0.0, e0, e1, e2, lane_sum, lane_max);
                if (a.y) a.y[row] = outv;
            }
        }
    }
#endif
    for (int64_t k = (int64_t)blockIdx.x * (ITS_BLOCK / ITS_WAVE) + w; k < n_long; k += wstride) {
        const int row = rows[k];
        const int rj0 = rp[row], rj1 = rp[row + 1];
#if ITS_LONG_LANEWALK
        if (!COLS && !FUSED && rj1 - rj0 >= LW_MIN_LEN) continue;      // (walked above)
#endif
        double acc1 = (FUSED && l == 1) ? chain2_start<EPI>(a, row) : 0.0;   // this lane's chain (lane 0: the epilogue's own, lane 1: the full one of the fused forms)
        double dg_l = 0.0; int dg_at = -1;               // Jacobi forms: the diagonal entry this lane staged last, and where
        int cn[LONG_LPL], cc[LONG_LPL]; double vn[LONG_LPL], vv[LONG_LPL], xg[LONG_LPL];
        const double2 *strip = (FUSED && l == 0) ? s_m[FUSED ? w : 0] : s_vx[w];
#define LONG_LOAD_CV(c0v)                                                                 \
        _Pragma("unroll") for (int u = 0; u < LONG_LPL; ++u) {                            \
            const int e_ = (c0v) + l + ITS_WAVE * u;                                      \
            cn[u] = e_ < rj1 ? col[e_] : 0;                                               \
            vn[u] = e_ < rj1 ? val[e_] : 0.0;                                             \
        }
#define LONG_GATHER()                                                                     \
        _Pragma("unroll") for (int u = 0; u < LONG_LPL; ++u) { vv[u] = vn[u]; cc[u] = cn[u]; xg[u] = a.x[cn[u]]; }
        LONG_LOAD_CV(rj0)
        LONG_GATHER()
        LONG_LOAD_CV(rj0 + LONG_CHUNK)
        for (int c0 = rj0; c0 < rj1; c0 += LONG_CHUNK) {
#pragma unroll
            for (int u = 0; u < LONG_LPL; ++u) {
                const int i = l + ITS_WAVE * u;
                bool out = false;                        // left out of the epilogue's chain (chain_step, its_epilogue.h)
                if (EpiKind<EPI>::skip_diag) { out = cc[u] == row; if (out && c0 + i < rj1) { dg_l = vv[u]; dg_at = c0 + i; } }
                if (EpiKind<EPI>::upper_only) out = !(cc[u] > row);
                const double2 full = make_double2(vv[u], xg[u]), masked = out ? make_double2(0.0, 0.0) : full;
                if (FUSED) { s_vx[w][i] = full; s_m[FUSED ? w : 0][i] = masked; }
                else s_vx[w][i] = COLS ? masked : full;
            }
            LONG_GATHER()
            LONG_LOAD_CV(c0 + 2 * LONG_CHUNK)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int ne = min(LONG_CHUNK, rj1 - c0);
            {
                // The walk: batches of LONG_WALK entries — all their LDS reads go out, the multiply-adds follow them in as they
                // arrive (one read latency per batch, then the chain's own 8 cycles per entry).  (Two alternating sets of
                // 8 with the next set read during the current one's multiply-adds compiled into 16 register copies per
                // trip and waits on reads just issued: 30 cycles per entry.)
#define LONG_RD(S, N, qv) _Pragma("unroll") for (int u = 0; u < (N); ++u) S[u] = strip[(qv) + u];
#define LONG_FMA(S, N, cnt) _Pragma("unroll") for (int u = 0; u < (N); ++u) if (u < (cnt)) acc1 = fma(S[u].x, S[u].y, acc1);
                int q = 0;
#if ITS_LONG_PIPE
                if (ne >= 2 * LONG_PIPE_W) {
                    // two register sets, each reloaded in place right after its multiply-adds: the reads of one batch are in
                    // flight during the other's chain (the last reload may run past the row's end, inside the strip's pad)
                    double2 pa[LONG_PIPE_W], pb[LONG_PIPE_W];
                    LONG_RD(pa, LONG_PIPE_W, 0)
                    // (the empty asm ties the order down: without it the scheduler hoists the reload above the chain that
                    // still reads the set, and pays with 32 register copies per trip, or moves a chain above the reads meant to run beside it)
                    for (; q + 2 * LONG_PIPE_W <= ne; q += 2 * LONG_PIPE_W) {
                        LONG_RD(pb, LONG_PIPE_W, q + LONG_PIPE_W)
                        asm volatile("" : "+v"(acc1) :: "memory");
                        LONG_FMA(pa, LONG_PIPE_W, LONG_PIPE_W)
                        asm volatile("" : "+v"(acc1) :: "memory");
                        LONG_RD(pa, LONG_PIPE_W, q + 2 * LONG_PIPE_W)
                        asm volatile("" : "+v"(acc1) :: "memory");
                        LONG_FMA(pb, LONG_PIPE_W, LONG_PIPE_W)
                        asm volatile("" : "+v"(acc1) :: "memory");
                    }
                    if (q + LONG_PIPE_W <= ne) { LONG_FMA(pa, LONG_PIPE_W, LONG_PIPE_W) q += LONG_PIPE_W; }
                }
#endif
                for (; q + LONG_WALK <= ne; q += LONG_WALK) {
                    double2 pw[LONG_WALK];
                    LONG_RD(pw, LONG_WALK, q)
                    LONG_FMA(pw, LONG_WALK, LONG_WALK)
                }
                for (; q < ne; q += 8) {
                    double2 p8[8];
                    LONG_RD(p8, 8, q)                  // (the strip is longer than a chunk)
                    LONG_FMA(p8, 8, ne - q)
                }
#undef LONG_RD
#undef LONG_FMA
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
#undef LONG_LOAD_CV
#undef LONG_GATHER
        // the diagonal the chain ends with is the LAST one staged (chain_step: diag = dg ? v : diag)
        double dg = 0.0;
        if (EpiKind<EPI>::skip_diag) {
            int best = dg_at;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o, ITS_WAVE));
            const unsigned long long who = __ballot(best >= 0 && dg_at == best);
            if (who) dg = __shfl(dg_l, __ffsll((long long)who) - 1, ITS_WAVE);
        }
        const double acc = __shfl(acc1, 0, ITS_WAVE), acc2 = FUSED ? __shfl(acc1, 1, ITS_WAVE) : 0.0;
        if (l == 0) {
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
            epi_operands<EPI, DOT>(a, row, e0, e1, e2);
            const double outv = row_finish<EPI, DOT>(acc, acc2, dg, e0, e1, e2, lane_sum, lane_max);
            if (a.y) a.y[row] = outv;
        }
    }
    if (DOT || EpiKind<EPI>::norm) {
        const double bs = block_sum(lane_sum, s_red);
        if (threadIdx.x == 0 && a.part_sum) a.part_sum[blockIdx.x] += bs;
        if (EpiKind<EPI>::norm && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (threadIdx.x == 0) a.part_max[blockIdx.x] = fmax(a.part_max[blockIdx.x], bm);
        }
    }
}

bool spmv_panel_eligible(const CsrDev &A, const SpmvArgs &a, int epi)
{
    (void)a;
    return g_pan_on && A.pan_key != nullptr &&
           (epi == EPI_STORE || epi == EPI_RESIDUAL || epi == EPI_JACOBI_SHIPPED || epi == EPI_JACOBI_REF || epi == EPI_JACOBI_FUSED ||
            epi == EPI_GS_RHS || epi == EPI_GS_FUSED);
}

void launch_spmv_panel(const CsrDev &A, const SpmvArgs &a_in, int epi, hipStream_t s)
{
    SpmvArgs a = a_in;
    a.part_accum = 0;
    int grid = g_pan_grid;
    if (epi == EPI_JACOBI_FUSED || epi == EPI_GS_FUSED) grid = (grid / 16) * 8;         // two running sums per row: 64 KB of LDS, 2 workgroups per CU
    if (grid < 8) grid = 8;
    const int need = ((A.pan_tiles + 7) / 8) * 8;
    if (grid > need) grid = need;
    dim3 g(grid), b(ITS_BLOCK);
#define ITS_LAUNCHP(E, D) \
    hipLaunchKernelGGL((spmv_panel_kernel<E, D>), g, b, 0, s, A.n_rows, A.pan_tiles, A.pan_tile_row, A.pan_tile_ptr, A.pan_desc, A.pan_key, A.pan_val, \
                       A.pan_long, A.pan_diag, A.pan_log2pw, a)
    switch (epi) {
    case EPI_STORE: if (a.dotv) ITS_LAUNCHP(EPI_STORE, true); else ITS_LAUNCHP(EPI_STORE, false); break;
    case EPI_RESIDUAL: ITS_LAUNCHP(EPI_RESIDUAL, false); break;
    case EPI_JACOBI_SHIPPED: ITS_LAUNCHP(EPI_JACOBI_SHIPPED, false); break;
    case EPI_JACOBI_REF: ITS_LAUNCHP(EPI_JACOBI_REF, false); break;
    case EPI_GS_RHS: ITS_LAUNCHP(EPI_GS_RHS, false); break;
    case EPI_GS_FUSED: ITS_LAUNCHP(EPI_GS_FUSED, false); break;
    default: ITS_LAUNCHP(EPI_JACOBI_FUSED, false); break;
    }
#undef ITS_LAUNCHP
    // the rows left out of the image: one wave each, partial sums added to the ones above (its grid never exceeds the
    // panel kernel's zero-filled ITS_NPART slots)
    launch_spmv_long(A, A.pan_long_rows, A.pan_n_long, a, epi, s);
}

void launch_spmv_long(const CsrDev &A, const int *rows, int n_long, const SpmvArgs &a, int epi, hipStream_t s)
{
    if (n_long <= 0) return;
    int gl = (n_long + 3) / 4;
    if (gl > ITS_NPART) gl = ITS_NPART;
    dim3 g2(gl), b(ITS_BLOCK);
#define ITS_LAUNCHL(E, D) \
    hipLaunchKernelGGL((spmv_long_kernel<E, D>), g2, b, 0, s, A.row_ptr, A.col, A.val, rows, n_long, a)
    switch (epi) {
    case EPI_STORE: if (a.dotv) ITS_LAUNCHL(EPI_STORE, true); else ITS_LAUNCHL(EPI_STORE, false); break;
    case EPI_RESIDUAL: ITS_LAUNCHL(EPI_RESIDUAL, false); break;
    case EPI_JACOBI_SHIPPED: ITS_LAUNCHL(EPI_JACOBI_SHIPPED, false); break;
    case EPI_JACOBI_REF: ITS_LAUNCHL(EPI_JACOBI_REF, false); break;
    case EPI_GS_RHS: ITS_LAUNCHL(EPI_GS_RHS, false); break;
    case EPI_GS_FUSED: ITS_LAUNCHL(EPI_GS_FUSED, false); break;
    default: ITS_LAUNCHL(EPI_JACOBI_FUSED, false); break;
    }
#undef ITS_LAUNCHL
}

}  // namespace its
