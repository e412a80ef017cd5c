// panellab — can the exact (in-order chain) SpMV of the 10 M-row power-law matrix get under the Infinity-Cache gather
// ceiling (tools/gatherlab: 56.5 G gathers/s -> 1.49 ms) by making the x gathers hit L2?   Not part of the library.
//
// Idea: the bit-exactness contract fixes the ORDER of a row's multiply-adds, not where its entries live.  If every row's
// columns ascend (they do after convert_to_crs-style sorting), walking x in column PANELS keeps each row's chain order.
// A workgroup owns a TILE of TR rows whose running sums stay in LDS; the tile's entries are stored panel-major
// ((panel, row, col) order); all workgroups of an XCD walk the panels at about the same pace, so the panel of x they
// gather from stays in that XCD's 4 MB L2.  Reuse per XCD = rows in flight x mean row length / sectors of x.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fopenmp -o tools/panellab tools/panellab.hip && ./tools/panellab [n] [log2 panel]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static inline uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline double u01(uint64_t h) { return ((h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

constexpr int ROWSHIFT = 19;            // key = row_in_tile << 19 | col_in_panel  (panel <= 2^19 doubles, tile <= 8192 rows)
constexpr uint32_t COLMASK = (1u << ROWSHIFT) - 1;

struct Desc { uint32_t off_lo; uint32_t off_hi_m_panel; };   // unused: descriptors are plain int4 below

// one chunk of one (tile, panel) piece: entries [off, off + m), panel p
// desc: x = off low 32, y = off high, z = m, w = panel
template <int TR, int CH, int NT>
__global__ __launch_bounds__(NT) void panel_kernel(const int *__restrict__ tile_desc_ptr, const int4 *__restrict__ desc,
                                                   const uint32_t *__restrict__ key, const double *__restrict__ val,
                                                   const double *__restrict__ x, double *__restrict__ y, int n_rows, int log2pw)
{
    constexpr int U = CH / NT;
    __shared__ double s_acc[TR];
    __shared__ double s_v[CH], s_x[CH];
    __shared__ unsigned short s_row[CH];
    const int t = threadIdx.x;
    const int tile = blockIdx.x;
    for (int i = t; i < TR; i += NT) s_acc[i] = 0.0;
    const int d0 = tile_desc_ptr[tile], d1 = tile_desc_ptr[tile + 1];
    uint32_t k0[U], k1[U]; double v0[U], v1[U], x0[U];
    int m0 = 0, m1 = 0, p1 = 0;
    auto load = [&](int d, uint32_t (&k)[U], double (&v)[U], int &m, int &p) {
        if (d < d1) {
            const int4 ds = desc[d];
            const int64_t off = (int64_t)(uint32_t)ds.x | ((int64_t)ds.y << 32);
            m = ds.z; p = ds.w;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = t + u * NT;
                k[u] = 0; v[u] = 0.0;
                if (i < m) { k[u] = __builtin_nontemporal_load(key + off + i); v[u] = __builtin_nontemporal_load(val + off + i); }
            }
        } else { m = 0; p = 0; }
    };
    int p0 = 0;
    load(d0, k0, v0, m0, p0);
    load(d0 + 1, k1, v1, m1, p1);
    {
        const double *xb = x + ((int64_t)p0 << log2pw);
#pragma unroll
        for (int u = 0; u < U; ++u) x0[u] = (t + u * NT < m0) ? xb[k0[u] & COLMASK] : 0.0;
    }
    __syncthreads();
    for (int d = d0; d < d1; ++d) {
        // stage chunk d
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = t + u * NT;
            if (i < m0) { s_v[i] = v0[u]; s_x[i] = x0[u]; s_row[i] = (unsigned short)(k0[u] >> ROWSHIFT); }
        }
        const int m = m0;
        __syncthreads();
        // gathers of chunk d + 1, loads of chunk d + 2: in flight during the chains
        {
            const double *xb = x + ((int64_t)p1 << log2pw);
#pragma unroll
            for (int u = 0; u < U; ++u) { x0[u] = (t + u * NT < m1) ? xb[k1[u] & COLMASK] : 0.0; k0[u] = k1[u]; v0[u] = v1[u]; }
            m0 = m1; p0 = p1;
            load(d + 2, k1, v1, m1, p1);
        }
        // chains: the lane that holds the first entry of a row's run walks the run
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = t + u * NT;
            if (i < m) {
                const int r = s_row[i];
                if (i == 0 || s_row[i - 1] != r) {
                    // (a batched walker — 8 LDS reads in flight per step of a long run — measured slower: 1.93 vs 1.80 ms)
                    double acc = s_acc[r];
                    int j = i;
                    do { acc = fma(s_v[j], s_x[j], acc); ++j; } while (j < m && s_row[j] == r);
                    s_acc[r] = acc;
                }
            }
        }
        __syncthreads();
    }
    for (int i = t; i < TR; i += NT) {
        const int row = tile * TR + i;
        if (row < n_rows) y[row] = s_acc[i];
    }
}

// ---- design B: no staging, no barriers.  A wave owns RW rows (running sums in LDS); its entries come as GROUPS of 64 whose
// rows are pairwise distinct, in an order that keeps every row's entries in CRS order.  key = row_in_wave << 22 | column
// relative to the group's base column.
constexpr int BSHIFT = 21;            // 11 bits of row (RW + 1 slots with the dummy row of the padding), 21 of column
constexpr uint32_t BMASK = (1u << BSHIFT) - 1;
template <int RW, int G, int WPB>
__global__ __launch_bounds__(WPB * 64) void wave_panel_kernel(int n_wstripes, const int *__restrict__ wptr, const int *__restrict__ gbase,
                                                              const uint32_t *__restrict__ key, const double *__restrict__ val,
                                                              const double *__restrict__ x, double *__restrict__ y, int n_rows,
                                                              const unsigned char *__restrict__ is_long)
{
    __shared__ double s_acc[WPB * (RW + 1)];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ws = blockIdx.x * WPB + w;
    if (ws >= n_wstripes) return;
    double *acc = s_acc + w * (RW + 1);
    for (int i = lane; i <= RW; i += 64) acc[i] = 0.0;
    const int g0 = wptr[ws], g1 = wptr[ws + 1];          // multiple of G groups
    uint32_t k[G], kn[G]; double v[G], vn[G];
#pragma unroll
    for (int u = 0; u < G; ++u) { kn[u] = 0; vn[u] = 0.0; }
    if (g0 < g1) {
#pragma unroll
        for (int u = 0; u < G; ++u) { kn[u] = __builtin_nontemporal_load(key + (int64_t)(g0 + u) * 64 + lane); vn[u] = __builtin_nontemporal_load(val + (int64_t)(g0 + u) * 64 + lane); }
    }
    for (int g = g0; g < g1; g += G) {
        double xx[G];
#pragma unroll
        for (int u = 0; u < G; ++u) { k[u] = kn[u]; v[u] = vn[u]; }
#pragma unroll
        for (int u = 0; u < G; ++u) xx[u] = x[(int64_t)gbase[g + u] + (k[u] & BMASK)];
        if (g + G < g1) {
#pragma unroll
            for (int u = 0; u < G; ++u) { kn[u] = __builtin_nontemporal_load(key + (int64_t)(g + G + u) * 64 + lane); vn[u] = __builtin_nontemporal_load(val + (int64_t)(g + G + u) * 64 + lane); }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) { const int r = k[u] >> BSHIFT; acc[r] = fma(v[u], xx[u], acc[r]); }
    }
    for (int i = lane; i < RW; i += 64) {
        const int row = ws * RW + i;
        if (row < n_rows && !is_long[row]) y[row] = acc[i];
    }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000000;
    const int log2pw = argc > 2 ? atoi(argv[2]) : 17;
    const int TRsel = argc > 3 ? atoi(argv[3]) : 4096;
    const int max_len = getenv("MAXLEN") ? atoi(getenv("MAXLEN")) : 8192;
    const bool force_p0 = getenv("FORCEP0") != nullptr;     // every gather from panel 0: the kernel's structure without the locality question
    const int only = getenv("ONLY") ? atoi(getenv("ONLY")) : -1;
    // ---- matrix: power-law row lengths, uniform-random columns, sorted, de-duplicated, diagonal present
    std::vector<int64_t> rp((size_t)n + 1, 0);
    std::vector<int> len(n);
#pragma omp parallel for
    for (int i = 0; i < n; ++i) {
        const double u = u01(mix((uint64_t)i * 0xD1B54A32D192ED03ull + 12345));
        double l = std::floor(3.0 * std::pow(u, -2.0 / 3.0));
        len[i] = (int)std::min<double>(std::max(l, 1.0), std::min(max_len, n));
    }
    for (int i = 0; i < n; ++i) rp[i + 1] = rp[i] + len[i];
    std::vector<int> col(rp[n]); std::vector<double> val(rp[n]);
#pragma omp parallel
    {
        std::vector<int> c;
#pragma omp for schedule(dynamic, 4096)
        for (int i = 0; i < n; ++i) {
            c.clear(); c.push_back(i);
            for (int k = 0; k + 1 < len[i]; ++k) c.push_back((int)(mix((uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)k * 0xC2B2AE3D27D4EB4Full + 12345) % (uint64_t)n));
            std::sort(c.begin(), c.end());
            c.erase(std::unique(c.begin(), c.end()), c.end());
            len[i] = (int)c.size();
            for (size_t k = 0; k < c.size(); ++k) { col[rp[i] + k] = c[k]; val[rp[i] + k] = c[k] == i ? 10.0 : -u01(mix((uint64_t)c[k] * 77 + i)); }
        }
    }
    // (rows keep their allotted space; len[i] is the real length — compact)
    std::vector<int64_t> rp2((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) rp2[i + 1] = rp2[i] + len[i];
    const int64_t nnz = rp2[n];
    {
        std::vector<int> c2(nnz); std::vector<double> v2(nnz);
#pragma omp parallel for
        for (int i = 0; i < n; ++i) for (int k = 0; k < len[i]; ++k) { c2[rp2[i] + k] = col[rp[i] + k]; v2[rp2[i] + k] = val[rp[i] + k]; }
        col.swap(c2); val.swap(v2); rp.swap(rp2);
    }
    printf("n %d nnz %lld  panel 2^%d doubles (%d panels)  tile %d rows\n", n, (long long)nnz, log2pw, (n + (1 << log2pw) - 1) >> log2pw, TRsel);
    std::vector<double> x(n), yref(n);
#pragma omp parallel for
    for (int i = 0; i < n; ++i) x[i] = u01(mix(i + 999)) - 0.5;
#pragma omp parallel for
    for (int i = 0; i < n; ++i) { double a = 0.0; for (int64_t j = rp[i]; j < rp[i + 1]; ++j) a = fma(val[j], x[col[j]], a); yref[i] = a; }

    double *dx, *dy, *dval; uint32_t *dkey; int4 *ddesc; int *dtp;
    CK(hipMalloc(&dx, (size_t)n * 8)); CK(hipMalloc(&dy, (size_t)n * 8));
    CK(hipMemcpy(dx, x.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    int run_no = 0;
    auto run = [&](auto kern, int TR, int CH, int NT, const char *tag) {
        if (only >= 0 && run_no++ != only) return;
        const int P = (n + (1 << log2pw) - 1) >> log2pw;
        const int n_tiles = (n + TR - 1) / TR;
        // image: per tile, entries in (panel, row, col) order; chunk descriptors
        std::vector<uint32_t> key(nnz); std::vector<double> pv(nnz);
        std::vector<std::vector<int4>> tdesc(n_tiles);
#pragma omp parallel
        {
            std::vector<int64_t> cnt(P + 1);
#pragma omp for schedule(dynamic, 16)
            for (int tl = 0; tl < n_tiles; ++tl) {
                const int r0 = tl * TR, r1 = std::min(n, r0 + TR);
                std::fill(cnt.begin(), cnt.end(), 0);
                for (int64_t j = rp[r0]; j < rp[r1]; ++j) cnt[(col[j] >> log2pw) + 1]++;
                for (int p = 0; p < P; ++p) cnt[p + 1] += cnt[p];
                const int64_t base = rp[r0];
                std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
                for (int r = r0; r < r1; ++r)
                    for (int64_t j = rp[r]; j < rp[r + 1]; ++j) {
                        const int p = col[j] >> log2pw;
                        const int64_t o = base + pos[p]++;
                        key[o] = ((uint32_t)(r - r0) << ROWSHIFT) | (uint32_t)(col[j] & ((1 << log2pw) - 1));
                        pv[o] = val[j];
                    }
                for (int p = 0; p < P; ++p)
                    for (int64_t c = cnt[p]; c < cnt[p + 1]; c += CH) {
                        const int64_t off = base + c;
                        tdesc[tl].push_back(make_int4((int)(uint32_t)(off & 0xffffffff), (int)(off >> 32), (int)std::min<int64_t>(CH, cnt[p + 1] - c), force_p0 ? 0 : p));
                    }
            }
        }
        std::vector<int> tp(n_tiles + 1, 0);
        for (int tl = 0; tl < n_tiles; ++tl) tp[tl + 1] = tp[tl] + (int)tdesc[tl].size();
        std::vector<int4> desc(tp[n_tiles]);
        for (int tl = 0; tl < n_tiles; ++tl) std::copy(tdesc[tl].begin(), tdesc[tl].end(), desc.begin() + tp[tl]);
        double fill = 0; for (auto &d : desc) fill += d.z; fill /= (double)desc.size() * CH;
        CK(hipMalloc(&dkey, nnz * 4)); CK(hipMalloc(&dval, nnz * 8)); CK(hipMalloc(&ddesc, desc.size() * 16)); CK(hipMalloc(&dtp, tp.size() * 4));
        CK(hipMemcpy(dkey, key.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dval, pv.data(), nnz * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(ddesc, desc.data(), desc.size() * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dtp, tp.data(), tp.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dy, 0, (size_t)n * 8));
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(NT), 0, 0, dtp, ddesc, dkey, dval, dx, dy, n, log2pw);
        CK(hipGetLastError());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(NT), 0, 0, dtp, ddesc, dkey, dval, dx, dy, n, log2pw);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        std::vector<double> yh(n);
        CK(hipMemcpy(yh.data(), dy, (size_t)n * 8, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int i = 0; i < n; ++i) bad += memcmp(&yh[i], &yref[i], 8) != 0;
        printf("%-34s tiles %5d chunks %7zu fill %.2f  %.4f ms  %.1f G entries/s  mismatching rows %lld\n", tag, n_tiles, desc.size(), fill, ms, nnz / ms / 1e6, (long long)bad);
        fflush(stdout);
        CK(hipFree(dkey)); CK(hipFree(dval)); CK(hipFree(ddesc)); CK(hipFree(dtp));
    };

    const int LCUT = getenv("LCUT") ? atoi(getenv("LCUT")) : 64;
    auto runB = [&](auto kern, int RW, int G, int WPB, const char *tag) {
        if (only >= 0 && run_no++ != only) return;
        const int P = (n + (1 << log2pw) - 1) >> log2pw;
        const int nws = (n + RW - 1) / RW;
        std::vector<unsigned char> is_long(n);
        int64_t nnz_short = 0;
        for (int i = 0; i < n; ++i) { is_long[i] = rp[i + 1] - rp[i] > LCUT; if (!is_long[i]) nnz_short += rp[i + 1] - rp[i]; }
        std::vector<std::vector<uint32_t>> wkey(nws); std::vector<std::vector<double>> wval(nws); std::vector<std::vector<int>> wbase(nws);
#pragma omp parallel
        {
            std::vector<int64_t> cur(RW);                // next entry of each row
            std::vector<int> ingroup(RW);                // group index a row was last placed in
#pragma omp for schedule(dynamic, 16)
            for (int ws = 0; ws < nws; ++ws) {
                const int r0 = ws * RW, r1 = std::min(n, r0 + RW);
                for (int r = r0; r < r1; ++r) { cur[r - r0] = rp[r]; ingroup[r - r0] = -1; }
                auto &K = wkey[ws]; auto &V = wval[ws]; auto &B = wbase[ws];
                int fill = 64;                           // slots used in the open group (64 = none open)
                int gi = -1; int base = 0;
                auto open = [&](int b) { while (fill < 64) { K.push_back((uint32_t)RW << BSHIFT); V.push_back(0.0); ++fill; } ++gi; base = b; B.push_back(b); fill = 0; };
                for (int p = 0; p < P; ++p) {
                    const int pend = std::min<int64_t>(n, ((int64_t)p + 1) << log2pw);
                    bool more = true;
                    while (more) {                       // rounds: the next entry of every row that still has one in this panel
                        more = false;
                        for (int r = r0; r < r1; ++r) {
                            if (is_long[r]) continue;
                            int64_t &c = cur[r - r0];
                            if (c < rp[r + 1] && col[c] < pend) {
                                if (fill == 64 || ingroup[r - r0] == gi || (int64_t)col[c] - base > (int64_t)BMASK) open(p << log2pw);
                                K.push_back(((uint32_t)(r - r0) << BSHIFT) | (uint32_t)(col[c] - base)); V.push_back(val[c]); ++fill;
                                ingroup[r - r0] = gi;
                                ++c;
                                if (c < rp[r + 1] && col[c] < pend) more = true;
                            }
                        }
                    }
                }
                while (fill < 64) { K.push_back((uint32_t)RW << BSHIFT); V.push_back(0.0); ++fill; }
                while (B.size() % G) { B.push_back(0); for (int q = 0; q < 64; ++q) { K.push_back((uint32_t)RW << BSHIFT); V.push_back(0.0); } }
            }
        }
        std::vector<int> wptr(nws + 1, 0);
        for (int ws = 0; ws < nws; ++ws) wptr[ws + 1] = wptr[ws] + (int)wbase[ws].size();
        const int64_t ng = wptr[nws];
        std::vector<uint32_t> key(ng * 64); std::vector<double> pv(ng * 64); std::vector<int> gb(ng);
#pragma omp parallel for schedule(dynamic, 64)
        for (int ws = 0; ws < nws; ++ws) {
            std::copy(wkey[ws].begin(), wkey[ws].end(), key.begin() + (int64_t)wptr[ws] * 64);
            std::copy(wval[ws].begin(), wval[ws].end(), pv.begin() + (int64_t)wptr[ws] * 64);
            std::copy(wbase[ws].begin(), wbase[ws].end(), gb.begin() + wptr[ws]);
        }
        int *dgb; unsigned char *dlong;
        CK(hipMalloc(&dkey, ng * 64 * 4)); CK(hipMalloc(&dval, ng * 64 * 8)); CK(hipMalloc(&dgb, ng * 4)); CK(hipMalloc(&dtp, wptr.size() * 4)); CK(hipMalloc(&dlong, n));
        CK(hipMemcpy(dkey, key.data(), ng * 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dval, pv.data(), ng * 64 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dgb, gb.data(), ng * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dtp, wptr.data(), wptr.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dlong, is_long.data(), n, hipMemcpyHostToDevice));
        CK(hipMemset(dy, 0, (size_t)n * 8));
        const int grid = (nws + WPB - 1) / WPB;
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WPB * 64), 0, 0, nws, dtp, dgb, dkey, dval, dx, dy, n, dlong);
        CK(hipGetLastError());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WPB * 64), 0, 0, nws, dtp, dgb, dkey, dval, dx, dy, n, dlong);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        std::vector<double> yh(n);
        CK(hipMemcpy(yh.data(), dy, (size_t)n * 8, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int i = 0; i < n; ++i) bad += !is_long[i] && memcmp(&yh[i], &yref[i], 8) != 0;
        printf("B %-26s LCUT %d: %lld of %lld entries, groups %lld fill %.3f  %.4f ms  %.1f G entries/s  mismatching rows %lld\n", tag, LCUT,
               (long long)nnz_short, (long long)nnz, (long long)ng, nnz_short / (ng * 64.0), ms, nnz_short / ms / 1e6, (long long)bad);
        fflush(stdout);
        CK(hipFree(dkey)); CK(hipFree(dval)); CK(hipFree(dgb)); CK(hipFree(dtp)); CK(hipFree(dlong));
    };
    if (getenv("DESIGN_B")) {
        runB(wave_panel_kernel<1024, 4, 4>, 1024, 4, 4, "RW 1024 G 4 WPB 4");
        runB(wave_panel_kernel<1024, 8, 4>, 1024, 8, 4, "RW 1024 G 8 WPB 4");
        runB(wave_panel_kernel<1024, 2, 4>, 1024, 2, 4, "RW 1024 G 2 WPB 4");
        runB(wave_panel_kernel<1024, 4, 1>, 1024, 4, 1, "RW 1024 G 4 WPB 1");
        runB(wave_panel_kernel<512, 4, 4>, 512, 4, 4, "RW 512 G 4 WPB 4");
        return 0;
    }
    if (TRsel == 4096) {
        run(panel_kernel<4096, 256, 256>, 4096, 256, 256, "TR 4096 CH 256 NT 256");
        run(panel_kernel<4096, 512, 256>, 4096, 512, 256, "TR 4096 CH 512 NT 256");
        run(panel_kernel<4096, 1024, 256>, 4096, 1024, 256, "TR 4096 CH 1024 NT 256");
        run(panel_kernel<4096, 1024, 512>, 4096, 1024, 512, "TR 4096 CH 1024 NT 512");
    } else if (TRsel == 8192) {
        run(panel_kernel<8192, 1024, 512>, 8192, 1024, 512, "TR 8192 CH 1024 NT 512");
        run(panel_kernel<8192, 2048, 512>, 8192, 2048, 512, "TR 8192 CH 2048 NT 512");
        run(panel_kernel<8192, 1024, 1024>, 8192, 1024, 1024, "TR 8192 CH 1024 NT 1024");
    } else {
        run(panel_kernel<2048, 512, 256>, 2048, 512, 256, "TR 2048 CH 512 NT 256");
        run(panel_kernel<2048, 256, 256>, 2048, 256, 256, "TR 2048 CH 256 NT 256");
    }
    return 0;
}
