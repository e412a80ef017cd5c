// its_host.cpp — host-only helpers of libits_hip.so (no device code, callable without a GPU):
// multicolour ordering for the Gauss-Seidel sweep and symmetric permutation of a CRS matrix.
//
// Why a permutation and not a different kernel: the reference's Gauss-Seidel is a natural-order
// forward substitution (kernels.hpp:451-473).  The library executes it level by level (rows of
// one dependency level in parallel, identical arithmetic per row).  On P A P^T with rows ordered
// colour by colour, the dependency levels ARE the colours, so the same level-scheduled sweep is
// the multicolour sweep — and the CPU reference run on the same permuted matrix gives bit-
// comparable iterates (SURVEY.md §7 "GS ordering").
#include "../../include/its_hip.h"
#include "its_hostplan.h"
#include "its_ap.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <omp.h>
#include <vector>

// ---------------------------------------------------------------------------------------
// OpenMP-parallel set-up passes (its_hostplan.h).  Every result is independent of the thread count: threads own
// contiguous, ordered ranges and per-thread results are merged in thread order.
// ---------------------------------------------------------------------------------------
namespace its_host {

int validate_crs(int n_rows, int n_cols, int64_t nnz, const int *rp, const int *col)
{
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(max : bad)
    for (int i = 0; i < n_rows; ++i) if (rp[i + 1] < rp[i]) bad = std::max(bad, 1);
    if (bad) return 1;
#pragma omp parallel for schedule(static) reduction(max : bad)
    for (int64_t k = 0; k < nnz; ++k) if (col[k] < 0 || col[k] >= n_cols) bad = std::max(bad, 2);
    return bad;
}

namespace {
// 1024-slot open-addressing table over at most 256 distinct 64-bit keys
struct SmallMap {
    uint64_t keys[1024]; int16_t ids[1024]; int n = 0;
    SmallMap() { for (auto &i : ids) i = -1; }
    static size_t slot(uint64_t k) { return (size_t)((k * 0x9E3779B97F4A7C15ull) >> 54); }
    int find(uint64_t k) const
    {
        size_t p = slot(k);
        while (ids[p] >= 0 && keys[p] != k) p = (p + 1) & 1023;
        return ids[p];
    }
    // returns the id, or -1 when the table already holds 256 keys
    int insert(uint64_t k)
    {
        size_t p = slot(k);
        while (ids[p] >= 0 && keys[p] != k) p = (p + 1) & 1023;
        if (ids[p] >= 0) return ids[p];
        if (n == 256) return -1;
        ids[p] = (int16_t)n; keys[p] = k;
        return n++;
    }
};

// Distinct keys of a sequence in FIRST-OCCURRENCE order, at most 256: key(k) for k in [0, count), threads own ordered
// contiguous ranges, so concatenating their discoveries in thread order (dropping repeats) is the global order.
template <class KeyAt>
bool distinct_in_order(int64_t count, KeyAt key_at, std::vector<uint64_t> &out)
{
    const int T = omp_get_max_threads();
    std::vector<std::vector<uint64_t>> found((size_t)T);
    bool ok = true;
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
        const int64_t a = count * t / nt, b = count * (t + 1) / nt;
        SmallMap m;
        std::vector<uint64_t> mine;
        uint64_t last = 0; bool have_last = false; bool fine = true;
        for (int64_t k = a; k < b && fine; ++k) {
            const uint64_t key = key_at(k);
            if (have_last && key == last) continue;
            last = key; have_last = true;
            const int before = m.n;
            if (m.insert(key) < 0) fine = false;
            else if (m.n != before) mine.push_back(key);
        }
        if (!fine) {
#pragma omp atomic write
            ok = false;
        }
        found[(size_t)t].swap(mine);
    }
    if (!ok) return false;
    SmallMap g;
    out.clear();
    for (auto &f : found)
        for (uint64_t key : f) {
            const int before = g.n;
            if (g.insert(key) < 0) return false;
            if (g.n != before) out.push_back(key);
        }
    return true;
}
}  // namespace

bool offset_dict(int n_rows, const int *rp, const int *col, unsigned char *cidx, int *dict256, int *n_off)
{
    *n_off = 0;
    const int64_t nnz = n_rows > 0 ? rp[n_rows] : 0;
    if (nnz == 0) return false;
    // offsets of the entries of a row range are scanned row by row: key(k) needs the row of entry k, so the scan is
    // organised over rows (threads own row ranges, which are entry ranges as well)
    const int T = omp_get_max_threads();
    std::vector<std::vector<uint64_t>> found((size_t)T);
    bool ok = true;
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
        const int a = (int)((int64_t)n_rows * t / nt), b = (int)((int64_t)n_rows * (t + 1) / nt);
        SmallMap m; std::vector<uint64_t> mine; bool fine = true;
        for (int i = a; i < b && fine; ++i)
            for (int k = rp[i]; k < rp[i + 1]; ++k) {
                const uint64_t key = (uint64_t)(uint32_t)(col[k] - i);
                const int before = m.n;
                if (m.insert(key) < 0) { fine = false; break; }
                if (m.n != before) mine.push_back(key);
            }
        if (!fine) {
#pragma omp atomic write
            ok = false;
        }
        found[(size_t)t].swap(mine);
    }
    if (!ok) return false;
    SmallMap g; int n = 0;
    for (auto &f : found)
        for (uint64_t key : f) {
            const int before = g.n;
            if (g.insert(key) < 0) return false;
            if (g.n != before) dict256[n++] = (int)(uint32_t)key;
        }
    for (int i = n; i < 256; ++i) dict256[i] = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_rows; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k) cidx[k] = (unsigned char)g.find((uint64_t)(uint32_t)(col[k] - i));
    *n_off = n;
    return true;
}

int frequent_offsets(int n_rows, const int *rp, const int *col, int *out, int max_out)
{
    if (n_rows < 1 || max_out < 1) return 0;
    const int samples = std::min(n_rows, 4096);
    std::vector<int> offs;
    for (int q = 0; q < samples; ++q) {
        const int i = (int)((int64_t)q * n_rows / samples);
        for (int k = rp[i]; k < rp[i + 1]; ++k) offs.push_back(col[k] - i);
    }
    if (offs.empty()) return 0;
    std::sort(offs.begin(), offs.end());
    std::vector<std::pair<int64_t, int>> freq;                      // (-count, offset): sorts by count descending, then offset
    for (size_t a = 0; a < offs.size();) {
        size_t b = a;
        while (b < offs.size() && offs[b] == offs[a]) ++b;
        freq.emplace_back(-(int64_t)(b - a), offs[a]);
        a = b;
    }
    std::sort(freq.begin(), freq.end());
    int n = 0;
    for (auto &f : freq) {
        if (n >= std::min(max_out, 16) || -f.first * 50 < (int64_t)offs.size()) break;
        out[n++] = f.second;
    }
    return n;
}

bool value_dict(int64_t nnz, const double *val, unsigned char *vidx, double *dict256, int *n_vals)
{
    *n_vals = 0;
    if (nnz == 0) return false;
    std::vector<uint64_t> keys;
    auto key_at = [&](int64_t k) { uint64_t b; memcpy(&b, &val[k], 8); return b; };
    if (!distinct_in_order(nnz, key_at, keys)) return false;
    SmallMap g;
    for (size_t i = 0; i < keys.size(); ++i) { g.insert(keys[i]); memcpy(&dict256[i], &keys[i], 8); }
    for (size_t i = keys.size(); i < 256; ++i) dict256[i] = 0.0;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nnz; ++k) vidx[k] = (unsigned char)g.find(key_at(k));
    *n_vals = (int)keys.size();
    return true;
}

void block_longest(int nb, const int *blk, const int *rp, int *longest)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nb; ++i) {
        int l = 0;
        for (int r = blk[i]; r < blk[i + 1]; ++r) l = std::max(l, rp[r + 1] - rp[r]);
        longest[i] = l;
    }
}

void row_stats(int n_rows, const int *rp, const int *col, const double *val, bool upper, int *max_row_len,
               int64_t *n_upper, int64_t *upper_col_sum, uint64_t *upper_val_xor)
{
    int mx = 0; int64_t nu = 0, cs = 0; uint64_t hx = 0;
#pragma omp parallel for schedule(static) reduction(max : mx) reduction(+ : nu, cs) reduction(^ : hx)
    for (int i = 0; i < n_rows; ++i) {
        mx = std::max(mx, rp[i + 1] - rp[i]);
        if (upper)
            for (int k = rp[i]; k < rp[i + 1]; ++k)
                if (col[k] > i) {
                    uint64_t bits; memcpy(&bits, &val[k], 8);
                    ++nu; cs += col[k]; hx ^= bits * 0x9E3779B97F4A7C15ull + (uint64_t)col[k];
                }
    }
    *max_row_len = mx; *n_upper = nu; *upper_col_sum = cs; *upper_val_xor = hx;
}

void row_maxcol(int n_rows, const int *rp, const int *col, int limit, int *maxcol)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_rows; ++i) {
        int m = -1;
        for (int k = rp[i]; k < rp[i + 1]; ++k) if (col[k] < limit) m = std::max(m, col[k]);
        maxcol[i] = m;
    }
}

void assemble_diag_upper(int n, const int *u_rp, const int *u_col, const double *u_val, const double *diag,
                         int *rp, int *col, double *val)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i <= n; ++i) rp[i] = u_rp[i] + i;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int i = 0; i < n; ++i) {
        int64_t o = (int64_t)u_rp[i] + i;
        col[o] = i; val[o] = diag[i]; ++o;
        for (int k = u_rp[i]; k < u_rp[i + 1]; ++k, ++o) { col[o] = u_col[k]; val[o] = u_val[k]; }
    }
}

// SELL-C-sigma: rows sorted by descending length inside windows of sigma rows (stable: ties keep row order, the layout of
// CUDA_tests/helpers/utilities.cpp:183-361), then the longest row of every chunk of C; returns the stored elements
int64_t sell_order(int n, int n_pad, const int *rp, int sigma, int C, int *order, int *clen)
{
    auto len = [&](int r) { return r < n ? rp[r + 1] - rp[r] : 0; };
    const int n_win = sigma > 1 ? (n_pad + sigma - 1) / sigma : 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_pad; ++i) order[i] = i;
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < n_win; ++w) {
        const int64_t lo = (int64_t)w * sigma, hi = std::min<int64_t>(n_pad, lo + sigma);
        std::stable_sort(order + lo, order + hi, [&](int a, int b) { return len(a) > len(b); });
    }
    const int n_chunks = n_pad / C;
    int64_t tot = 0;
#pragma omp parallel for schedule(static) reduction(+ : tot)
    for (int c = 0; c < n_chunks; ++c) {
        int mx = 0;
        for (int r = 0; r < C; ++r) mx = std::max(mx, len(order[(size_t)c * C + r]));
        clen[c] = mx;
        tot += (int64_t)mx * C;
    }
    return tot;
}

void split_by_magnitude(int n_rows, const int *rp, const int *col, const double *val, double thr,
                        int *rp_d, int *col_d, double *val_d, int *rp_s, int *col_s, float *val_s)
{
    rp_d[0] = 0; rp_s[0] = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_rows; ++i) {
        int d = 0;
        for (int k = rp[i]; k < rp[i + 1]; ++k) d += !(std::abs(val[k]) < thr);        // NaN stays fp64
        rp_d[i + 1] = d; rp_s[i + 1] = rp[i + 1] - rp[i] - d;
    }
    for (int i = 0; i < n_rows; ++i) { rp_d[i + 1] += rp_d[i]; rp_s[i + 1] += rp_s[i]; }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_rows; ++i) {
        int d = rp_d[i], q = rp_s[i];
        for (int k = rp[i]; k < rp[i + 1]; ++k) {
            if (!(std::abs(val[k]) < thr)) { col_d[d] = col[k]; val_d[d] = val[k]; ++d; }
            else { col_s[q] = col[k]; val_s[q] = (float)val[k]; ++q; }
        }
    }
}

void split_by_tiers(int n_rows, const int *rp, const int *col, const double *val, int mode, double t1, double t2,
                    int *rp_d, int *col_d, double *val_d, int *rp_s, int *col_s, float *val_s,
                    int *rp_h, int *col_h, uint16_t *val_h)
{
    rp_d[0] = 0; rp_s[0] = 0; rp_h[0] = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_rows; ++i) {
        int c[3] = {0, 0, 0};
        for (int k = rp[i]; k < rp[i + 1]; ++k) ++c[its_ap::tier(val[k], mode, t1, t2)];
        rp_d[i + 1] = c[0]; rp_s[i + 1] = c[1]; rp_h[i + 1] = c[2];
    }
    for (int i = 0; i < n_rows; ++i) { rp_d[i + 1] += rp_d[i]; rp_s[i + 1] += rp_s[i]; rp_h[i + 1] += rp_h[i]; }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_rows; ++i) {
        int d = rp_d[i], q = rp_s[i], g = rp_h[i];
        for (int k = rp[i]; k < rp[i + 1]; ++k) {
            const int t = its_ap::tier(val[k], mode, t1, t2);
            if (t == its_ap::TIER_FP64) { col_d[d] = col[k]; val_d[d] = val[k]; ++d; }
            else if (t == its_ap::TIER_FP32) { col_s[q] = col[k]; val_s[q] = (float)val[k]; ++q; }
            else { col_h[g] = col[k]; val_h[g] = its_ap::half_bits((float)val[k]); ++g; }
        }
    }
}

}  // namespace its_host

// extract_diag (utility_funcs.hpp:157-174): D[i] = a_ii, rows without a diagonal entry keep 0 (the last one wins if a
// row lists its diagonal twice, as in the reference's loop)
extern "C" int its_extract_diag(int n, const int *row_ptr, const int *col, const double *val, double *D)
{
    if (n < 0 || !row_ptr || !D || (n > 0 && row_ptr[n] > 0 && (!col || !val))) return ITS_EINVAL;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double d = 0.0;
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) if (col[k] == i) d = val[k];
        D[i] = d;
    }
    return ITS_OK;
}

// split_L_U + convert_to_crs (utility_funcs.hpp:240-324, methods/gauss_seidel.hpp:433-450): strictly lower and
// strictly upper part in CRS, entry order inside a row kept.  Outputs sized like the inputs (n + 1 / nnz).
extern "C" int its_split_lu(int n, const int *row_ptr, const int *col, const double *val,
                            int *Lp, int *Lc, double *Lv, int *Up, int *Uc, double *Uv)
{
    if (n < 0 || !row_ptr || !Lp || !Up) return ITS_EINVAL;
    const int64_t nnz = n ? row_ptr[n] : 0;
    if (nnz > 0 && (!col || !val || !Lc || !Lv || !Uc || !Uv)) return ITS_EINVAL;
    Lp[0] = 0; Up[0] = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int l = 0, u = 0;
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) { l += col[k] < i; u += col[k] > i; }
        Lp[i + 1] = l; Up[i + 1] = u;
    }
    for (int i = 0; i < n; ++i) { Lp[i + 1] += Lp[i]; Up[i + 1] += Up[i]; }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int l = Lp[i], u = Up[i];
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            if (col[k] < i) { Lc[l] = col[k]; Lv[l] = val[k]; ++l; }
            else if (col[k] > i) { Uc[u] = col[k]; Uv[u] = val[k]; ++u; }
        }
    }
    return ITS_OK;
}

// Greedy distance-1 colouring of the graph of A + A^T in a fixed vertex order — rows of at least HUB_ROW entries
// first, then all others, each group in ascending index — computed in parallel and still EXACTLY the sequential greedy
// result of that order (Jones-Plassmann with dependency counters: a vertex is coloured as soon as all its neighbours
// that precede it in the order are; what it gets depends on those neighbours only, never on the schedule).
// Why hubs first: a colour step of the sweep cannot finish before the serial walk of its longest row;
// with the plain natural order the long rows of a power-law matrix end up spread over nearly all colours (13 of 14
// at 10 M rows), with this order they share the first few.  Matrices without such rows (every stencil) are coloured
// exactly as by the natural-order greedy: 5-/7-point -> 2 colours (red-black), 27-point -> 8.
static constexpr int HUB_ROW = 512;

extern "C" int its_multicolor_order(int n, const int *row_ptr, const int *col, int *perm_new2old, int *n_colors)
{
    if (n < 0 || !row_ptr || (n > 0 && !perm_new2old)) return ITS_EINVAL;
    const int64_t nnz = n ? row_ptr[n] : 0;
    if (nnz > 0 && !col) return ITS_EINVAL;
    {
        int bad = 0;
#pragma omp parallel for schedule(static) reduction(max : bad)
        for (int64_t k = 0; k < nnz; ++k) if (col[k] < 0 || col[k] >= n) bad = 1;
        if (bad) return ITS_EINVAL;
    }
    // structure of A^T (rows that reference vertex j): a_ij != 0 or a_ji != 0 both forbid i and j in one colour
    std::vector<int64_t> tp((size_t)n + 1, 0);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nnz; ++k) {
#pragma omp atomic
        ++tp[(size_t)col[k] + 1];
    }
    for (int i = 0; i < n; ++i) tp[i + 1] += tp[i];
    std::vector<int> tcol((size_t)nnz);
    {
        std::vector<int64_t> fill(tp.begin(), tp.end() - 1);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i)
            for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
                int64_t w;
#pragma omp atomic capture
                w = fill[col[k]]++;
                tcol[(size_t)w] = i;          // order inside a list varies with the schedule; only membership is used
            }
    }
    // position of a vertex in the colouring order: hubs first
    std::vector<unsigned char> hub((size_t)n, 0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) hub[i] = row_ptr[i + 1] - row_ptr[i] >= HUB_ROW;
    auto before = [&](int u, int v) {           // u is coloured before v
        return hub[u] != hub[v] ? hub[u] != 0 : u < v;
    };
    std::vector<int> pending((size_t)n, 0), colour((size_t)n, -1);
    std::vector<int> frontier, next;
#pragma omp parallel
    {
        std::vector<int> mine;
#pragma omp for schedule(static)
        for (int i = 0; i < n; ++i) {
            int c = 0;
            for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) c += col[k] != i && before(col[k], i);
            for (int64_t k = tp[i]; k < tp[i + 1]; ++k) c += tcol[(size_t)k] != i && before(tcol[(size_t)k], i);
            pending[i] = c;
            if (c == 0) mine.push_back(i);
        }
#pragma omp critical
        frontier.insert(frontier.end(), mine.begin(), mine.end());
    }
    int nc = 0;
    int64_t done = 0;
    while (!frontier.empty()) {
        const int64_t nf = (int64_t)frontier.size();
        int nc_round = 0;
#pragma omp parallel reduction(max : nc_round)
        {
            std::vector<uint64_t> used;
            // 1. colours: smallest one no earlier neighbour holds (all of them are coloured)
#pragma omp for schedule(dynamic, 256)
            for (int64_t f = 0; f < nf; ++f) {
                const int i = frontier[(size_t)f];
                used.assign(used.size(), 0);
                auto see = [&](int j) {
                    if (j == i) return;
                    const int cj = colour[j];
                    if (cj < 0 || !before(j, i)) return;
                    if ((size_t)(cj >> 6) >= used.size()) used.resize((size_t)(cj >> 6) + 1, 0);
                    used[(size_t)(cj >> 6)] |= 1ull << (cj & 63);
                };
                for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) see(col[k]);
                for (int64_t k = tp[i]; k < tp[i + 1]; ++k) see(tcol[(size_t)k]);
                int c = 0;
                for (size_t wd = 0;; ++wd) {
                    const uint64_t bits = wd < used.size() ? used[wd] : 0;
                    if (~bits) { c = (int)(wd * 64) + __builtin_ctzll(~bits); break; }
                }
                colour[i] = c;
                nc_round = std::max(nc_round, c + 1);
            }
            // (implicit barrier: every colour of this round is written before anything below reads `colour`; the
            // vertices of one round are never adjacent, so step 1 only read colours of earlier rounds)
            // 2. release the later neighbours
            std::vector<int> mine;
#pragma omp for schedule(dynamic, 256)
            for (int64_t f = 0; f < nf; ++f) {
                const int i = frontier[(size_t)f];
                auto release = [&](int j) {
                    if (j == i || !before(i, j)) return;
                    int left;
#pragma omp atomic capture
                    left = --pending[j];
                    if (left == 0) mine.push_back(j);
                };
                for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k) release(col[k]);
                for (int64_t k = tp[i]; k < tp[i + 1]; ++k) release(tcol[(size_t)k]);
            }
#pragma omp critical
            next.insert(next.end(), mine.begin(), mine.end());
        }
        nc = std::max(nc, nc_round);
        done += nf;
        frontier.swap(next);
        next.clear();
    }
    if (done != n) return ITS_EINVAL;                 // cannot happen: the order is total
    // stable counting sort by colour: ascending original index inside a colour
    std::vector<int> start((size_t)nc + 1, 0);
    for (int i = 0; i < n; ++i) ++start[(size_t)colour[i] + 1];
    for (int c = 0; c < nc; ++c) start[c + 1] += start[c];
    for (int i = 0; i < n; ++i) perm_new2old[start[colour[i]]++] = i;
    if (n_colors) *n_colors = nc;
    return ITS_OK;
}

extern "C" int its_permute_crs(int n, const int *row_ptr, const int *col, const double *val,
                               const int *perm_new2old, int *row_ptr_out, int *col_out, double *val_out)
{
    if (n < 0 || !row_ptr || !perm_new2old || !row_ptr_out) return ITS_EINVAL;
    std::vector<int> old2new((size_t)n, -1);
    for (int i = 0; i < n; ++i) {
        const int o = perm_new2old[i];
        if (o < 0 || o >= n || old2new[o] != -1) return ITS_EINVAL;      // not a permutation
        old2new[o] = i;
    }
    row_ptr_out[0] = 0;
    for (int i = 0; i < n; ++i) { const int o = perm_new2old[i]; row_ptr_out[i + 1] = row_ptr_out[i] + (row_ptr[o + 1] - row_ptr[o]); }
#pragma omp parallel
    {
        std::vector<std::pair<int, double>> row;
#pragma omp for schedule(dynamic, 4096)
        for (int i = 0; i < n; ++i) {
            const int o = perm_new2old[i];
            row.clear();
            for (int k = row_ptr[o]; k < row_ptr[o + 1]; ++k) row.emplace_back(old2new[col[k]], val[k]);
            std::stable_sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
            int w = row_ptr_out[i];
            for (const auto &e : row) { col_out[w] = e.first; val_out[w] = e.second; ++w; }
        }
    }
    return ITS_OK;
}
