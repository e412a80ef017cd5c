// its_lines.hip — the NATURAL-ORDER triangular sweep of a stencil matrix as one launch: a wavefront pipelined over grid lines.
//
// spltsv_crs (kernels.hpp:451-473) walks the rows in order; the level schedule of its_api.hip reproduces it exactly, but a
// 3-D stencil in natural order has one level per hyperplane x + y + z — 766 launches of scattered rows at 256^3, 4.5 us
// each whatever they hold (3.5 ms per sweep; its_solve -gs, the reference's default order).  What the dependencies really
// are: with strictly lower offsets {-1, -a, -b} (a = rows per grid line, b = rows per plane) row (p, y, z) reads
// (p - 1, y, z), (p, y - 1, z) and (p, y, z - 1).  So
//   * a LANE owns a line and walks it row by row, the chain over the -1 entry staying in its registers;
//   * a WAVE owns 64 neighbouring lines — 16 (y) x 4 (z) of a 3-D grid, 64 consecutive ones of a 2-D grid — its lanes
//     skewed by ONE ROW per line: lane (ty, tz) works on row i - rho at the wave's i-th row iteration, rho = ty + tz, so
//     the values it needs from the lines y - 1 and z - 1 are what the lanes next to it produced one iteration earlier, and
//     they come across the wave by DPP (row_shr:1 / wave_shr:1, the lane at the start of a row keeping what it had) and
//     ds_bpermute (16 lanes down): no memory, no barrier;
//   * a WORKGROUP owns a tile of 16 x 16 lines (256 x 1 in 2-D) = four such waves, each 4 rows + its own skew behind the
//     one below it; rows are handed from wave to wave in STEPS of LN_CH = 4 through LDS (double-buffered, one LDS-only
//     barrier per step);
//   * tiles are taken in row-major order from a counter, so whatever a tile waits for has been started before it; the lines
//     on a tile's high edges are also written, row-major, into a small edge buffer that is preset to a NaN pattern no sweep
//     produces; a fifth wave of the neighbouring tile polls the rows it needs there with agent-scope loads (data and arrival
//     flag are the same word: no fence, no counter to order against) and puts them into LDS a step ahead.  Every poll is
//     bounded by the wall clock; a sweep that gives up says so (sticky flag) and the host reports it instead of a result.
// Per row the operations are the reference's in its order — fma over the row's entries by ascending column from 0.0, then
// (rhs - acc) / D — so x is bit-identical to the level schedule's and to spltsv_crs.
//
// The first complete versions skewed the lines by a whole step (4 rows, LDS + barrier between ALL lines): a tile's last
// line then starts 30 steps after its first and a hop from tile to tile costs 15 steps + the distance requests travel
// ahead + a memory round trip, 25 - 30 us; at 256^3 thirty hops were 0.9 of the sweep's 1.02 ms (Gauss-Seidel trip 1.44 ms;
// level schedule 3.63).  With one row per line a tile's last line starts 39 ROWS after its first.
// Eligibility is verified row by row on the device when the image is built (lines_build_kernel): every entry of L is one
// of the three offsets, present only where the grid has that neighbour (no wrap-around), columns ascending.
#include "its_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace its {

constexpr unsigned long long LINES_PENDING = 0x7ff8dead0badf11dull;    // the preset pattern of the edge buffer (a quiet NaN with a payload)
constexpr long long LINES_WAIT_TICKS = 5000000;                         // 50 ms of the 100 MHz wall clock per poll
constexpr int LINES_WG = 256;
constexpr int LN_CH = 4;            // rows per step of a lane
constexpr int LN_K = 5;             // operand sets in flight per lane: a step's operands are requested LN_K - 1 steps early
constexpr int LN_KE = 4;            // the same for the neighbouring tiles' rows (edge wave)
constexpr int LN_EDGE = 32;         // lines on a tile's low edges: <= 16 along z (ty == 0) + <= 16 along y (tz == 0)
constexpr int LN_G = 4;             // steps per workgroup of the kernels that build and (un)pack
constexpr int LN_PAD = 4;           // rows of padding on both ends of a line of the edge buffer
constexpr int LINES_THREADS = LINES_WG + ITS_WAVE;   // four waves of lanes that own lines + the edge wave
typedef double lines_v2d __attribute__((ext_vector_type(2)));
typedef double lines_v2du __attribute__((ext_vector_type(2), aligned(8)));    // row-order side: a window starts at any row
struct LinesGeom { int n, a, b, ny, nz, nby, nbz, nsteps, tslots, start, mute; const int *sbase; const int4 *tabs; };

// ---- Geometry.  Lane t of a tile: 3-D (ty, tz) = (t & 15, t >> 4) of 16 x 16 lines, 2-D ty = t of 256 x 1.
// rho(t) = rows the lane is behind lane 0: one per line inside a wave, and from wave to wave 4 more than the last lane
// the next wave reads from (3-D: lanes (ty, 4 w + 3) -> (ty, 4 w + 4): 7 per wave; 2-D: lane 64 w + 63 -> 64 w + 64: 67),
// so that the window of rows a consumer works on at step S is exactly the one its producer finished at step S - 1.
__host__ __device__ __forceinline__ int ln_rho(int t, bool two_d) { return two_d ? (t & 63) + 67 * (t >> 6) : (t & 15) + ((t >> 4) & 3) + 7 * (t >> 6); }
__host__ __device__ __forceinline__ int ln_rho_max(bool two_d) { return two_d ? 63 + 3 * 67 : 15 + 3 + 21; }
// At step S lane t works on rows [4 S - rho, 4 S - rho + 3] of its line (those inside [0, a)); `active` = any of them is.

// ---- The order everything the sweep streams is stored in: BY STEP.  At one step the lanes of a wave are a line apart (and
// at different rows); from row-major storage every lane touches a cache line of its own and the texture path, which
// takes one line per cycle, is the bound (12 loads x 64 lines x 4 waves = 1.3 us per step measured).  So the operands of
// the windows that run at step S of a tile lie side by side in lane order, followed by ONE neutral slot (no entries,
// diagonal 1, right-hand side 0) that every lane without a window at that step reads: rows outside a line compute
// exactly 0.0 from 0.0 — nothing a line's first row or a neighbour could pick up.
//     slot(tile, S, lane) = tile * tslots + sbase[S] + rank(S, lane)         rank = active lanes below; cnt(S) for the rest
// and inside a step's block one plane per pair of rows (u >> 1) of each operand (cnt(S) + 1 slots wide), so that a
// wave's 16-byte loads are consecutive memory:
//     img [((slot0 * NA + k) * 2 + (u >> 1)) * (cnt + 1) * 2 + rank * 2 + (u & 1)]     k: [-b,] -a, -1, diagonal  (NA = 3 | 4)
//     rimg, ximg [(slot0 * 2 + (u >> 1) * (cnt + 1) + rank) * 2 + (u & 1)]             right-hand side in, x out
//     mimg[slot0 + rank]                                                                presence bits, a byte per row
// tabs[S] = {cnt(S), active lanes of wave 0, of waves 0..1, of waves 0..2}: a lane's rank is a table word + the count of
// active lanes below it in its own wave (ballot).
struct LnWhere { int tile, S, w0, ty, tz, y, z, rank, cnt; bool active, line_valid; size_t slot0; };
__device__ __forceinline__ void ln_where(const LinesGeom &g, int64_t bid, int t, LnWhere (&w)[LN_G], int (*s_w)[LINES_WG / ITS_WAVE])
{
    const bool two_d = g.b == 0;
    const int groups = (g.nsteps + LN_G - 1) / LN_G;
    const int tile = (int)(bid / groups), S0 = (int)(bid % groups) * LN_G;
    const int by = tile % g.nby, bz = tile / g.nby;
    const int lane = t & (ITS_WAVE - 1), wv = t / ITS_WAVE;
    const int rho = ln_rho(t, two_d);
    unsigned long long m[LN_G];
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        LnWhere &x = w[j];
        x.tile = tile; x.S = S0 + j;
        x.ty = two_d ? t : (t & 15); x.tz = two_d ? 0 : (t >> 4);
        x.y = by * (two_d ? 256 : 16) + x.ty; x.z = bz * (two_d ? 1 : 16) + x.tz;
        x.w0 = 4 * x.S - rho;
        x.active = x.S < g.nsteps && x.w0 > -LN_CH && x.w0 < g.a;
        x.line_valid = x.y < g.ny && x.z < g.nz;
        m[j] = __ballot(x.active);
        if (lane == 0) s_w[j][wv] = __popcll(m[j]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        int below = 0, cnt = 0;
#pragma unroll
        for (int i = 0; i < LINES_WG / ITS_WAVE; ++i) { if (i < wv) below += s_w[j][i]; cnt += s_w[j][i]; }
        w[j].rank = w[j].active ? below + __popcll(m[j] & ((1ull << lane) - 1ull)) : cnt;
        w[j].cnt = cnt;
        w[j].slot0 = (size_t)tile * g.tslots + (w[j].S < g.nsteps ? g.sbase[w[j].S] : 0);
    }
}
__device__ __forceinline__ size_t ln_edge_off(const LinesGeom &g, int tile, int side, int idx)
{ return (((size_t)tile * 2 + side) * 16 + idx) * (size_t)(g.a + 2 * LN_PAD) + LN_PAD; }

// Row-order side of the two passes below.  A lane's LN_G windows are 16 consecutive rows of its line, and the lines of a
// tile lie a whole line apart: lane by lane every 16-byte access would touch a cache line of its own.  So the workgroup
// moves the tile's 256 x 16 rows through LDS: piece p = (line p / 8, rows 2 (p % 8) ..+1), eight consecutive threads on the
// 128 consecutive bytes of one line.
constexpr int LN_ROWS = LN_G * LN_CH;                         // rows of a line per workgroup
struct LnPiece { int line, r; bool ok0, ok1; int64_t at; };
__device__ __forceinline__ LnPiece ln_piece(const LinesGeom &g, int tile, int S0, int p)
{
    const bool two_d = g.b == 0;
    LnPiece q;
    q.line = p / (LN_ROWS / 2);
    const int part = p % (LN_ROWS / 2);
    const int by = tile % g.nby, bz = tile / g.nby;
    const int ty = two_d ? q.line : (q.line & 15), tz = two_d ? 0 : (q.line >> 4);
    const int y = by * (two_d ? 256 : 16) + ty, z = bz * (two_d ? 1 : 16) + tz;
    q.r = 4 * S0 - ln_rho(q.line, two_d) + 2 * part;
    const bool lv = y < g.ny && z < g.nz;
    q.ok0 = lv && q.r >= 0 && q.r < g.a;
    q.ok1 = lv && q.r + 1 >= 0 && q.r + 1 < g.a;
    q.at = ((int64_t)z * g.ny + y) * g.a + q.r;
    return q;
}

// The (<= 3) entries of each row into fixed slots — offset -b, -a, -1, which is ascending column order — flags[0] |= 1 when
// a row is not of that form, |= 2 when a row lacks an entry where the grid has the neighbour (the sweep must then skip by
// presence bits).  Rows outside the line or the grid, and the neutral slot: no entries, diagonal 1.
__global__ __launch_bounds__(LINES_WG) void lines_build_kernel(
    LinesGeom g, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ D,
    double *__restrict__ img, unsigned int *__restrict__ mimg, int *__restrict__ flags)
{
    __shared__ int s_w[LN_G][LINES_WG / ITS_WAVE];
    const int t = threadIdx.x;
    LnWhere ws[LN_G];
    ln_where(g, blockIdx.x, t, ws, s_w);
    const int NA = g.b > 0 ? 4 : 3;
    bool bad = false, holes = false;
    for (int j = 0; j < LN_G; ++j) {
        const LnWhere &w = ws[j];
        if (w.S >= g.nsteps) continue;
        const bool writes_neutral = t == 0;                  // (one writer for the step's neutral slot)
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: this lane's own slot (if it has a window); pass 1: the neutral slot (one lane)
            if (pass == 0 && !w.active) continue;
            if (pass == 1 && !writes_neutral) continue;
            const int rank = pass == 0 ? w.rank : w.cnt;
            double *o = img + w.slot0 * NA * 4 + (size_t)rank * 2;
            unsigned int mword = 0;
            for (int u = 0; u < 4; ++u) {
                double v[3] = {0.0, 0.0, 0.0}, dg = 1.0;
                int m = 0;
                const int p = w.w0 + u;
                if (pass == 0 && w.line_valid && p >= 0 && p < g.a) {
                    const int r = (w.z * g.ny + w.y) * g.a + p;
                    dg = D[r];
                    int prev = -1;
                    for (int jj = rp[r]; jj < rp[r + 1]; ++jj) {
                        const int c = col[jj], off = r - c;
                        bad |= c <= prev; prev = c;
                        int slot = -1;
                        if (off == 1 && p >= 1) slot = 2;
                        else if (off == g.a && w.y >= 1) slot = 1;
                        else if (g.b > 0 && off == g.b) slot = 0;
                        if (slot < 0 || (m >> slot & 1)) { bad = true; continue; }
                        m |= 1 << slot; v[slot] = val[jj];
                    }
                    const int full = (p >= 1 ? 4 : 0) | (w.y >= 1 ? 2 : 0) | ((g.b > 0 && w.z >= 1) ? 1 : 0);
                    holes |= m != full;
                }
                const size_t plane = (size_t)(w.cnt + 1) * 2, e = (size_t)(u >> 1) * plane + (u & 1);
                int k = 0;
                if (g.b > 0) o[(size_t)(k++) * 2 * plane + e] = v[0];
                o[(size_t)(k++) * 2 * plane + e] = v[1];
                o[(size_t)(k++) * 2 * plane + e] = v[2];
                o[(size_t)k * 2 * plane + e] = dg;
                mword |= (unsigned int)m << (8 * u);
            }
            mimg[w.slot0 + rank] = mword;
        }
    }
    if (bad) atomicOr(flags, 1);
    if (holes) atomicOr(flags, 2);
}

// In front of every sweep: the right-hand side into that order (rimg, 0.0 for rows outside the lines and in the neutral
// slots), the pending pattern into the edge buffer's rows of the lines on the tiles' high edges, tile counter := 0.
// Behind it: x out of ximg into row order.  Both are streaming passes.
__global__ __launch_bounds__(LINES_WG) void lines_pack_kernel(
    LinesGeom g, const double *__restrict__ rhs, double *__restrict__ rimg, unsigned long long *__restrict__ xedge, int *counter)
{
    __shared__ int s_w[LN_G][LINES_WG / ITS_WAVE];
    const int t = threadIdx.x;
    const bool two_d = g.b == 0;
    LnWhere ws[LN_G];
    ln_where(g, blockIdx.x, t, ws, s_w);
    if (blockIdx.x == 0 && t == 0) *counter = 0;
    __shared__ double s_t[LINES_WG][LN_ROWS + 1];
    {
        const int tile = ws[0].tile, S0 = ws[0].S;
#pragma unroll
        for (int k = 0; k < LN_ROWS / 2; ++k) {
            const LnPiece q = ln_piece(g, tile, S0, t + LINES_WG * k);
            double v0 = 0.0, v1 = 0.0;
            if (q.ok0 && q.ok1) { const lines_v2du a = *reinterpret_cast<const lines_v2du *>(rhs + q.at); v0 = a.x; v1 = a.y; }
            else { if (q.ok0) v0 = rhs[q.at]; if (q.ok1) v1 = rhs[q.at + 1]; }
            const int c = 2 * ((t + LINES_WG * k) % (LN_ROWS / 2));
            s_t[q.line][c] = v0; s_t[q.line][c + 1] = v1;
        }
    }
    __syncthreads();
    double r[LN_G][LN_CH];
#pragma unroll
    for (int j = 0; j < LN_G; ++j)
#pragma unroll
        for (int u = 0; u < LN_CH; ++u) r[j][u] = s_t[t][j * LN_CH + u];
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        const LnWhere &w = ws[j];
        if (w.S >= g.nsteps) continue;
        const size_t plane = (size_t)(w.cnt + 1) * 2;
        if (w.active) {
            const size_t o = (w.slot0 * 2 + w.rank) * 2;
            lines_v2d a01, a23; a01.x = r[j][0]; a01.y = r[j][1]; a23.x = r[j][2]; a23.y = r[j][3];
            *reinterpret_cast<lines_v2d *>(rimg + o) = a01; *reinterpret_cast<lines_v2d *>(rimg + o + plane) = a23;
            if (w.line_valid) {
                const bool pub_a = w.ty == (two_d ? 255 : 15) && w.y + 1 < g.ny;
                const bool pub_b = !two_d && w.tz == 15 && w.z + 1 < g.nz;
#pragma unroll
                for (int u = 0; u < LN_CH; ++u) {
                    const int p = w.w0 + u;
                    if (p < 0 || p >= g.a) continue;
                    if (pub_a) xedge[ln_edge_off(g, w.tile, 0, two_d ? 0 : w.tz) + p] = LINES_PENDING;
                    if (pub_b) xedge[ln_edge_off(g, w.tile, 1, w.ty) + p] = LINES_PENDING;
                }
            }
        }
        if (t == 0) {                                                          // the step's neutral slot
            const size_t o = (w.slot0 * 2 + w.cnt) * 2;
            lines_v2d z; z.x = 0.0; z.y = 0.0;
            *reinterpret_cast<lines_v2d *>(rimg + o) = z; *reinterpret_cast<lines_v2d *>(rimg + o + plane) = z;
        }
    }
}

__global__ __launch_bounds__(LINES_WG) void lines_unpack_kernel(LinesGeom g, const double *__restrict__ ximg, double *__restrict__ x)
{
    __shared__ int s_w[LN_G][LINES_WG / ITS_WAVE];
    const int t = threadIdx.x;
    LnWhere ws[LN_G];
    ln_where(g, blockIdx.x, t, ws, s_w);
    lines_v2d r01[LN_G], r23[LN_G];
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        const LnWhere &w = ws[j];
        r01[j].x = 0.0; r01[j].y = 0.0; r23[j] = r01[j];
        if (!(w.active && w.line_valid)) continue;
        const size_t o = (w.slot0 * 2 + w.rank) * 2, plane = (size_t)(w.cnt + 1) * 2;
        r01[j] = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(ximg + o));
        r23[j] = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(ximg + o + plane));
    }
    __shared__ double s_t[LINES_WG][LN_ROWS + 1];
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        s_t[t][j * LN_CH] = r01[j].x; s_t[t][j * LN_CH + 1] = r01[j].y; s_t[t][j * LN_CH + 2] = r23[j].x; s_t[t][j * LN_CH + 3] = r23[j].y;
    }
    __syncthreads();
    {
        const int tile = ws[0].tile, S0 = ws[0].S;
#pragma unroll
        for (int k = 0; k < LN_ROWS / 2; ++k) {
            const LnPiece q = ln_piece(g, tile, S0, t + LINES_WG * k);
            const int c = 2 * ((t + LINES_WG * k) % (LN_ROWS / 2));
            // (a row inside its line always lies in a window that has run: every row of every line is some step's)
            if (q.ok0 && q.ok1) { lines_v2du a; a.x = s_t[q.line][c]; a.y = s_t[q.line][c + 1]; *reinterpret_cast<lines_v2du *>(x + q.at) = a; }
            else { if (q.ok0) x[q.at] = s_t[q.line][c]; if (q.ok1) x[q.at + 1] = s_t[q.line][c + 1]; }
        }
    }
}

#ifndef LN_DEBUG
#define LN_DEBUG 0                  // 1: count the edge wave's slow-path entries and re-polls (printed after every sweep)
#endif

#if LN_DEBUG
__device__ long long ln_dbg[4096][4];      // per tile: taken, first neighbour rows in hand, done, slow-path entries (lanes)
#endif
// One window's operands: the entries of its rows (slots -b, -a, -1), diagonal, right-hand side, presence bits.  None of
// them depends on x: they are requested LN_K - 1 steps ahead, by UNCONDITIONAL loads — straight-line code, so that the
// compiler's wait counts stay exact (with loads under divergent branches or next to a polling loop it drains everything
// in flight every step: a memory round trip per step, 1.2 .. 11 us in the first versions).
struct LnSet { double v0[LN_CH], v1[LN_CH], v2[LN_CH], dd[LN_CH], rr[LN_CH]; unsigned int m, vo; size_t slot0; int cnt1; };

// Addresses as (uniform base) + (32-bit lane offset): the bases live in scalar registers and the loads take them as such —
// a lane in this kernel is alone on its SIMD, every vector instruction it does not issue is time off the step.
template <bool HAVE_B, bool MASKED>
__device__ __forceinline__ void ln_load(LnSet &c, size_t slot0, int cnt1, int rank, const double *__restrict__ img, const unsigned int *__restrict__ mimg,
                                        const double *__restrict__ rimg)
{
    constexpr int NA = HAVE_B ? 4 : 3;
    const size_t plane_b = (size_t)cnt1 * 16;                    // (a plane: cnt + 1 pairs of doubles)
    const char *ib = reinterpret_cast<const char *>(img) + slot0 * (NA * 32);
    const char *rb = reinterpret_cast<const char *>(rimg) + slot0 * 32;
    const unsigned int vo = (unsigned int)rank * 16u;
    c.slot0 = slot0; c.cnt1 = cnt1; c.vo = vo;
#pragma unroll
    for (int u = 0; u < LN_CH; u += 2) {
        const int h = u >> 1;
        int k = 0;
        c.v0[u] = 0.0; c.v0[u + 1] = 0.0;
        if (HAVE_B) { const lines_v2d q0 = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(ib + (size_t)((k++) * 2 + h) * plane_b + vo)); c.v0[u] = q0.x; c.v0[u + 1] = q0.y; }
        const lines_v2d q1 = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(ib + (size_t)((k++) * 2 + h) * plane_b + vo));
        const lines_v2d q2 = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(ib + (size_t)((k++) * 2 + h) * plane_b + vo));
        const lines_v2d qd = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(ib + (size_t)(k * 2 + h) * plane_b + vo));
        const lines_v2d qr = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(rb + (size_t)h * plane_b + vo));
        c.v1[u] = q1.x; c.v1[u + 1] = q1.y; c.v2[u] = q2.x; c.v2[u + 1] = q2.y;
        c.dd[u] = qd.x; c.dd[u + 1] = qd.y; c.rr[u] = qr.x; c.rr[u + 1] = qr.y;
    }
    c.m = MASKED ? __builtin_nontemporal_load(reinterpret_cast<const unsigned int *>(reinterpret_cast<const char *>(mimg) + slot0 * 4 + (vo >> 2))) : 0u;
}

// The slow path of a poll — load again and wait — as inline asm, invisible to the compiler's wait-count bookkeeping (a
// loop with vector-memory loads in it makes it drain everything in flight at the loop's exit, on the fast path too).  The
// asm waits for its own load, and with it for everything older: the compiler's later waits are then trivially true.
__device__ __forceinline__ unsigned long long lines_ld_agent(const void *p)
{
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int lines_ld_agent_int(const void *p)
{
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// barrier that orders LDS only: __syncthreads would also wait for every global load and store in flight
__device__ __forceinline__ void ln_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the value lane - 1 holds (3-D: inside a row of 16 lanes = the 16 lines of a plane; 2-D: across the wave); a lane with
// nothing below it in that range keeps `old`
template <bool TWO_D>
__device__ __forceinline__ double ln_from_lane_below(double old, double v)
{
    constexpr int ctrl = TWO_D ? 0x138 /* wave_shr:1 */ : 0x111 /* row_shr:1 */;
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// the value the lane 16 below holds (the same line of the plane below, 3-D)
__device__ __forceinline__ double ln_from_16_below(double v, int lane)
{
    const int addr = ((lane - 16) & (ITS_WAVE - 1)) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// MASKED: rows may lack an entry where the grid has the neighbour (presence bits decide).  Otherwise the only absent
// entries are those of rows on the grid's low faces: stored as 0.0 they multiply an operand that is finite by
// construction — 0.0 from the edge wave for the lines at y = 0 / z = 0, the exact 0.0 of the rows outside a line in front
// of its first row — and fma(0.0, finite, acc) == acc bit for bit (acc is never -0.0: it starts from +0.0), which saves
// the 15 selects and tests per row-triple of a step.
template <bool HAVE_B, bool HAVE_ACC, bool MASKED>
__global__ __launch_bounds__(LINES_THREADS) void sweep_lines_kernel(
    LinesGeom g, const int4 *__restrict__ tabs, const double *__restrict__ img, const unsigned int *__restrict__ mimg, int *counter, int *fail,
    unsigned long long *fail_host, double *__restrict__ ximg, const double *__restrict__ rimg, double *__restrict__ xedge, double *__restrict__ acc_out)
{
    constexpr bool TWO_D = !HAVE_B;
    constexpr int TY = TWO_D ? 256 : 16, TZ = TWO_D ? 1 : 16;
    // rows 0 .. 255: every line's last window (the next wave reads it one step later); rows 256 ..: the neighbouring tiles'
    // rows for the lines on the low edges (edge wave)
    __shared__ double s_c[2][LINES_WG + LN_EDGE][LN_CH + 1];
    __shared__ int s_tile;
    const int t = threadIdx.x;
    if (t == 0) s_tile = atomicAdd(counter, 1);
    __syncthreads();
    const int tile = __builtin_amdgcn_readfirstlane(s_tile);
#if LN_DEBUG
    if (t == LINES_WG && tile < 4096) { ln_dbg[tile][0] = wall_clock64(); ln_dbg[tile][3] = 0; }
#endif
    const int by = tile % g.nby, bz = tile / g.nby;
    const int nsteps = g.nsteps;
    // The lanes' loop runs whole rounds of LN_K steps without a branch in between (steps past the last one find only
    // neutral slots): with `if (S < nsteps)` around every step the loop has an edge from the first step straight back to
    // its head, and on that edge the step's requests are the youngest loads — the compiler then drains everything in
    // flight at the head of every round (vmcnt(0): a memory round trip per 5 steps, 0.3 us per step measured).
    constexpr int K = HAVE_B ? LN_K - 1 : LN_K;          // (3-D: one set less, or the lanes' registers spill)
    const int nsp = (nsteps + K - 1) / K * K;
    const size_t tile_slot0 = (size_t)tile * g.tslots;
    if (t < LINES_WG) {
        // ---- a lane that owns a line
        const int ty = TWO_D ? t : (t & 15), tz = TWO_D ? 0 : (t >> 4);
        const int y = by * TY + ty, z = bz * TZ + tz;
        const bool valid = y < g.ny && z < g.nz;
        const int64_t row_base = valid ? ((int64_t)z * g.ny + y) * g.a : 0;
        const int rho = ln_rho(t, TWO_D);
        const int lane = t & (ITS_WAVE - 1);
        const int wv = __builtin_amdgcn_readfirstlane(t / ITS_WAVE);
        const int ia = (TWO_D ? t > 0 : ty > 0) ? t - 1 : LINES_WG + tz;         // line y - 1 when it is not the lane below: last wave's, or fetched by the edge wave
        const int ib = tz > 0 ? t - 16 : LINES_WG + 16 + ty;                      // line z - 1 likewise
        const bool z_in_wave = !TWO_D && (t & 48) != 0;
        const bool mute = g.mute != 0 && tile == 0;                            // (test hook: see its_set_option "sweep_lines_fault")
        const bool pub_a = valid && ty == TY - 1 && y + 1 < g.ny && !mute;
        const bool pub_b = HAVE_B && valid && tz == TZ - 1 && z + 1 < g.nz && !mute;
        double *xe_a = xedge + ln_edge_off(g, tile, 0, TWO_D ? 0 : tz);
        double *xe_b = xedge + ln_edge_off(g, tile, 1, ty);
        double xl = 0.0;                                 // the line's newest value
        LnSet set[K];
        // the place of the step whose operands are requested next: its table word one step ahead (a scalar load)
        int Ti = 0, baseI = 0;
        int4 tbI = tabs[0], tbN = tabs[nsteps > 1 ? 1 : 0];      // (a kernel argument of its own: restrict + const is what makes these scalar loads)
#define LN_ISSUE(I) {                                                                                             \
            const int w0i = 4 * Ti - rho;                                                                         \
            const bool ai = w0i > -LN_CH && w0i < g.a;                                                            \
            const unsigned long long mk = __ballot(ai);                                                           \
            const int below = wv == 0 ? 0 : (wv == 1 ? tbI.y : (wv == 2 ? tbI.z : tbI.w));                        \
            const int rank = ai ? below + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mk, 0u)) : tbI.x; \
            ln_load<HAVE_B, MASKED>(set[I], tile_slot0 + (size_t)baseI, tbI.x + 1, rank, img, mimg, rimg);        \
            if (Ti + 1 < nsteps) { baseI += tbI.x + 1; ++Ti; tbI = tbN; tbN = tabs[Ti + 1 < nsteps ? Ti + 1 : nsteps - 1]; } }
        LN_ISSUE(0) LN_ISSUE(1) LN_ISSUE(2)
        if constexpr (K == 5) LN_ISSUE(3)
        static_assert(LN_K == 5, "the step loop is unrolled by hand");
        // what step 0 reads as "the window the wave below finished last": nothing has been written there yet, and LDS keeps
        // whatever the last kernel on this CU left — a NaN there would turn 0.0 * x into NaN in front of a line's first row
        // (found as a poll that never ended: the NaN was a previous sweep's pending pattern, stored as a value)
        _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) s_c[1][t][u] = 0.0;
        ln_barrier();                                    // (the edge wave has written the rows of step 0)
        int buf = 0;
#define LN_STEP(I)                                                                                                \
        {                                                                                                         \
            const int S = S0 + (I);                                                                               \
            LN_ISSUE(((I) + K - 1) % K)                                                                           \
            const LnSet &c = set[I];                                                                              \
            const int w0 = 4 * S - rho;                                                                           \
            const bool act = valid && w0 > -LN_CH && w0 < g.a;                                                    \
            double xal[LN_CH], xbl[LN_CH];                                                                        \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) {                                                   \
                xal[u] = s_c[buf ^ 1][ia][u];                                                                     \
                xbl[u] = HAVE_B ? s_c[buf ^ 1][ib][u] : 0.0;                                                      \
            }                                                                                                     \
            double out[LN_CH];                                                                                    \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) {                                                   \
                /* the neighbouring lines' value of this row: made one row iteration ago by the lanes below */    \
                const double xa = ln_from_lane_below<TWO_D>(xal[u], xl);                                          \
                double xb = 0.0;                                                                                  \
                if (HAVE_B) { const double nb = ln_from_16_below(xl, lane); xb = z_in_wave ? nb : xbl[u]; }       \
                /* the row's chain in CRS order: -b, -a, -1 (ascending columns), from 0.0 */                      \
                const unsigned int m = (c.m >> (8 * u)) & 255u;                                                   \
                double acc = 0.0;                                                                                 \
                if (HAVE_B && (!MASKED || (m & 1u))) acc = fma(c.v0[u], xb, acc);                                 \
                if (!MASKED || (m & 2u)) acc = fma(c.v1[u], xa, acc);                                             \
                if (!MASKED || (m & 4u)) acc = fma(c.v2[u], xl, acc);                                             \
                xl = (c.rr[u] - acc) / c.dd[u];                                                                   \
                out[u] = xl;                                                                                      \
                if (HAVE_ACC && act && w0 + u >= 0 && w0 + u < g.a) acc_out[row_base + w0 + u] = acc;             \
            }                                                                                                     \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) s_c[buf][t][u] = out[u];                            \
            if (act) {                                                                                            \
                char *xo = reinterpret_cast<char *>(ximg) + c.slot0 * 32;                                         \
                const size_t plane_b = (size_t)c.cnt1 * 16;                                                       \
                _Pragma("unroll") for (int u = 0; u < LN_CH; u += 2) { lines_v2d o2; o2.x = out[u]; o2.y = out[u + 1]; *reinterpret_cast<lines_v2d *>(xo + (size_t)(u >> 1) * plane_b + c.vo) = o2; } \
                /* agent scope: lines on a tile's high edges are read by other workgroups while this one runs (rows */ \
                /* outside the line land in the padding) */                                                       \
                if (pub_a) { _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                    \
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(xe_a + w0 + u), (unsigned long long)__double_as_longlong(out[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } \
                if (pub_b) { _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                    \
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(xe_b + w0 + u), (unsigned long long)__double_as_longlong(out[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } \
            }                                                                                                     \
            ln_barrier();                                                                                         \
            buf ^= 1;                                                                                             \
        }
        for (int S0 = 0; S0 < nsp; S0 += K) { LN_STEP(0) LN_STEP(1) LN_STEP(2) LN_STEP(3) if constexpr (K == 5) LN_STEP(4) }
#undef LN_STEP
#undef LN_ISSUE
    } else {
        // ---- the edge wave: lane e < 16 fetches, for the tile's line (ty = 0, tz = e), the rows of line y - 1 (another
        // tile's last); lane 16 + e' for the line (ty = e', tz = 0) those of line z - 1.  A window is requested LN_KE steps
        // before the step that uses it; what is still the pending pattern then is fetched again until it is not (bounded).
        const int e = t - LINES_WG;
        const bool is_a = e < 16;
        const int ty = is_a ? 0 : e - 16, tz = is_a ? e : 0;
        const int y = by * TY + ty, z = bz * TZ + tz;
        const bool have = e < LN_EDGE && (TWO_D ? e == 0 : true) && y < g.ny && z < g.nz && (is_a ? by > 0 : (HAVE_B && bz > 0));
        // the neighbouring tile's line in the edge buffer: tile (by - 1, bz), side 0, index tz; resp. tile (by, bz - 1), side 1, index ty
        const double *x = xedge + (have ? ln_edge_off(g, is_a ? tile - 1 : tile - g.nby, is_a ? 0 : 1, is_a ? (TWO_D ? 0 : tz) : ty) : ln_edge_off(g, 0, 0, 0));
        const int rho = ln_rho(TWO_D ? 0 : (ty | (tz << 4)), TWO_D);          // of the line this lane fetches for
        const bool any_have = __any(have);               // (a tile with no neighbour below it: nothing to fetch, nothing to wait for)
        const int es = e < LN_EDGE ? e : 0;
        bool gave_up = false;
        double pf[LN_KE][LN_CH] = {};
        int pfw[LN_KE];                                  // first row of the window a request is for (clamped into the padded line)
        // Every vector-memory access of this wave is inline asm with hand-counted waits: with its polling loop in sight the
        // compiler would drain all loads in flight before each use — the requests LN_KE steps ahead included, one uncached
        // round trip per step for the whole workgroup (1.6 us per step measured).  A request is LN_CH loads; between a
        // window's request and its use lie LN_KE - 1 later requests, so vmcnt(LN_CH * (LN_KE - 1)) = 12 says it has arrived
        // (loads return in issue order).  The "+v" ties keep the compiler from touching the registers before that.
#define LN_EISSUE(I, T) { int w_ = 4 * (T) - rho; w_ = w_ < -LN_PAD ? -LN_PAD : (w_ > g.a ? g.a : w_); pfw[I] = w_;  \
            const double *src_ = x + w_;                                                                          \
            if (any_have) _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                       \
                asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(pf[I][u]) : "v"(src_ + u) : "memory"); }
#if LN_DEBUG
#define LN_DBG_SLOW if (spins_ == 1) { atomicAdd(counter + 5, 1); if (tile < 4096) atomicAdd(reinterpret_cast<unsigned long long *>(&ln_dbg[tile][3]), 1ull); } if (spins_ >= 1) atomicAdd(counter + 6, 1);
#else
#define LN_DBG_SLOW
#endif
#define LN_EPUT(I, T) {                                                                                           \
            if (any_have) asm volatile("s_waitcnt vmcnt(12)" : "+v"(pf[I][0]), "+v"(pf[I][1]), "+v"(pf[I][2]), "+v"(pf[I][3]) :: "memory"); \
            const int w_ = pfw[I];                                                                                \
            if (have && !gave_up) {                                                                               \
                const long long t0_ = wall_clock64();                                                             \
                for (int spins_ = 0;; ++spins_) {                                                                 \
                    bool ok_ = true;                                                                              \
                    _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                             \
                        ok_ = ok_ && (w_ + u < 0 || w_ + u >= g.a || (unsigned long long)__double_as_longlong(pf[I][u]) != LINES_PENDING); \
                    LN_DBG_SLOW                                                                                   \
                    if (ok_) break;                                                                               \
                    if ((spins_ & 15) == 15 && (lines_ld_agent_int(fail) != 0 || wall_clock64() - t0_ > LINES_WAIT_TICKS)) { gave_up = true; break; } \
                    __builtin_amdgcn_s_sleep(1);                                                                  \
                    /* ask again — for this window and for the LN_KE - 1 requested after it, which were asked for at about */ \
                    /* the same time and would each come back here otherwise — and wait ONCE for all of it */         \
                    _Pragma("unroll") for (int j_ = 0; j_ < LN_KE; ++j_) {                                         \
                        const double *sj_ = x + pfw[j_];                                                          \
                        _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                         \
                            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(pf[j_][u]) : "v"(sj_ + u) : "memory"); \
                    }                                                                                             \
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]), "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]), \
                                 "+v"(pf[2][0]), "+v"(pf[2][1]), "+v"(pf[2][2]), "+v"(pf[2][3]), "+v"(pf[3][0]), "+v"(pf[3][1]), "+v"(pf[3][2]), "+v"(pf[3][3]) :: "memory"); \
                }                                                                                                 \
            }                                                                                                     \
            /* rows outside the line, and lines with no neighbour: exactly 0.0 (what an absent entry multiplies) */ \
            if (e < LN_EDGE) { _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                  \
                s_c[((T) + 1) & 1][LINES_WG + es][u] = (have && w_ + u >= 0 && w_ + u < g.a) ? pf[I][u] : 0.0; } }
        static_assert(LN_CH * (LN_KE - 1) == 12, "the hand-counted wait above");
        // Pacing.  A request travels LN_KE steps ahead of its use, so this tile runs without re-polls only once it is that
        // far + a memory round trip behind its neighbour.  Started as soon as the first rows are there, it gets there by
        // failing: every early step finds the pattern and pays a round trip.  So the lanes whose lines start first wait
        // until their neighbour has the rows of g.start steps in memory — with sleeps that grow: most tiles wait for
        // hundreds of us here, and 255 waves polling flat out slow the memory of the few that run.
        {
            const bool starter = have && rho == 0;
            int rw = 4 * g.start + 3; rw = rw < g.a - 1 ? rw : g.a - 1;
            if (starter) {
                const long long t0_ = wall_clock64();
                for (int spins_ = 0;; ++spins_) {
                    if (lines_ld_agent(x + rw) != LINES_PENDING) break;
                    if ((spins_ & 7) == 7 && (lines_ld_agent_int(fail) != 0 || wall_clock64() - t0_ > LINES_WAIT_TICKS)) break;   // (the polls below give up)
                    if (spins_ < 4) __builtin_amdgcn_s_sleep(8); else if (spins_ < 16) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(100);
                }
            }
        }
        LN_EISSUE(0, 0) LN_EISSUE(1, 1) LN_EISSUE(2, 2) LN_EISSUE(3, 3)
        static_assert(LN_KE == 4, "the step loop is unrolled by hand");
        LN_EPUT(0, 0)
#if LN_DEBUG
        if (e == 0 && tile < 4096) ln_dbg[tile][1] = wall_clock64();
#endif
        LN_EISSUE(0, 4)
        ln_barrier();
        // iteration S: the rows of step S + 1 into LDS, the request for step S + 1 + LN_KE
#define LN_ESTEP(I)                                                                                               \
        if (S0 + (I) < nsp) {                                                                                     \
            const int S = S0 + (I);                                                                               \
            LN_EPUT(((I) + 1) % LN_KE, S + 1)                                                                     \
            LN_EISSUE(((I) + 1) % LN_KE, S + 1 + LN_KE)                                                           \
            ln_barrier();                                                                                         \
        }
        for (int S0 = 0; S0 < nsp; S0 += LN_KE) { LN_ESTEP(0) LN_ESTEP(1) LN_ESTEP(2) LN_ESTEP(3) }
#undef LN_ESTEP
#undef LN_EPUT
#undef LN_EISSUE
#undef LN_DBG_SLOW
#if LN_DEBUG
        if (e == 0 && tile < 4096) ln_dbg[tile][2] = wall_clock64();
#endif
        if (gave_up) {
            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fail_host) __hip_atomic_store(fail_host, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- host side
// lanes of wave w with rho <= r, r = -1 .. rho_max (index r + 1): what the step tables are made of
static void lines_rho_prefix(bool two_d, std::vector<int> (&cum)[LINES_WG / ITS_WAVE])
{
    const int rmax = ln_rho_max(two_d);
    for (int w = 0; w < LINES_WG / ITS_WAVE; ++w) {
        cum[w].assign((size_t)rmax + 2, 0);
        for (int l = 0; l < ITS_WAVE; ++l) cum[w][(size_t)ln_rho(w * ITS_WAVE + l, two_d) + 1] += 1;
        for (int r = 0; r <= rmax; ++r) cum[w][(size_t)r + 1] += cum[w][(size_t)r];
    }
}
// lanes of a wave that have a window at step S: those with 4 S - a < rho < 4 S + 4
static int lines_active(const std::vector<int> &cum, int rmax, int a, int64_t S)
{
    int64_t lo = 4 * S - a + 1, hi = 4 * S + 3;
    if (lo < 0) lo = 0;
    if (hi > rmax) hi = rmax;
    return hi >= lo ? cum[(size_t)hi + 1] - cum[(size_t)lo] : 0;
}
void lines_plan_geometry(LinesPlan &P)
{
    const bool two_d = P.b == 0;
    P.ty = two_d ? 256 : 16; P.tz = two_d ? 1 : 16; P.ch = LN_CH;
    P.nby = (P.ny + P.ty - 1) / P.ty; P.nbz = (P.nz + P.tz - 1) / P.tz;
    P.nsteps = (P.a - 1 + ln_rho_max(two_d)) / LN_CH + 1;
    // every lane has a window at (a + 3) / 4 or one more steps, and every step has one neutral slot
    std::vector<int> cum[LINES_WG / ITS_WAVE];
    lines_rho_prefix(two_d, cum);
    int64_t slots = 0;
    for (int S = 0; S < P.nsteps; ++S) {
        for (int w = 0; w < LINES_WG / ITS_WAVE; ++w) slots += lines_active(cum[w], ln_rho_max(two_d), P.a, S);
        slots += 1;
    }
    P.tslots = (int)slots;
}
// sbase[S] = slots of a tile before step S (S = 0 .. nsteps); tabs[4 S ..] = {cnt, active lanes of wave 0, of waves 0..1, of waves 0..2}
void lines_step_tables(const LinesPlan &P, int *sbase, int *tabs)
{
    const bool two_d = P.b == 0;
    std::vector<int> cum[LINES_WG / ITS_WAVE];
    lines_rho_prefix(two_d, cum);
    int base = 0;
    for (int S = 0; S < P.nsteps; ++S) {
        int per_wave[LINES_WG / ITS_WAVE];
        for (int w = 0; w < LINES_WG / ITS_WAVE; ++w) per_wave[w] = lines_active(cum[w], ln_rho_max(two_d), P.a, S);
        const int cnt = per_wave[0] + per_wave[1] + per_wave[2] + per_wave[3];
        sbase[S] = base;
        tabs[4 * S] = cnt; tabs[4 * S + 1] = per_wave[0]; tabs[4 * S + 2] = per_wave[0] + per_wave[1]; tabs[4 * S + 3] = per_wave[0] + per_wave[1] + per_wave[2];
        base += cnt + 1;
    }
    sbase[P.nsteps] = base;
}

static LinesGeom lines_geom(const LinesPlan &P)
{
    LinesGeom g; g.n = P.n; g.a = P.a; g.b = P.b; g.ny = P.ny; g.nz = P.nz; g.nby = P.nby; g.nbz = P.nbz; g.nsteps = P.nsteps; g.tslots = P.tslots;
    g.sbase = P.sbase; g.tabs = reinterpret_cast<const int4 *>(P.tabs);
    static const int start = getenv("ITS_LINES_START") ? atoi(getenv("ITS_LINES_START")) : 6;
    g.start = start < 0 ? 0 : start;
    g.mute = P.mute_first_tile ? 1 : 0;
    return g;
}
int lines_steps(const LinesPlan &P) { return P.nsteps; }

void launch_lines_build(const LinesPlan &P, const CsrDev &L, const double *D, int *flags, hipStream_t s)
{
    const int64_t blocks = (int64_t)P.nby * P.nbz * ((P.nsteps + LN_G - 1) / LN_G);
    hipLaunchKernelGGL(lines_build_kernel, dim3((unsigned)blocks), dim3(LINES_WG), 0, s, lines_geom(P), L.row_ptr, L.col, L.val, D, P.img, P.mimg, flags);
}

size_t lines_image_doubles(const LinesPlan &P) { return (size_t)P.nby * P.nbz * P.tslots * (P.b ? 4 : 3) * LN_CH; }
size_t lines_image_masks(const LinesPlan &P) { return (size_t)P.nby * P.nbz * P.tslots; }
size_t lines_edge_doubles(const LinesPlan &P) { return (size_t)P.nby * P.nbz * 2 * 16 * (size_t)(P.a + 2 * LN_PAD); }

void launch_sweep_lines(const LinesPlan &P, double *x, const double *rhs, double *acc_out, hipStream_t s)
{
    const LinesGeom g = lines_geom(P);
    const int tiles = P.nby * P.nbz;
    const unsigned blocks = (unsigned)((int64_t)tiles * ((P.nsteps + LN_G - 1) / LN_G));
    hipLaunchKernelGGL(lines_pack_kernel, dim3(blocks), dim3(LINES_WG), 0, s, g, rhs, P.rimg, reinterpret_cast<unsigned long long *>(P.xedge), P.counter);
#define ITS_LAUNCHLN(B, A) do { \
    if (P.masked) hipLaunchKernelGGL((sweep_lines_kernel<B, A, true>), dim3(tiles), dim3(LINES_THREADS), 0, s, g, g.tabs, P.img, P.mimg, P.counter, P.fail, P.fail_host, P.ximg, P.rimg, P.xedge, acc_out); \
    else hipLaunchKernelGGL((sweep_lines_kernel<B, A, false>), dim3(tiles), dim3(LINES_THREADS), 0, s, g, g.tabs, P.img, P.mimg, P.counter, P.fail, P.fail_host, P.ximg, P.rimg, P.xedge, acc_out); } while (0)
    if (P.b) { if (acc_out) ITS_LAUNCHLN(true, true); else ITS_LAUNCHLN(true, false); }
    else { if (acc_out) ITS_LAUNCHLN(false, true); else ITS_LAUNCHLN(false, false); }
#undef ITS_LAUNCHLN
    hipLaunchKernelGGL(lines_unpack_kernel, dim3(blocks), dim3(LINES_WG), 0, s, g, P.ximg, x);
    if (LN_DEBUG) {
        int c[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(c, P.counter, sizeof c, hipMemcpyDeviceToHost);
        fprintf(stderr, "lines debug: tiles %d, slow-path entries %d, re-polls %d\n", c[0], c[5], c[6]);
        (void)hipMemset(P.counter + 5, 0, 2 * sizeof(int));
#if LN_DEBUG
        static long long hd[4096][4];
        (void)hipMemcpyFromSymbol(hd, HIP_SYMBOL(ln_dbg), sizeof hd);
        long long t0 = hd[0][0];
        for (int i = 0; i < tiles && i < 4096; ++i) t0 = hd[i][0] < t0 ? hd[i][0] : t0;
        if (getenv("ITS_LN_DUMP"))
            for (int i = 0; i < tiles && i < 4096; ++i)
                if (i % P.nby == i / P.nby || i < P.nby || i % P.nby == 0 || tiles <= 64)
                    fprintf(stderr, "  tile %4d (%2d,%2d): taken %8.2f us, first rows %8.2f, done %8.2f, slow polls %lld (lanes) in %d steps\n", i, i % P.nby, i / P.nby,
                            (hd[i][0] - t0) * 0.01, (hd[i][1] - t0) * 0.01, (hd[i][2] - t0) * 0.01, hd[i][3], P.nsteps);
#endif
    }
}

}  // namespace its
