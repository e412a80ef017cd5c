// its_lines.hip — the NATURAL-ORDER triangular sweep of a stencil matrix as one launch: a wavefront pipelined over grid lines.
//
// spltsv_crs (kernels.hpp:451-473) walks the rows in order; the level schedule of its_api.hip reproduces it exactly, but a
// 3-D stencil in natural order has one level per hyperplane x + y + z — 766 launches of scattered rows at 256^3, 4.5 us
// each whatever they hold (3.5 ms per sweep; its_solve -gs, the reference's default order).  What the dependencies really
// are: with strictly lower offsets {-1, -a, -b} (a = rows per grid line, b = rows per plane) row (p, y, z) reads
// (p - 1, y, z), (p, y - 1, z) and (p, y, z - 1).  So
//   * a LANE owns a line and walks it in chunks of CH consecutive rows, the chain over the -1 entry staying in its
//     registers;
//   * a WORKGROUP owns a tile of TY x TZ lines; lane (ty, tz) runs chunk q at step q + ty + tz, one step behind both
//     neighbours it reads, whose chunk it finds in LDS (double-buffered, one barrier per step);
//   * tiles are taken in row-major order from a counter, so whatever a tile waits for has been started before it; the lines
//     on a tile's low edges read their neighbours' chunks — written by another workgroup, possibly on another XCD — from the
//     sweep's output buffer itself: it is preset to a NaN pattern no sweep produces, values are stored and polled with
//     agent-scope accesses, and a chunk is used once none of its CH values is the pattern (data and arrival flag are
//     the same word: no fence, no counter to order against).  Every poll is bounded by the wall clock; a sweep that gives
//     up says so (sticky flag) and the host reports it instead of a result.
// Per row the operations are the reference's in its order — fma over the row's entries by ascending column from 0.0, then
// (rhs - acc) / D — so x is bit-identical to the level schedule's and to spltsv_crs.
// The critical path is (a / CH + ny + nz) chunk steps instead of one launch per hyperplane.  Measured: a
// step is bound by the instructions one wave issues — a lane is alone on its SIMD: 0.66 us on a 2-D grid, 0.9 us on a
// 3-D one, of which the four dependent divisions are 0.33 — once (1) everything a step touches is stored BY STEP (below),
// (2) the loop has no edge on which the newest loads are the oldest, and (3) the waiting tiles poll with sleeps that
// grow.  A hop from tile to tile through memory costs what the skew inside a tile costs (TY - 1 steps) plus the
// distance requests travel ahead (LN_KE steps) plus a memory round trip: 25 - 30 us on 16 x 16 tiles.
// Gauss-Seidel trip in natural order, level schedule -> this: 2-D 256 x 256 0.393 -> 0.24 ms, 1000 x 1000 4.41 -> 1.00,
// 256 lines of 4096 rows 5.2 -> 1.15; 3-D 64^3 0.53 -> 0.20, 128^3 1.35 -> 0.51, 256^3 3.63 -> 1.49 (sweep 1.01 + pack 0.08 +
// unpack 0.07 + the pass that records the residual 0.26).  Taken wherever the matrix qualifies (option "sweep_lines":
// 0 = level schedule).
// Eligibility is verified row by row on the device when the image is built (lines_build_kernel): every entry of L is one
// of the three offsets, present only where the grid has that neighbour (no wrap-around), columns ascending.
#include "its_kernels.h"

#include <cstdio>
#include <cstdlib>

namespace its {

constexpr unsigned long long LINES_PENDING = 0x7ff8dead0badf11dull;    // the preset pattern of x (a quiet NaN with a payload)
constexpr long long LINES_WAIT_TICKS = 5000000;                         // 50 ms of the 100 MHz wall clock per poll
constexpr int LINES_WG = 256;
typedef double lines_v2d __attribute__((ext_vector_type(2)));
struct LinesGeom { int n, a, b, ny, nz, ty, tz, nby, nbz, start; const int *sbase; };

// ---- The order everything the sweep touches is stored in.
// A lane owns a line, and lane (ty, tz) of a tile runs chunk q at step S = q + ty + tz: at one step the lanes of a wave are
// at rows a whole line apart AND at different chunks.  Read from row-major storage (or from an image ordered by chunk,
// the first complete version) every lane touches a cache line of its own, and the texture path, which takes one line per
// cycle, is the bound: 12 loads x 64 lines x 4 waves = 1.3 us per step, what that version measured (1.0 .. 1.24 us).  So
// everything is stored BY STEP: the chunks that run at step S of a tile — the lanes with 0 <= S - ty - tz < nq, `active` —
// side by side in lane order,
//     slot(tile, S, lane) = tile * nq * 256 + base(S) + rank(S, lane)
//     base(S) = chunks of the tile's earlier steps, cnt(S) = active lanes at S, rank = active lanes below this one
// and inside a step's block one plane per pair of rows (u >> 1) of each operand, so that a wave's 16-byte loads are
// consecutive memory:
//     img [((slot0(S) * NA + k) * 2 + (u >> 1)) * cnt(S) * 2 + rank * 2 + (u & 1)]     k: [-b,] -a, -1, diagonal  (NA = 3 | 4)
//     rimg, ximg [(slot0(S) * 2 + (u >> 1) * cnt(S) + rank) * 2 + (u & 1)]             right-hand side in, x out
//     mimg[slot0(S) + rank]                                                            presence bits, a byte per row
// (slot0(S) = tile * nq * 256 + base(S)).  A lane keeps its place incrementally: from S to S + 1 the lanes with
// ty + tz = S + 1 join and those with ty + tz = S + 1 - nq leave, both counted in closed form.
__host__ __device__ __forceinline__ int ln_diagtotal(int d, int TY, int TZ)       // lanes of a tile with ty + tz == d
{
    const int lo = d - TY + 1 > 0 ? d - TY + 1 : 0, hi = TZ - 1 < d ? TZ - 1 : d;
    return hi >= lo ? hi - lo + 1 : 0;
}
template <bool TWO_D>                                                              // (TWO_D: tiles of TY x 1 lines, tz == 0)
__device__ __forceinline__ int ln_diagcount(int d, int ty, int tz, int TY)         // those of them below lane (ty, tz)
{
    if (TWO_D) return (unsigned int)d < (unsigned int)ty ? 1 : 0;
    const int lo = d - TY + 1 > 0 ? d - TY + 1 : 0, hi = tz - 1 < d ? tz - 1 : d;
    return (hi >= lo ? hi - lo + 1 : 0) + ((d >= tz && d - tz < ty) ? 1 : 0);
}
struct LnPos { int S, base, cnt, rank; };
template <bool TWO_D>
__device__ __forceinline__ void ln_pos_first(LnPos &p, int ty, int tz, int TY, int TZ)
{ p.S = 0; p.base = 0; p.cnt = ln_diagtotal(0, TY, TZ); p.rank = ln_diagcount<TWO_D>(0, ty, tz, TY); }
template <bool TWO_D>
__device__ __forceinline__ void ln_pos_next(LnPos &p, int ty, int tz, int TY, int TZ, int nq)
{
    p.base += p.cnt; ++p.S;
    p.cnt += ln_diagtotal(p.S, TY, TZ) - ln_diagtotal(p.S - nq, TY, TZ);
    p.rank += ln_diagcount<TWO_D>(p.S, ty, tz, TY) - ln_diagcount<TWO_D>(p.S - nq, ty, tz, TY);
}

// The kernels that build and (un)pack run one workgroup per (tile, LN_G consecutive steps): a lane's chunks of consecutive
// steps are consecutive rows, so in row order it touches LN_G * 32 bytes in one piece.  Rank and count come from the
// lanes' own flags.
constexpr int LN_G = 4;
struct LnWhere { int tile, S, q, ty, tz, y, z, rank, cnt; bool active, valid; size_t slot0; };
__device__ __forceinline__ void ln_where(const LinesGeom &g, int64_t bid, int t, LnWhere (&w)[LN_G], int (*s_w)[LINES_WG / ITS_WAVE])
{
    const int nq = g.a / 4, nsteps = nq + g.ty + g.tz - 2, groups = (nsteps + LN_G - 1) / LN_G;
    const int tile = (int)(bid / groups), S0 = (int)(bid % groups) * LN_G;
    const int by = tile % g.nby, bz = tile / g.nby;
    const int lane = t & (ITS_WAVE - 1), wv = t / ITS_WAVE;
    unsigned long long m[LN_G];
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        LnWhere &x = w[j];
        x.tile = tile; x.S = S0 + j;
        x.ty = t % g.ty; x.tz = t / g.ty;
        x.q = x.S - x.ty - x.tz;
        x.y = by * g.ty + x.ty; x.z = bz * g.tz + x.tz;
        x.active = x.S < nsteps && x.tz < g.tz && x.q >= 0 && x.q < nq;
        x.valid = x.active && x.y < g.ny && x.z < g.nz;
        m[j] = __ballot(x.active);
        if (lane == 0) s_w[j][wv] = __popcll(m[j]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        int below = 0, cnt = 0;
#pragma unroll
        for (int i = 0; i < LINES_WG / ITS_WAVE; ++i) { if (i < wv) below += s_w[j][i]; cnt += s_w[j][i]; }
        w[j].rank = below + __popcll(m[j] & ((1ull << lane) - 1ull));
        w[j].cnt = cnt;
        w[j].slot0 = (size_t)tile * nq * LINES_WG + (w[j].S < nsteps ? g.sbase[w[j].S] : 0);
    }
}

// The (<= 3) entries of each row into fixed slots — offset -b, -a, -1, which is ascending column order — flags[0] |= 1 when
// a row is not of that form.  Lanes past the grid's edge (ragged tiles): no entries, diagonal 1.
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
        if (!w.active) continue;
        double *o = img + w.slot0 * NA * 4 + (size_t)w.rank * 2;
        unsigned int mword = 0;
        for (int u = 0; u < 4; ++u) {
            double v[3] = {0.0, 0.0, 0.0}, dg = 1.0;
            int m = 0;
            if (w.valid) {
                const int p = w.q * 4 + u;
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
                // a row without an entry where the grid has the neighbour: the sweep must skip it by its presence bit
                // (flags[0] & 2); without such rows absent entries are stored as 0.0 and meet operands that are finite
                const int full = (p >= 1 ? 4 : 0) | (w.y >= 1 ? 2 : 0) | ((g.b > 0 && w.z >= 1) ? 1 : 0);
                holes |= m != full;
            }
            const size_t plane = (size_t)w.cnt * 2, e = (size_t)(u >> 1) * plane + (u & 1);
            int k = 0;
            if (g.b > 0) o[(size_t)(k++) * 2 * plane + e] = v[0];
            o[(size_t)(k++) * 2 * plane + e] = v[1];
            o[(size_t)(k++) * 2 * plane + e] = v[2];
            o[(size_t)k * 2 * plane + e] = dg;
            mword |= (unsigned int)m << (8 * u);
        }
        mimg[w.slot0 + w.rank] = mword;
    }
    if (bad) atomicOr(flags, 1);
    if (holes) atomicOr(flags, 2);
}

// In front of every sweep: the right-hand side into that order (rimg); in the sweep's output buffer (ximg, same order) the
// pending pattern wherever another workgroup will poll — the lines on a tile's high edges; tile counter := 0.  Behind it:
// x out of ximg into row order.  Both are streaming passes.
__global__ __launch_bounds__(LINES_WG) void lines_pack_kernel(
    LinesGeom g, const double *__restrict__ rhs, double *__restrict__ rimg, unsigned long long *__restrict__ ximg, int *counter)
{
    __shared__ int s_w[LN_G][LINES_WG / ITS_WAVE];
    const int t = threadIdx.x;
    LnWhere ws[LN_G];
    ln_where(g, blockIdx.x, t, ws, s_w);
    if (blockIdx.x == 0 && t == 0) *counter = 0;
    lines_v2d r01[LN_G], r23[LN_G];
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        const LnWhere &w = ws[j];
        r01[j].x = 0.0; r01[j].y = 0.0; r23[j] = r01[j];
        if (w.valid) {
            const int64_t r0 = ((int64_t)w.z * g.ny + w.y) * g.a + (int64_t)w.q * 4;
            r01[j] = *reinterpret_cast<const lines_v2d *>(rhs + r0);          // (plain loads: the four chunks of a lane share a cache line)
            r23[j] = *reinterpret_cast<const lines_v2d *>(rhs + r0 + 2);
        }
    }
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        const LnWhere &w = ws[j];
        if (!w.active) continue;
        const size_t o = (w.slot0 * 2 + w.rank) * 2, plane = (size_t)w.cnt * 2;
        *reinterpret_cast<lines_v2d *>(rimg + o) = r01[j]; *reinterpret_cast<lines_v2d *>(rimg + o + plane) = r23[j];
        if (w.valid && ((w.ty == g.ty - 1 && w.y + 1 < g.ny) || (g.b > 0 && w.tz == g.tz - 1 && w.z + 1 < g.nz))) {   // the sweep's `pub` lanes
            ximg[o] = LINES_PENDING; ximg[o + 1] = LINES_PENDING; ximg[o + plane] = LINES_PENDING; ximg[o + plane + 1] = LINES_PENDING;
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
        if (!w.valid) continue;
        const size_t o = (w.slot0 * 2 + w.rank) * 2, plane = (size_t)w.cnt * 2;
        r01[j] = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(ximg + o));
        r23[j] = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(ximg + o + plane));
    }
#pragma unroll
    for (int j = 0; j < LN_G; ++j) {
        const LnWhere &w = ws[j];
        if (!w.valid) continue;
        const int64_t r0 = ((int64_t)w.z * g.ny + w.y) * g.a + (int64_t)w.q * 4;
        *reinterpret_cast<lines_v2d *>(x + r0) = r01[j];
        *reinterpret_cast<lines_v2d *>(x + r0 + 2) = r23[j];
    }
}

#ifndef LN_DEBUG
#define LN_DEBUG 0                  // 1: count the edge wave's slow-path entries and re-polls (printed after every sweep)
#endif
#if LN_DEBUG
__device__ long long ln_dbg[4096][4];      // per tile: taken, first neighbour chunk in hand, done, slow-path entries
#endif
constexpr int LN_CH = 4;            // rows per chunk (one step of a lane)
constexpr int LN_K = 5;             // operand sets in flight per lane: a step's operands are requested LN_K - 1 steps early
constexpr int LN_KE = 4;            // the same for the neighbouring tiles' chunks (edge wave)
constexpr int LN_EDGE = 32;         // lines on a tile's low edges: <= 16 along z (ty == 0) + <= 16 along y (tz == 0)
constexpr int LINES_THREADS = LINES_WG + ITS_WAVE;   // four waves of lanes that own lines + the edge wave

// One chunk's operands: the entries of its rows (slots -b, -a, -1), diagonal, right-hand side, presence bits.  None of
// them depends on x: they are requested LN_K - 1 steps ahead, by UNCONDITIONAL loads at clamped addresses — straight-line
// code, so that the compiler's wait counts stay exact (with loads under divergent branches or next to a polling loop it
// drains everything in flight every step: a memory round trip per step, 1.2 .. 11 us in the first versions).
struct LnSet { double v0[LN_CH], v1[LN_CH], v2[LN_CH], dd[LN_CH], rr[LN_CH]; unsigned int m, vo; size_t slot0; int cnt; };

// Addresses as (uniform base) + (32-bit lane offset): the bases live in scalar registers and the loads take them as such —
// a lane in this kernel is alone on its SIMD, every vector instruction it does not issue is time off the step.
template <bool HAVE_B, bool MASKED>
__device__ __forceinline__ void ln_load(LnSet &c, size_t tile_slot0, const LnPos &p, const double *__restrict__ img, const unsigned int *__restrict__ mimg,
                                        const double *__restrict__ rimg)
{
    constexpr int NA = HAVE_B ? 4 : 3;
    const int r = p.rank < p.cnt ? p.rank : p.cnt - 1;           // a lane that is not active this step reads an active lane's slot (unused)
    const size_t slot0 = tile_slot0 + (size_t)p.base, plane_b = (size_t)p.cnt * 16;        // (a plane: cnt pairs of doubles)
    const char *ib = reinterpret_cast<const char *>(img) + slot0 * (NA * 32);
    const char *rb = reinterpret_cast<const char *>(rimg) + slot0 * 32;
    const unsigned int vo = (unsigned int)r * 16u;
    c.slot0 = slot0; c.cnt = p.cnt; c.vo = vo;
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

// MASKED: rows may lack an entry where the grid has the neighbour (presence bits decide, as everywhere before this
// variant existed).  Otherwise the only absent entries are those of rows on the grid's low faces: stored as 0.0 they
// multiply an operand that is made finite for them — zeros from the edge wave for the lines at y = 0 / z = 0, xprev reset at
// the head of a line — and fma(0.0, finite, acc) == acc bit for bit (acc is never -0.0: it starts from +0.0), which
// saves the 15 selects and tests per row-triple of a step.
template <bool HAVE_B, bool HAVE_ACC, bool MASKED>
__global__ __launch_bounds__(LINES_THREADS) void sweep_lines_kernel(
    LinesGeom g, const double *__restrict__ img, const unsigned int *__restrict__ mimg, int *counter, int *fail,
    unsigned long long *fail_host, double *ximg, const double *__restrict__ rimg, double *acc_out)
{
    // rows 0 .. 255: every line's last chunk (its neighbours in the tile read it one step later); rows 256 ..: the neighbouring
    // tiles' chunks for the lines on the low edges (edge wave) — one array, so that a lane's two neighbours are two row
    // numbers fixed for the whole sweep
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
    const int nq = g.a / LN_CH, nsteps = nq + g.ty + g.tz - 2;
    // The lanes' loop runs whole rounds of LN_K steps without a branch in between (steps past the last one have no active
    // lane): with `if (S < nsteps)` around every step the loop has an edge from the first step straight back to its head,
    // and on that edge the step's requests are the youngest loads — the compiler then drains everything in flight at
    // the head of every round (vmcnt(0): a memory round trip per 5 steps, 0.3 us per step measured).
    const int nsp = (nsteps + LN_K - 1) / LN_K * LN_K;
    if (t < LINES_WG) {
        // ---- a lane that owns a line
        const int ty = t % g.ty, tz = t / g.ty;
        const int y = by * g.ty + ty, z = bz * g.tz + tz;
        const bool valid = tz < g.tz && y < g.ny && z < g.nz;
        const int64_t row_base = valid ? ((int64_t)z * g.ny + y) * g.a : 0;
        const int skew = ty + tz;
        const int ia = ty > 0 ? t - 1 : LINES_WG + tz;                       // line y - 1: in the tile, or fetched by the edge wave
        const int ib = tz > 0 ? t - g.ty : LINES_WG + 16 + ty;               // line z - 1
        const bool pub = valid && ((ty == g.ty - 1 && y + 1 < g.ny) || (HAVE_B && tz == g.tz - 1 && z + 1 < g.nz));
        double xprev = 0.0;
        LnSet set[LN_K];
        const size_t tile_slot0 = (size_t)tile * nq * LINES_WG;
        LnPos pi;                                        // the place of the step whose operands are requested next
        ln_pos_first<!HAVE_B>(pi, ty, tz, g.ty, g.tz);
#define LN_ISSUE(I, T) { ln_load<HAVE_B, MASKED>(set[I], tile_slot0, pi, img, mimg, rimg); if (pi.S + 1 < nsteps) ln_pos_next<!HAVE_B>(pi, ty, tz, g.ty, g.tz, nq); }
        LN_ISSUE(0, 0) LN_ISSUE(1, 1) LN_ISSUE(2, 2) LN_ISSUE(3, 3)
        static_assert(LN_K == 5, "the step loop is unrolled by hand");
        ln_barrier();                                    // (the edge wave has written the chunks of step 0)
        int buf = 0;
#define LN_STEP(I)                                                                                                \
        {                                                                                                         \
            const int S = S0 + (I);                                                                               \
            LN_ISSUE(((I) + LN_K - 1) % LN_K, S + LN_K - 1)                                                       \
            const LnSet &c = set[I];                                                                              \
            const int q = S - skew;                                                                               \
            const bool act = valid && q >= 0 && q < nq;                                                           \
            const int64_t r0 = row_base + (int64_t)(q < 0 ? 0 : (q >= nq ? nq - 1 : q)) * LN_CH;                  \
            double xa[LN_CH], xb[LN_CH];                                                                          \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) {                                                   \
                xa[u] = s_c[buf ^ 1][ia][u];                                                                      \
                xb[u] = HAVE_B ? s_c[buf ^ 1][ib][u] : 0.0;                                                       \
            }                                                                                                     \
            double out[LN_CH];                                                                                    \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) {                                                   \
                /* the row's chain in CRS order: -b, -a, -1 (ascending columns), from 0.0 */                      \
                const unsigned int m = (c.m >> (8 * u)) & 255u;                                                   \
                double acc = 0.0;                                                                                 \
                if (HAVE_B && (!MASKED || (m & 1u))) acc = fma(c.v0[u], xb[u], acc);                              \
                if (!MASKED || (m & 2u)) acc = fma(c.v1[u], xa[u], acc);                                          \
                if (!MASKED || (m & 4u)) acc = fma(c.v2[u], xprev, acc);                                          \
                xprev = (c.rr[u] - acc) / c.dd[u];                                                                \
                out[u] = xprev;                                                                                   \
                if (HAVE_ACC && act) acc_out[r0 + u] = acc;                                                       \
            }                                                                                                     \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) s_c[buf][t][u] = out[u];                            \
            if (!MASKED && (q < 0 || q >= nq)) xprev = 0.0;        /* (a line's first row has no -1 entry: 0.0 times this) */ \
            if (act) {                                                                                            \
                char *xo = reinterpret_cast<char *>(ximg) + c.slot0 * 32;                                         \
                const size_t plane_b = (size_t)c.cnt * 16;                                                        \
                if (pub) {                                                                                        \
                    /* agent scope: lines on a tile's high edges are read by other workgroups while this one runs */ \
                    _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                             \
                        __hip_atomic_store(reinterpret_cast<unsigned long long *>(xo + ((size_t)(u >> 1) * plane_b + (u & 1) * 8) + c.vo), \
                                           (unsigned long long)__double_as_longlong(out[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                } else {                                                                                          \
                    _Pragma("unroll") for (int u = 0; u < LN_CH; u += 2) { lines_v2d o2; o2.x = out[u]; o2.y = out[u + 1]; *reinterpret_cast<lines_v2d *>(xo + (size_t)(u >> 1) * plane_b + c.vo) = o2; } \
                }                                                                                                 \
            }                                                                                                     \
            ln_barrier();                                                                                         \
            buf ^= 1;                                                                                             \
        }
        for (int S0 = 0; S0 < nsp; S0 += LN_K) { LN_STEP(0) LN_STEP(1) LN_STEP(2) LN_STEP(3) LN_STEP(4) }
#undef LN_STEP
#undef LN_ISSUE
    } else {
        // ---- the edge wave: lane e < 16 fetches, for the tile's line (ty = 0, tz = e), the chunks of line y - 1 (another
        // tile's); lane 16 + e' for the line (ty = e', tz = 0) those of line z - 1.  A chunk is requested LN_KE steps before
        // the step that uses it; what is still the pending pattern then is fetched again until it is not (bounded).
        const int e = t - LINES_WG;
        const bool is_a = e < 16;
        const int ty = is_a ? 0 : e - 16, tz = is_a ? e : 0;
        const int y = by * g.ty + ty, z = bz * g.tz + tz;
        const bool have = e < LN_EDGE && ty < g.ty && tz < g.tz && y < g.ny && z < g.nz && (is_a ? y > 0 : (HAVE_B && z > 0));
        // the neighbouring tile's line in ximg: tile (by - 1, bz), lane (TY - 1, tz), resp. tile (by, bz - 1), lane (ty, TZ - 1)
        const int ntile = have ? (is_a ? tile - 1 : tile - g.nby) : 0;
        const int nty = have ? (is_a ? g.ty - 1 : ty) : 0, ntz = have ? (is_a ? tz : g.tz - 1) : 0;
        const double *x = ximg;
        const size_t ntile_slot0 = (size_t)ntile * nq * LINES_WG;
        LnPos pe;                                        // that line's place at the step it runs the chunk requested next
        ln_pos_first<!HAVE_B>(pe, nty, ntz, g.ty, g.tz);
        for (int i = 0; i < nty + ntz; ++i) ln_pos_next<!HAVE_B>(pe, nty, ntz, g.ty, g.tz, nq);      // chunk 0 runs at step nty + ntz
        int qcur = 0;
        unsigned int pfo[LN_KE]; int pfc[LN_KE];
        const bool any_have = __any(have);               // (a tile with no neighbour below it: nothing to fetch, nothing to wait for)
        const int skew = ty + tz;
        const int es = e < LN_EDGE ? e : 0;
        bool gave_up = false;
        double pf[LN_KE][LN_CH] = {};
        // Every vector-memory access of this wave is inline asm with hand-counted waits: with its polling loop in sight the
        // compiler would drain all loads in flight before each use — the requests LN_KE steps ahead included, one uncached
        // round trip per step for the whole workgroup (1.6 us per step measured).  A request is LN_CH loads; between a
        // chunk's request and its use lie LN_KE - 1 later requests, so vmcnt(LN_CH * (LN_KE - 1)) = 12 says it has arrived
        // (loads return in issue order).  The "+v" ties keep the compiler from touching the registers before that.
#define LN_EISSUE(I, T) { int q_ = (T) - skew; q_ = q_ < 0 ? 0 : (q_ >= nq ? nq - 1 : q_);                          \
            if (q_ > qcur) { ln_pos_next<!HAVE_B>(pe, nty, ntz, g.ty, g.tz, nq); qcur = q_; }       /* (by one at most per step) */ \
            pfo[I] = (unsigned int)((ntile_slot0 + (size_t)pe.base) * 2 + pe.rank); pfc[I] = pe.cnt;              \
            const double *src_ = x + (size_t)pfo[I] * 2;                                                          \
            if (any_have) _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                       \
                asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(pf[I][u]) : "v"(src_ + (size_t)(u >> 1) * pfc[I] * 2 + (u & 1)) : "memory"); }
#if LN_DEBUG
#define LN_DBG_SLOW if (spins_ == 1 && tile < 4096) atomicAdd(reinterpret_cast<unsigned long long *>(&ln_dbg[tile][3]), 1ull);
#else
#define LN_DBG_SLOW
#endif
#define LN_EPUT(I, T) { const int q_ = (T) - skew;                                                                \
            if (any_have) asm volatile("s_waitcnt vmcnt(12)" : "+v"(pf[I][0]), "+v"(pf[I][1]), "+v"(pf[I][2]), "+v"(pf[I][3]) :: "memory"); \
            if (have && q_ >= 0 && q_ < nq && !gave_up) {                                                         \
                const long long t0_ = wall_clock64();                                                             \
                for (int spins_ = 0;; ++spins_) {                                                                 \
                    bool ok_ = true;                                                                              \
                    _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) ok_ = ok_ && (unsigned long long)__double_as_longlong(pf[I][u]) != LINES_PENDING; \
                    if (LN_DEBUG && spins_ == 1) atomicAdd(counter + 5, 1);                                       \
                    LN_DBG_SLOW                                                                                   \
                    if (LN_DEBUG && spins_ >= 1) atomicAdd(counter + 6, 1);                                       \
                    if (ok_) break;                                                                               \
                    if ((spins_ & 15) == 15 && (lines_ld_agent_int(fail) != 0 || wall_clock64() - t0_ > LINES_WAIT_TICKS)) { gave_up = true; break; } \
                    __builtin_amdgcn_s_sleep(1);                                                                  \
                    /* ask again — for this chunk and for the LN_KE - 1 requested after it, which were asked for at about */ \
                    /* the same time and would each come back here otherwise — and wait ONCE for all of it */         \
                    _Pragma("unroll") for (int j_ = 0; j_ < LN_KE; ++j_) {                                         \
                        const double *sj_ = x + (size_t)pfo[j_] * 2;                                              \
                        _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                         \
                            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(pf[j_][u]) : "v"(sj_ + (size_t)(u >> 1) * pfc[j_] * 2 + (u & 1)) : "memory"); \
                    }                                                                                             \
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]), "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]), \
                                 "+v"(pf[2][0]), "+v"(pf[2][1]), "+v"(pf[2][2]), "+v"(pf[2][3]), "+v"(pf[3][0]), "+v"(pf[3][1]), "+v"(pf[3][2]), "+v"(pf[3][3]) :: "memory"); \
                }                                                                                                 \
            }                                                                                                     \
            if (e < LN_EDGE) { _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) s_c[((T) + 1) & 1][LINES_WG + es][u] = have ? pf[I][u] : 0.0; } }
        static_assert(LN_CH * (LN_KE - 1) == 12, "the hand-counted wait above");
        // Pacing.  A request travels LN_KE steps ahead of its use, so this tile runs without re-polls only once it is
        // (TY - 1) + LN_KE steps + a memory round trip behind its neighbour.  Started as soon as chunk 0 is there, it gets
        // there by failing: every early step finds the pattern and pays a round trip (measured: 5 or more per tile, 30 us
        // per tile-to-tile hop where 15 steps are 14).  So the two lanes whose lines start at step 0 first wait until their
        // neighbour has chunk g.start (LN_KE + 2; env ITS_LINES_START) in memory — with sleeps that grow: most tiles wait for hundreds of us here, and 255
        // waves polling flat out slow the memory of the few that run.
        {
            const bool starter = have && skew == 0;
            LnPos pw = pe;
            const int qw = g.start < nq - 1 ? g.start : nq - 1;
            for (int i = 0; i < qw; ++i) ln_pos_next<!HAVE_B>(pw, nty, ntz, g.ty, g.tz, nq);
            const double *sw_ = x + ((ntile_slot0 + (size_t)pw.base) * 2 + pw.rank) * 2 + (size_t)pw.cnt * 2 + 1;    // the chunk's last row
            if (starter) {
                const long long t0_ = wall_clock64();
                for (int spins_ = 0;; ++spins_) {
                    if (lines_ld_agent(sw_) != LINES_PENDING) break;
                    if ((spins_ & 7) == 7 && (lines_ld_agent_int(fail) != 0 || wall_clock64() - t0_ > LINES_WAIT_TICKS)) break;   // (the polls below give up)
                    if (spins_ < 4) __builtin_amdgcn_s_sleep(8); else if (spins_ < 16) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(100);
                }
            }
        }
        LN_EISSUE(0, 0) LN_EISSUE(1, 1) LN_EISSUE(2, 2) LN_EISSUE(3, 3)
        static_assert(LN_KE == 4, "the step loop is unrolled by hand");
        LN_EPUT(0, 0)
#if LN_DEBUG
        if (e == 0 && tile < 4096) { ln_dbg[tile][1] = wall_clock64(); }
#endif
        LN_EISSUE(0, 4)
        ln_barrier();
        // iteration S: the chunk of step S + 1 into LDS, the request for step S + 1 + LN_KE
#define LN_ESTEP(I)                                                                                               \
        if (S0 + (I) < nsp) {                                                                                  \
            const int S = S0 + (I);                                                                               \
            LN_EPUT(((I) + 1) % LN_KE, S + 1)                                                                     \
            LN_EISSUE(((I) + 1) % LN_KE, S + 1 + LN_KE)                                                           \
            ln_barrier();                                                                                         \
        }
        for (int S0 = 0; S0 < nsp; S0 += LN_KE) { LN_ESTEP(0) LN_ESTEP(1) LN_ESTEP(2) LN_ESTEP(3) }
#undef LN_ESTEP
#undef LN_EPUT
#undef LN_EISSUE
#if LN_DEBUG
        if (e == 0 && tile < 4096) { ln_dbg[tile][2] = wall_clock64(); }
#endif
        if (gave_up) {
            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fail_host) __hip_atomic_store(fail_host, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static LinesGeom lines_geom(const LinesPlan &P)
{
    LinesGeom g; g.n = P.n; g.a = P.a; g.b = P.b; g.ny = P.ny; g.nz = P.nz; g.ty = P.ty; g.tz = P.tz; g.nby = P.nby; g.nbz = P.nbz; g.sbase = P.sbase;
    static const int start = getenv("ITS_LINES_START") ? atoi(getenv("ITS_LINES_START")) : 6;
    g.start = start < 0 ? 0 : start;
    return g;
}
int lines_steps(const LinesPlan &P) { return P.a / LN_CH + P.ty + P.tz - 2; }
// base(S) of the layout above, S = 0 .. steps (the last entry: chunks per tile)
void lines_step_bases(const LinesPlan &P, int *sbase)
{
    const int nq = P.a / LN_CH, nsteps = lines_steps(P);
    int cnt = ln_diagtotal(0, P.ty, P.tz), base = 0;
    for (int S = 0; S <= nsteps; ++S) {
        sbase[S] = base;
        base += cnt;
        cnt += ln_diagtotal(S + 1, P.ty, P.tz) - ln_diagtotal(S + 1 - nq, P.ty, P.tz);
    }
}

void launch_lines_build(const LinesPlan &P, const CsrDev &L, const double *D, int *flags, hipStream_t s)
{
    const int64_t blocks = (int64_t)P.nby * P.nbz * ((lines_steps(P) + LN_G - 1) / LN_G);
    hipLaunchKernelGGL(lines_build_kernel, dim3((unsigned)blocks), dim3(LINES_WG), 0, s, lines_geom(P), L.row_ptr, L.col, L.val, D, P.img, P.mimg, flags);
}

size_t lines_image_doubles(const LinesPlan &P) { return (size_t)P.nby * P.nbz * (P.a / LN_CH) * (P.b ? 4 : 3) * (LINES_WG * LN_CH); }
size_t lines_image_masks(const LinesPlan &P) { return (size_t)P.nby * P.nbz * (P.a / LN_CH) * LINES_WG; }

void launch_sweep_lines(const LinesPlan &P, double *x, const double *rhs, double *acc_out, hipStream_t s)
{
    const LinesGeom g = lines_geom(P);
    const int tiles = P.nby * P.nbz;
    const unsigned blocks = (unsigned)((int64_t)tiles * ((lines_steps(P) + LN_G - 1) / LN_G));
    hipLaunchKernelGGL(lines_pack_kernel, dim3(blocks), dim3(LINES_WG), 0, s, g, rhs, P.rimg, reinterpret_cast<unsigned long long *>(P.ximg), P.counter);
#define ITS_LAUNCHLN(B, A) do { \
    if (P.masked) hipLaunchKernelGGL((sweep_lines_kernel<B, A, true>), dim3(tiles), dim3(LINES_THREADS), 0, s, g, P.img, P.mimg, P.counter, P.fail, P.fail_host, P.ximg, P.rimg, acc_out); \
    else hipLaunchKernelGGL((sweep_lines_kernel<B, A, false>), dim3(tiles), dim3(LINES_THREADS), 0, s, g, P.img, P.mimg, P.counter, P.fail, P.fail_host, P.ximg, P.rimg, acc_out); } while (0)
    if (P.b) { if (acc_out) ITS_LAUNCHLN(true, true); else ITS_LAUNCHLN(true, false); }
    else { if (acc_out) ITS_LAUNCHLN(false, true); else ITS_LAUNCHLN(false, false); }
#undef ITS_LAUNCHLN
    hipLaunchKernelGGL(lines_unpack_kernel, dim3(blocks), dim3(LINES_WG), 0, s, g, P.ximg, x);
    if (LN_DEBUG) {
        int c[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(c, P.counter, sizeof c, hipMemcpyDeviceToHost);
        fprintf(stderr, "lines debug: tiles %d, slow-path entries %d, re-polls %d\n", c[0], c[5], c[6]);
#if LN_DEBUG
        static long long hd[4096][4];
        (void)hipMemcpyFromSymbol(hd, HIP_SYMBOL(ln_dbg), sizeof hd);
        long long t0 = hd[0][0];
        for (int i = 0; i < tiles && i < 4096; ++i) t0 = hd[i][0] < t0 ? hd[i][0] : t0;
        if (getenv("ITS_LN_DUMP"))
            for (int i = 0; i < tiles && i < 4096; ++i)
                if (i % P.nby == i / P.nby || i < P.nby || tiles <= 64)
                    fprintf(stderr, "  tile %4d (%2d,%2d): taken %8.2f us, first chunk %8.2f, done %8.2f, slow polls %lld of %d steps\n", i, i % P.nby, i / P.nby,
                            (hd[i][0] - t0) * 0.01, (hd[i][1] - t0) * 0.01, (hd[i][2] - t0) * 0.01, hd[i][3], lines_steps(P));
#endif
        (void)hipMemset(P.counter + 5, 0, 2 * sizeof(int));
    }
}

}  // namespace its
