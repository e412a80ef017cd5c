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
//     on a tile's low edges read their neighbours' chunks — written by another workgroup, possibly on another XCD — from x
//     itself: x is preset to a NaN pattern no sweep produces, values are stored and polled with agent-scope accesses, and a
//     chunk is used once none of its CH values is the pattern (data and arrival flag are the same word: no fence, no
//     counter to order against).  Every poll is bounded by the wall clock; a sweep that gives up says so (sticky flag) and
//     the host reports it instead of a result.
// Per row the operations are the reference's in its order — fma over the row's entries by ascending column from 0.0, then
// (rhs - acc) / D — so x is bit-identical to the level schedule's and to spltsv_crs.
// The critical path is (a / CH + ny + nz) chunk steps instead of one launch per hyperplane.  Measured: one
// step costs 0.33 us of arithmetic (four dependent divisions) once nothing waits for memory, 1.0 us with the right-hand
// side read and x written a line apart from lane to lane (the two streams that cannot be stored in consumption order),
// and more the longer the lines are (the lanes' rows then lie in different pages); with tiles handing chunks to each
// other through memory another 0.3 .. 0.5 us.  Net: faster than the level schedule where one or two tiles hold the grid
// (2-D 256 x 256: 0.393 -> 0.328 ms per Gauss-Seidel trip; 256 lines of 4096 rows: 5.19 -> 4.55), slower beyond (256^3:
// 3.63 -> 3.94 ms) — its_solver_setup takes it by default only for the former (option "sweep_lines": 0 off, 1 where it
// pays, 2 wherever the matrix qualifies).
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
struct LinesGeom { int n, a, b, ny, nz, ty, tz, nby, nbz; };
typedef LinesGeom LinesGeomB;

// The image the sweep streams.  A lane owns a line, so at one step the lanes of a wave want rows that lie a whole line
// apart: read from the CRS arrays (or any row-major copy) every lane touches a cache line of its own, and the texture
// path, which takes one line per cycle, becomes the bound (1.2 us of the first versions' 1.5 us per step were operand
// loads, measured by leaving them out).  So the operands are stored in the order they are consumed:
//     img[((tile * nq + q) * NA + k) * 256 * CH + lane * CH + u]      k: 0 .. NA-1 = [-b,] -a, -1, diagonal
//     mimg[(tile * nq + q) * 256 + lane]                              presence bits of the chunk's CH rows, a byte each
// — the 256 lanes of a tile side by side, a wave's load of one operand of one chunk 2 KB of consecutive memory.
// One lane per (tile, chunk, lane): the (<= 3) entries of each of its rows into fixed slots — offset -b, -a, -1, which is
// ascending column order — flags[0] |= 1 when a row is not of that form.
__global__ __launch_bounds__(LINES_WG) void lines_build_kernel(
    LinesGeomB g, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ D,
    double *__restrict__ img, unsigned int *__restrict__ mimg, int *__restrict__ flags)
{
    const int t = threadIdx.x;
    const int64_t tq = blockIdx.x;                               // tile * nq + q
    const int nq = g.a / 4;
    const int tile = (int)(tq / nq), q = (int)(tq % nq);
    const int by = tile % g.nby, bz = tile / g.nby;
    const int ty = t % g.ty, tz = t / g.ty;
    const int y = by * g.ty + ty, z = bz * g.tz + tz;
    const bool valid = tz < g.tz && y < g.ny && z < g.nz;
    const int NA = g.b > 0 ? 4 : 3;
    double *o = img + (size_t)tq * NA * (LINES_WG * 4) + (size_t)t * 4;
    unsigned int mword = 0;
    bool bad = false;
    for (int u = 0; u < 4; ++u) {
        double v[3] = {0.0, 0.0, 0.0}, dg = 1.0;
        int m = 0;
        if (valid) {
            const int p = q * 4 + u;
            const int r = (z * g.ny + y) * g.a + p;
            dg = D[r];
            int prev = -1;
            for (int j = rp[r]; j < rp[r + 1]; ++j) {
                const int c = col[j], off = r - c;
                bad |= c <= prev; prev = c;
                int slot = -1;
                if (off == 1 && p >= 1) slot = 2;
                else if (off == g.a && y >= 1) slot = 1;
                else if (g.b > 0 && off == g.b) slot = 0;
                if (slot < 0 || (m >> slot & 1)) { bad = true; continue; }
                m |= 1 << slot; v[slot] = val[j];
            }
        }
        int k = 0;
        if (g.b > 0) o[(size_t)(k++) * (LINES_WG * 4) + u] = v[0];
        o[(size_t)(k++) * (LINES_WG * 4) + u] = v[1];
        o[(size_t)(k++) * (LINES_WG * 4) + u] = v[2];
        o[(size_t)k * (LINES_WG * 4) + u] = dg;
        mword |= (unsigned int)m << (8 * u);
    }
    mimg[(size_t)tq * LINES_WG + t] = mword;
    if (bad) atomicOr(flags, 1);
}

// x := the pending pattern, tile counter := 0 (one launch in front of every sweep)
__global__ __launch_bounds__(ITS_BLOCK) void lines_prepare_kernel(unsigned long long *x, int64_t n, int *counter)
{
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    for (int64_t i = tid; i < n; i += nth) x[i] = LINES_PENDING;
    if (tid == 0) *counter = 0;
}

#ifndef LN_DEBUG
#define LN_DEBUG 0                  // 1: count the edge wave's slow-path entries and re-polls (printed after every sweep)
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
struct LnSet { double v0[LN_CH], v1[LN_CH], v2[LN_CH], dd[LN_CH], rr[LN_CH]; unsigned int m; };

template <bool HAVE_B>
__device__ __forceinline__ void ln_load(LnSet &c, int64_t tq, int t, int64_t r0, const double *__restrict__ img, const unsigned int *__restrict__ mimg,
                                        const double *rhs)
{
    constexpr int NA = HAVE_B ? 4 : 3;
    const double *o = img + (size_t)tq * NA * (LINES_WG * LN_CH) + (size_t)t * LN_CH;
#pragma unroll
    for (int u = 0; u < LN_CH; u += 2) {
        int k = 0;
        c.v0[u] = 0.0; c.v0[u + 1] = 0.0;
        if (HAVE_B) { const lines_v2d q0 = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(o + (size_t)(k++) * (LINES_WG * LN_CH) + u)); c.v0[u] = q0.x; c.v0[u + 1] = q0.y; }
        const lines_v2d q1 = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(o + (size_t)(k++) * (LINES_WG * LN_CH) + u));
        const lines_v2d q2 = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(o + (size_t)(k++) * (LINES_WG * LN_CH) + u));
        const lines_v2d qd = __builtin_nontemporal_load(reinterpret_cast<const lines_v2d *>(o + (size_t)k * (LINES_WG * LN_CH) + u));
        const lines_v2d qr = *reinterpret_cast<const lines_v2d *>(rhs + r0 + u);     // (a line apart from lane to lane: the one operand that is not in the image)
        c.v1[u] = q1.x; c.v1[u + 1] = q1.y; c.v2[u] = q2.x; c.v2[u + 1] = q2.y;
        c.dd[u] = qd.x; c.dd[u + 1] = qd.y; c.rr[u] = qr.x; c.rr[u + 1] = qr.y;
    }
    c.m = __builtin_nontemporal_load(mimg + (size_t)tq * LINES_WG + t);
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

template <bool HAVE_B>
__global__ __launch_bounds__(LINES_THREADS) void sweep_lines_kernel(
    LinesGeom g, const double *__restrict__ img, const unsigned int *__restrict__ mimg, int *counter, int *fail,
    unsigned long long *fail_host, double *x, const double *rhs, double *acc_out)
{
    __shared__ double s_x[2][LINES_WG][LN_CH + 1];       // every line's last chunk (its neighbours in the tile read it one step later)
    __shared__ double s_e[2][LN_EDGE][LN_CH];            // the neighbouring tiles' chunks for the lines on the low edges
    __shared__ int s_tile;
    const int t = threadIdx.x;
    if (t == 0) s_tile = atomicAdd(counter, 1);
    __syncthreads();
    const int tile = s_tile;
    const int by = tile % g.nby, bz = tile / g.nby;
    const int nq = g.a / LN_CH, nsteps = nq + g.ty + g.tz - 2;
    if (t < LINES_WG) {
        // ---- a lane that owns a line
        const int ty = t % g.ty, tz = t / g.ty;
        const int y = by * g.ty + ty, z = bz * g.tz + tz;
        const bool valid = tz < g.tz && y < g.ny && z < g.nz;
        const int64_t row_base = valid ? ((int64_t)z * g.ny + y) * g.a : 0;
        const int skew = ty + tz;
        const int e_a = tz, e_b = 16 + ty;               // this line's slots in s_e
        const bool pub = valid && ((ty == g.ty - 1 && y + 1 < g.ny) || (HAVE_B && tz == g.tz - 1 && z + 1 < g.nz));
        double xprev = 0.0;
        LnSet set[LN_K];
#define LN_ISSUE(I, T) { int q_ = (T) - skew; q_ = q_ < 0 ? 0 : (q_ >= nq ? nq - 1 : q_); ln_load<HAVE_B>(set[I], (int64_t)tile * nq + q_, t, row_base + (int64_t)q_ * LN_CH, img, mimg, rhs); }
        LN_ISSUE(0, 0) LN_ISSUE(1, 1) LN_ISSUE(2, 2) LN_ISSUE(3, 3)
        static_assert(LN_K == 5, "the step loop is unrolled by hand");
        ln_barrier();                                    // (the edge wave has written the chunks of step 0)
        int buf = 0;
#define LN_STEP(I)                                                                                                \
        if (S0 + (I) < nsteps) {                                                                                  \
            const int S = S0 + (I);                                                                               \
            LN_ISSUE(((I) + LN_K - 1) % LN_K, S + LN_K - 1)                                                       \
            const LnSet &c = set[I];                                                                              \
            const int q = S - skew;                                                                               \
            const bool act = valid && q >= 0 && q < nq;                                                           \
            const int64_t r0 = row_base + (int64_t)(q < 0 ? 0 : (q >= nq ? nq - 1 : q)) * LN_CH;                  \
            double xa[LN_CH], xb[LN_CH];                                                                          \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) {                                                   \
                xa[u] = ty > 0 ? s_x[buf ^ 1][t > 0 ? t - 1 : 0][u] : s_e[S & 1][e_a][u];                         \
                xb[u] = HAVE_B ? (tz > 0 ? s_x[buf ^ 1][t >= g.ty ? t - g.ty : 0][u] : s_e[S & 1][e_b][u]) : 0.0; \
            }                                                                                                     \
            double out[LN_CH];                                                                                    \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) {                                                   \
                /* the row's chain in CRS order: -b, -a, -1 (ascending columns), from 0.0 */                      \
                const unsigned int m = (c.m >> (8 * u)) & 255u;                                                   \
                double acc = 0.0;                                                                                 \
                if (HAVE_B && (m & 1u)) acc = fma(c.v0[u], xb[u], acc);                                           \
                if (m & 2u) acc = fma(c.v1[u], xa[u], acc);                                                       \
                if (m & 4u) acc = fma(c.v2[u], xprev, acc);                                                       \
                xprev = (c.rr[u] - acc) / c.dd[u];                                                                \
                out[u] = xprev;                                                                                   \
                if (acc_out && act) acc_out[r0 + u] = acc;                                                        \
            }                                                                                                     \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) s_x[buf][t][u] = out[u];                            \
            if (act) {                                                                                            \
                if (pub) {                                                                                        \
                    /* agent scope: lines on a tile's high edges are read by other workgroups while this one runs */ \
                    _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                             \
                        __hip_atomic_store(reinterpret_cast<unsigned long long *>(x + r0 + u), (unsigned long long)__double_as_longlong(out[u]), \
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                           \
                } else {                                                                                          \
                    _Pragma("unroll") for (int u = 0; u < LN_CH; u += 2) { lines_v2d o2; o2.x = out[u]; o2.y = out[u + 1]; *reinterpret_cast<lines_v2d *>(x + r0 + u) = o2; } \
                }                                                                                                 \
            }                                                                                                     \
            ln_barrier();                                                                                         \
            buf ^= 1;                                                                                             \
        }
        for (int S0 = 0; S0 < nsteps; S0 += LN_K) { LN_STEP(0) LN_STEP(1) LN_STEP(2) LN_STEP(3) LN_STEP(4) }
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
        const int64_t src_base = have ? ((int64_t)z * g.ny + y) * g.a - (is_a ? g.a : g.b) : 0;
        const int skew = ty + tz;
        const int es = e < LN_EDGE ? e : 0;
        bool gave_up = false;
        double pf[LN_KE][LN_CH];
        // Every vector-memory access of this wave is inline asm with hand-counted waits: with its polling loop in sight the
        // compiler would drain all loads in flight before each use — the requests LN_KE steps ahead included, one uncached
        // round trip per step for the whole workgroup (1.6 us per step measured).  A request is LN_CH loads; between a
        // chunk's request and its use lie LN_KE - 1 later requests, so vmcnt(LN_CH * (LN_KE - 1)) = 12 says it has arrived
        // (loads return in issue order).  The "+v" ties keep the compiler from touching the registers before that.
#define LN_EISSUE(I, T) { int q_ = (T) - skew; q_ = q_ < 0 ? 0 : (q_ >= nq ? nq - 1 : q_);                          \
            const double *src_ = x + src_base + (int64_t)q_ * LN_CH;                                              \
            _Pragma("unroll") for (int u = 0; u < LN_CH; ++u)                                                     \
                asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(pf[I][u]) : "v"(src_ + u) : "memory"); }
#define LN_EPUT(I, T) { const int q_ = (T) - skew;                                                                \
            asm volatile("s_waitcnt vmcnt(12)" : "+v"(pf[I][0]), "+v"(pf[I][1]), "+v"(pf[I][2]), "+v"(pf[I][3]) :: "memory"); \
            if (have && q_ >= 0 && q_ < nq && !gave_up) {                                                         \
                const double *src_ = x + src_base + (int64_t)q_ * LN_CH;                                          \
                const long long t0_ = wall_clock64();                                                             \
                for (int spins_ = 0;; ++spins_) {                                                                 \
                    bool ok_ = true;                                                                              \
                    _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) ok_ = ok_ && (unsigned long long)__double_as_longlong(pf[I][u]) != LINES_PENDING; \
                    if (LN_DEBUG && spins_ == 1) atomicAdd(counter + 5, 1);                                       \
                    if (LN_DEBUG && spins_ >= 1) atomicAdd(counter + 6, 1);                                       \
                    if (ok_) break;                                                                               \
                    if ((spins_ & 15) == 15 && (lines_ld_agent_int(fail) != 0 || wall_clock64() - t0_ > LINES_WAIT_TICKS)) { gave_up = true; break; } \
                    __builtin_amdgcn_s_sleep(1);                                                                  \
                    _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) pf[I][u] = __longlong_as_double((long long)lines_ld_agent(src_ + u)); \
                }                                                                                                 \
            }                                                                                                     \
            if (e < LN_EDGE) { _Pragma("unroll") for (int u = 0; u < LN_CH; ++u) s_e[(T) & 1][es][u] = pf[I][u]; } }
        static_assert(LN_CH * (LN_KE - 1) == 12, "the hand-counted wait above");
        LN_EISSUE(0, 0) LN_EISSUE(1, 1) LN_EISSUE(2, 2) LN_EISSUE(3, 3)
        static_assert(LN_KE == 4, "the step loop is unrolled by hand");
        LN_EPUT(0, 0)
        LN_EISSUE(0, 4)
        ln_barrier();
        // iteration S: the chunk of step S + 1 into LDS, the request for step S + 1 + LN_KE
#define LN_ESTEP(I)                                                                                               \
        if (S0 + (I) < nsteps) {                                                                                  \
            const int S = S0 + (I);                                                                               \
            LN_EPUT(((I) + 1) % LN_KE, S + 1)                                                                     \
            LN_EISSUE(((I) + 1) % LN_KE, S + 1 + LN_KE)                                                           \
            ln_barrier();                                                                                         \
        }
        for (int S0 = 0; S0 < nsteps; S0 += LN_KE) { LN_ESTEP(0) LN_ESTEP(1) LN_ESTEP(2) LN_ESTEP(3) }
#undef LN_ESTEP
#undef LN_EPUT
#undef LN_EISSUE
        if (gave_up) {
            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fail_host) __hip_atomic_store(fail_host, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

void launch_lines_build(const LinesPlan &P, const CsrDev &L, const double *D, int *flags, hipStream_t s)
{
    LinesGeom g; g.n = P.n; g.a = P.a; g.b = P.b; g.ny = P.ny; g.nz = P.nz; g.ty = P.ty; g.tz = P.tz; g.nby = P.nby; g.nbz = P.nbz;
    const int64_t blocks = (int64_t)P.nby * P.nbz * (P.a / LN_CH);
    hipLaunchKernelGGL(lines_build_kernel, dim3((unsigned)blocks), dim3(LINES_WG), 0, s, g, L.row_ptr, L.col, L.val, D, P.img, P.mimg, flags);
}

size_t lines_image_doubles(const LinesPlan &P) { return (size_t)P.nby * P.nbz * (P.a / LN_CH) * (P.b ? 4 : 3) * (LINES_WG * LN_CH); }
size_t lines_image_masks(const LinesPlan &P) { return (size_t)P.nby * P.nbz * (P.a / LN_CH) * LINES_WG; }

void launch_sweep_lines(const LinesPlan &P, double *x, const double *rhs, double *acc_out, hipStream_t s)
{
    int gp = (int)std::min<int64_t>(((int64_t)P.n + ITS_BLOCK - 1) / ITS_BLOCK, 4 * ITS_NPART);
    hipLaunchKernelGGL(lines_prepare_kernel, dim3(gp), dim3(ITS_BLOCK), 0, s, reinterpret_cast<unsigned long long *>(x), (int64_t)P.n, P.counter);
    LinesGeom g; g.n = P.n; g.a = P.a; g.b = P.b; g.ny = P.ny; g.nz = P.nz; g.ty = P.ty; g.tz = P.tz; g.nby = P.nby; g.nbz = P.nbz;
    const int tiles = P.nby * P.nbz;
#define ITS_LAUNCHLN(B) \
    hipLaunchKernelGGL((sweep_lines_kernel<B>), dim3(tiles), dim3(LINES_THREADS), 0, s, g, P.img, P.mimg, P.counter, P.fail, P.fail_host, x, rhs, acc_out)
    if (P.b) ITS_LAUNCHLN(true); else ITS_LAUNCHLN(false);
#undef ITS_LAUNCHLN
    if (LN_DEBUG) {
        int c[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(c, P.counter, sizeof c, hipMemcpyDeviceToHost);
        fprintf(stderr, "lines debug: tiles %d, slow-path entries %d, re-polls %d\n", c[0], c[5], c[6]);
        (void)hipMemset(P.counter + 5, 0, 2 * sizeof(int));
    }
}

}  // namespace its
