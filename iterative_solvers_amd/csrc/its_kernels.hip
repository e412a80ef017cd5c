// its_kernels.hip — hand-written gfx950 kernels for the iterative-solver hot path.
// All kernels: 256-thread workgroups (4 wavefronts of 64), persistent grids walking their share of the
// work (2-5 workgroups per CU for the SpMV pipelines, ITS_NPART for the vector kernels); HBM-bound, no MFMA.
#include "its_kernels.h"
#include "its_epilogue.h"
#include <cstdlib>
#include <cstring>

namespace its {

// ======================================================================================
// SpMV, scalar CSR, bit-exact: the "stream" kernels.  The col/val slice of a row block (<= 256 rows,
// <= SPMV_NNZ_MAX entries, plan built at upload) is copied from HBM into LDS with 16-byte loads by all 256
// lanes, then lane i walks row i from LDS in CRS order with fused multiply-adds from 0.0 — the operations
// and the order of spmv_crs_cpu (kernels.hpp:380-409).  Three pipelines share that idea: v2 (rows of 9..32
// entries), v4 / v5 (rows of at most 8) and the two-phase kernel for irregular rows.
// ======================================================================================
// --------------------------------------------------------------------------------------
// v2 of the stream kernel: software-pipelined, one barrier per row block.
//   * while one lane per row walks row block rb out of LDS buffer `cur`, the col/val/row_ptr
//     segment of the NEXT row block is in flight into registers; it is committed to the other
//     LDS buffer at the top of the next trip.
//   * measured alternatives that were NOT faster on MI355X (256^3 7-pt): issuing the
//     x gathers before unconditional prefetch loads so that a counted vmcnt keeps the prefetch
//     in flight (0.40-0.43 ms vs 0.37 ms), and a wave-autonomous variant without workgroup
//     barriers (0.41 ms).
//   * the walk gathers x in batches of 8 entries (the L2 round trips overlap); the fused
//     multiply-adds still run strictly in CRS order from 0.0 — bit-exact.
//   * work map (a.map_mode): 0 = each workgroup owns one contiguous run of row blocks
//     (XCD-aware slot); 1 = each XCD (blockIdx % 8) owns a contiguous eighth of the row blocks
//     and its G/8 workgroups interleave over it, so an XCD works inside one compact sliding
//     window (few open DRAM pages, x window resident in its L2).
// Row blocks are described by blk_desc[rb] = {first row, rows, first nnz, end nnz}.
// --------------------------------------------------------------------------------------
constexpr int S2_CAP = SPMV_NNZ_MAX + 8;
// non-temporal hint on the col/val/row_ptr stream of spmv_stream4_kernel: measured a LOSS inside the CG loop
// (2104 vs 2123 it/s), although those bytes are read once — kept as a switch for re-measurement
constexpr int SPMV4_NT = 0;
static_assert(SPMV_NNZ_MAX + 3 <= 2 * 4 * ITS_BLOCK, "staging is written for two slices of 4 entries per lane");

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef double v2d_t __attribute__((ext_vector_type(2)));
// streaming loads: NT != 0 marks the col/val/row_ptr stream non-temporal so that it does not
// push the gathered x window out of the XCD's L2
template <int NT> __device__ __forceinline__ int4 ld_i4(const int *p)
{
    const v4i_t v = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(p)) : *reinterpret_cast<const v4i_t *>(p);
    return make_int4(v.x, v.y, v.z, v.w);
}
template <int NT> __device__ __forceinline__ double2 ld_d2(const double *p)
{
    const v2d_t v = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p)) : *reinterpret_cast<const v2d_t *>(p);
    return make_double2(v.x, v.y);
}
template <int NT> __device__ __forceinline__ int ld_i1(const int *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int NT> __device__ __forceinline__ void st_d1(double *p, double v)
{
    if (NT & 2) __builtin_nontemporal_store(v, p); else *p = v;
}

// IDX != 0: stream the 1-byte column-offset dictionary index instead of col (see spmv_stream4_kernel, CMP = 1)
//   1  decoded to col - row when the block is committed to LDS (4 bytes per entry there)
//   2  kept as a byte in LDS and decoded in the walk: room for the larger row blocks at four workgroups per CU.
//      Chosen for rows of <= 8 entries (7-pt: plain SpMV 0.304 -> 0.292 ms); for 27-entry rows the second
//      dependent LDS read per entry costs more than it saves (GMRES trip 0.542 -> 0.566 ms), they keep 1
template <int EPI, bool DOT, int IDX>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_stream2_kernel(
    int nb, const int4 *__restrict__ blk_desc, const int *__restrict__ rp,
    const int *__restrict__ col, const double *__restrict__ val, int n_rows,
    const unsigned char *__restrict__ cidx, const int *__restrict__ off_dict, SpmvArgs a)
{
    // bit 0: non-temporal col/val/row_ptr loads (a loss: plain SpMV 0.291 -> 0.324 ms); bit 1: non-temporal store of
    // the result (plain 0.291 -> 0.285 ms, Jacobi trip 0.341 -> 0.337 ms) — not for the sweep, whose result the
    // next level gathers right away
    constexpr int NT = EPI == EPI_SWEEP ? 0 : 2;
    // IDX == 2: 9 instead of 12 bytes of LDS per entry, so four workgroups per CU still fit with blocks of up to
    // SPMV_NNZ_BIG entries (full 256-row blocks for 7-pt)
    constexpr int CAP = IDX == 2 ? SPMV_NNZ_BIG + 8 : S2_CAP;
    constexpr int NNZ_CAP = IDX == 2 ? SPMV_NNZ_BIG : SPMV_NNZ_MAX;
    __shared__ __attribute__((aligned(16))) double s_val[2][CAP];
    __shared__ __attribute__((aligned(16))) int s_col[2][IDX == 2 ? 4 : CAP];    // IDX == 1: col - row
    __shared__ __attribute__((aligned(16))) unsigned char s_idx[2][IDX == 2 ? CAP : 16];
    __shared__ int s_rp[2][ITS_BLOCK + 4];
    __shared__ int s_dict[IDX ? 256 : 1];
    __shared__ double s_red[4];

    const int t = threadIdx.x;
    const int G = gridDim.x;
    if (IDX) { s_dict[t] = off_dict[t]; __syncthreads(); }
    int rb0, rb1, rstep;
    if (a.map_mode == 0) {
        const int slot = xcd_slot(blockIdx.x, G);
        rb0 = (int)(((int64_t)slot * nb) / G);
        rb1 = (int)(((int64_t)(slot + 1) * nb) / G);
        rstep = 1;
    } else {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        const int X0 = (int)(((int64_t)xcd * nb) / 8), X1 = (int)(((int64_t)(xcd + 1) * nb) / 8);
        rb0 = X0 + jx; rb1 = X1; rstep = G >> 3;
    }
    const int niter = rb0 < rb1 ? (rb1 - rb0 + rstep - 1) / rstep : 0;
    const int rb_last = rb0 + (niter - 1) * rstep;

    double lane_sum = 0.0, lane_max = 0.0;

    // staging registers of the block in flight: named scalars (arrays would go to scratch)
    int4 rcA, rcB;
    double2 rvA0, rvA1, rvB0, rvB1;
    int rrp;

#define S2_ISSUE(d)                                                                       \
    do {                                                                                  \
        const int base_ = (d).z & ~3;                                                     \
        const int iA_ = base_ + 4 * t, iB_ = iA_ + 4 * ITS_BLOCK;                         \
        if (iA_ < (d).w) {                                                                \
            if (IDX) rcA.x = *reinterpret_cast<const int *>(cidx + iA_);                  \
            else rcA = ld_i4<NT & 1>(col + iA_);                                          \
            rvA0 = ld_d2<NT & 1>(val + iA_);                                              \
            rvA1 = ld_d2<NT & 1>(val + iA_ + 2);                                          \
        }                                                                                 \
        if (iB_ < (d).w) {                                                                \
            if (IDX) rcB.x = *reinterpret_cast<const int *>(cidx + iB_);                  \
            else rcB = ld_i4<NT & 1>(col + iB_);                                          \
            rvB0 = ld_d2<NT & 1>(val + iB_);                                              \
            rvB1 = ld_d2<NT & 1>(val + iB_ + 2);                                          \
        }                                                                                 \
        if (t < (d).y) rrp = ld_i1<NT & 1>(rp + (d).x + t);                               \
    } while (0)
#define S2_DECODE(w)                                                                      \
    make_int4(s_dict[(unsigned)(w) & 255u], s_dict[((unsigned)(w) >> 8) & 255u],          \
              s_dict[((unsigned)(w) >> 16) & 255u], s_dict[(unsigned)(w) >> 24])
#define S2_COMMIT(d, buf)                                                                 \
    do {                                                                                  \
        const int base_ = (d).z & ~3;                                                     \
        const int oA_ = 4 * t, oB_ = oA_ + 4 * ITS_BLOCK;                                 \
        if (base_ + oA_ < (d).w) {                                                        \
            if (IDX == 2) *reinterpret_cast<int *>(&s_idx[buf][oA_]) = rcA.x;             \
            else *reinterpret_cast<int4 *>(&s_col[buf][oA_]) = IDX ? S2_DECODE(rcA.x) : rcA; \
            *reinterpret_cast<double2 *>(&s_val[buf][oA_]) = rvA0;                        \
            *reinterpret_cast<double2 *>(&s_val[buf][oA_ + 2]) = rvA1;                    \
        }                                                                                 \
        if (base_ + oB_ < (d).w) {                                                        \
            if (IDX == 2) *reinterpret_cast<int *>(&s_idx[buf][oB_]) = rcB.x;             \
            else *reinterpret_cast<int4 *>(&s_col[buf][oB_]) = IDX ? S2_DECODE(rcB.x) : rcB; \
            *reinterpret_cast<double2 *>(&s_val[buf][oB_]) = rvB0;                        \
            *reinterpret_cast<double2 *>(&s_val[buf][oB_ + 2]) = rvB1;                    \
        }                                                                                 \
        if (t < (d).y) s_rp[buf][t] = rrp;                                                \
        if (t == 0) s_rp[buf][(d).y] = (d).w;                                             \
    } while (0)

    int cur = 0;
    bool have = false;
    int4 d = make_int4(0, 0, 0, 0), dn = d;
    if (niter > 0) {
        d = blk_desc[rb0]; d.y &= 0xffff;
        dn = blk_desc[min(rb0 + rstep, rb_last)]; dn.y &= 0xffff;
    }
    for (int it = 0; it < niter; ++it) {
        const int rb = rb0 + it * rstep;
        int4 dnn = blk_desc[min(rb + 2 * rstep, rb_last)]; dnn.y &= 0xffff;
        const int rs = d.x, nr = d.y;
        const int row = rs + t;
        double acc = 0.0, acc2 = t < nr ? chain2_start<EPI>(a, row) : 0.0, diag = 0.0;
        // (a matrix with a dictionary has no row longer than its 256 offsets: no long-row path under IDX)
        const bool lng = !IDX && d.w - d.z > NNZ_CAP;
        if (!lng) {
            if (!have) S2_ISSUE(d);
            S2_COMMIT(d, cur);
        }
        __syncthreads();
        // the next block's col/val/row_ptr go out before the walk and stay in flight during it
        have = false;
        if (it + 1 < niter && (IDX || dn.w - dn.z <= NNZ_CAP)) { S2_ISSUE(dn); have = true; }
        if (!lng) {
            if (t < nr) {
                const int base = d.z & ~3;
                const int j0 = s_rp[cur][t] - base, j1 = s_rp[cur][t + 1] - base;
                for (int j = j0; j < j1; j += 8) {
                    int c[8]; double v[8], xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int jj = (j + u < j1) ? j + u : j;
                        c[u] = IDX == 2 ? row + s_dict[s_idx[cur][jj]] : (IDX ? row + s_col[cur][jj] : s_col[cur][jj]);
                        v[u] = s_val[cur][jj];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = a.x[c[u]];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j + u < j1) chain_step<EPI>(c[u], row, v[u], xv[u], acc, acc2, diag);
                }
            }
        } else {
            // a single row longer than the LDS capacity: stream it through buffer `cur` in
            // chunks; lane 0 keeps the in-order chain.  (Rare: power-law tails.)
            for (int c0 = d.z; c0 < d.w; c0 += NNZ_CAP) {
                const int c1 = min(d.w, c0 + NNZ_CAP);
                const int base = c0 & ~3;
                __syncthreads();
                for (int i = base + 4 * t; i < c1; i += 4 * ITS_BLOCK) {
                    const int o = i - base;
                    *reinterpret_cast<int4 *>(&s_col[cur][o]) = *reinterpret_cast<const int4 *>(col + i);
                    *reinterpret_cast<double2 *>(&s_val[cur][o]) = *reinterpret_cast<const double2 *>(val + i);
                    *reinterpret_cast<double2 *>(&s_val[cur][o + 2]) = *reinterpret_cast<const double2 *>(val + i + 2);
                }
                __syncthreads();
                if (t == 0) {
                    for (int j = c0 - base; j < c1 - base; ++j) {
                        const int cc = s_col[cur][j];
                        chain_step<EPI>(cc, row, s_val[cur][j], a.x[cc], acc, acc2, diag);
                    }
                }
            }
            __syncthreads();
        }
        if (t < nr) {
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
            epi_operands<EPI, DOT>(a, row, e0, e1, e2);
            const double outv = row_finish<EPI, DOT>(acc, acc2, diag, e0, e1, e2, lane_sum, lane_max);
            if (a.y) st_d1<NT>(a.y + row, outv);             // y == nullptr: only the norm is wanted
            sweep_note<EPI>(a, row, acc);
        }
        cur ^= 1;
        d = dn; dn = dnn;
    }
#undef S2_ISSUE
#undef S2_DECODE
#undef S2_COMMIT
    if (DOT || EpiKind<EPI>::norm) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) {
            if (a.part_accum) a.part_sum[blockIdx.x] += bs;
            else {
                a.part_sum[blockIdx.x] = bs;
                for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_sum[z] = 0.0;
            }
        }
        if (EpiKind<EPI>::norm && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (t == 0) {
                if (a.part_accum) a.part_max[blockIdx.x] = fmax(a.part_max[blockIdx.x], bm);
                else {
                    a.part_max[blockIdx.x] = bm;
                    for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_max[z] = 0.0;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// v4 of the stream kernel: two row blocks of col/val/row_ptr in flight, gathers issued FIRST.
//
// gfx9 returns vector-memory operations in issue order (one vmcnt queue for loads and stores).
// In v2 the x gathers are issued after the prefetch of the next row block, so the first fused
// multiply-add waits for that whole prefetch, and the y store is the newest entry of the queue
// when the next trip needs its staged registers: every trip pays prefetch latency + store
// acknowledge with nothing else in flight.  Here each trip
//   1. commits the register set that was issued TWO trips ago (long since complete),
//   2. stores the previous trip's result, issues the epilogue loads, the LDS reads and the gathers,
//   3. only then issues the loads of the row block two trips ahead — unconditionally (indices
//      clamped to 0), as straight-line code, so that the waits in front of the multiply-adds can
//      count them as "newer than the gathers" and leave them in flight,
//   4. runs the in-order multiply-add chain and parks the result for step 2 of the next trip.
// The loop is unrolled by two over the register sets P and Q (no register copies).  Rows longer
// than 8 entries run their leading batches of 8 before step 2's final batch.  blk_desc[rb].y
// carries rows | (longest row of the block << 16).
// --------------------------------------------------------------------------------------
struct StageRegs { int4 cA, cB; double2 vA0, vA1, vB0, vB1; int rp; };

// CMP: 0 = plain col/val streams (12 B per nonzero); 1 = 1-byte column-offset index + val (9 B);
//      2 = 1-byte offset index + 1-byte value index (2 B; opt-in, matrices with <= 256 distinct values).
// Both dictionaries are lossless: the decoded column and the decoded fp64 value are the CRS entries bit for bit.
template <int EPI, bool DOT, int CMP>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_stream4_kernel(
    int nb, const int4 *__restrict__ blk_desc, const int *__restrict__ rp,
    const int *__restrict__ col, const double *__restrict__ val, int n_rows,
    const unsigned char *__restrict__ cidx, const int *__restrict__ off_dict,
    const unsigned char *__restrict__ vidx, const double *__restrict__ val_dict, SpmvArgs a)
{
    constexpr bool IDX = CMP >= 1, VD = CMP == 2;
    __shared__ double s_vdict[VD ? 256 : 1];
    constexpr int S4_CAP = SPMV_NNZ_BIG + 8;
    static_assert(SPMV_NNZ_BIG + 3 <= 2 * 4 * ITS_BLOCK, "staging is written for two slices of 4 entries per lane");
    __shared__ __attribute__((aligned(16))) double s_val[2][S4_CAP];
    __shared__ __attribute__((aligned(16))) int s_col[2][S4_CAP];    // IDX: col - row, decoded at commit
    __shared__ int s_rp[2][ITS_BLOCK + 4];
    __shared__ int s_dict[IDX ? 256 : 1];
    __shared__ double s_red[4];

    const int t = threadIdx.x;
    const int G = gridDim.x;
    if (IDX) {
        s_dict[t] = off_dict[t];
        if (VD) s_vdict[t] = val_dict[t];
        __syncthreads();
    }
    int rb0, rb1, rstep;
    if (a.map_mode == 0) {
        const int slot = xcd_slot(blockIdx.x, G);
        rb0 = (int)(((int64_t)slot * nb) / G);
        rb1 = (int)(((int64_t)(slot + 1) * nb) / G);
        rstep = 1;
    } else {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        const int X0 = (int)(((int64_t)xcd * nb) / 8), X1 = (int)(((int64_t)(xcd + 1) * nb) / 8);
        rb0 = X0 + jx; rb1 = X1; rstep = G >> 3;
    }
    const int niter = rb0 < rb1 ? (rb1 - rb0 + rstep - 1) / rstep : 0;

    double lane_sum = 0.0, lane_max = 0.0;
    double pend_val = 0.0;
    int pend_row = -1;

#define S4_DESC(i) ((i) < niter ? blk_desc[rb0 + (i) * rstep] : make_int4(0, 0, 0, 0))
    // all loads of one row block, unconditional: lanes past the end read entry 0
#define S4_ISSUE(R, d)                                                                    \
    do {                                                                                  \
        const int base_ = (d).z & ~3;                                                     \
        const int iA_ = base_ + 4 * t, iB_ = iA_ + 4 * ITS_BLOCK;                         \
        const int jA_ = iA_ < (d).w ? iA_ : 0, jB_ = iB_ < (d).w ? iB_ : 0;               \
        const int jr_ = t < ((d).y & 0xffff) ? (d).x + t : 0;                             \
        if (IDX) R.cA.x = ld_i1<SPMV4_NT>(reinterpret_cast<const int *>(cidx + jA_));     \
        else R.cA = ld_i4<SPMV4_NT>(col + jA_);                                           \
        if (VD) R.cA.y = *reinterpret_cast<const int *>(vidx + jA_);                      \
        else { R.vA0 = ld_d2<SPMV4_NT>(val + jA_); R.vA1 = ld_d2<SPMV4_NT>(val + jA_ + 2); } \
        if (IDX) R.cB.x = ld_i1<SPMV4_NT>(reinterpret_cast<const int *>(cidx + jB_));     \
        else R.cB = ld_i4<SPMV4_NT>(col + jB_);                                           \
        if (VD) R.cB.y = *reinterpret_cast<const int *>(vidx + jB_);                      \
        else { R.vB0 = ld_d2<SPMV4_NT>(val + jB_); R.vB1 = ld_d2<SPMV4_NT>(val + jB_ + 2); } \
        R.rp = ld_i1<SPMV4_NT>(rp + jr_);                                                 \
    } while (0)
    // four packed dictionary indices -> four (col - row) offsets
#define S4_DECODE(w)                                                                      \
    make_int4(s_dict[(unsigned)(w) & 255u], s_dict[((unsigned)(w) >> 8) & 255u],          \
              s_dict[((unsigned)(w) >> 16) & 255u], s_dict[(unsigned)(w) >> 24])
#define S4_VDEC0(w) make_double2(s_vdict[(unsigned)(w) & 255u], s_vdict[((unsigned)(w) >> 8) & 255u])
#define S4_VDEC1(w) make_double2(s_vdict[((unsigned)(w) >> 16) & 255u], s_vdict[(unsigned)(w) >> 24])
#define S4_COMMIT(R, d, buf)                                                              \
    do {                                                                                  \
        const int base_ = (d).z & ~3;                                                     \
        const int oA_ = 4 * t, oB_ = oA_ + 4 * ITS_BLOCK;                                 \
        const int nr_ = (d).y & 0xffff;                                                   \
        if (base_ + oA_ < (d).w) {                                                        \
            *reinterpret_cast<int4 *>(&s_col[buf][oA_]) = IDX ? S4_DECODE(R.cA.x) : R.cA; \
            *reinterpret_cast<double2 *>(&s_val[buf][oA_]) = VD ? S4_VDEC0(R.cA.y) : R.vA0;     \
            *reinterpret_cast<double2 *>(&s_val[buf][oA_ + 2]) = VD ? S4_VDEC1(R.cA.y) : R.vA1; \
        }                                                                                 \
        if (base_ + oB_ < (d).w) {                                                        \
            *reinterpret_cast<int4 *>(&s_col[buf][oB_]) = IDX ? S4_DECODE(R.cB.x) : R.cB; \
            *reinterpret_cast<double2 *>(&s_val[buf][oB_]) = VD ? S4_VDEC0(R.cB.y) : R.vB0;     \
            *reinterpret_cast<double2 *>(&s_val[buf][oB_ + 2]) = VD ? S4_VDEC1(R.cB.y) : R.vB1; \
        }                                                                                 \
        if (t < nr_) s_rp[buf][t] = R.rp;                                                 \
        if (t == 0) s_rp[buf][nr_] = (d).w;                                               \
    } while (0)
    // one batch of 8 entries of this lane's row: LDS reads, then the x gathers (column 0 for
    // slots past the end of the row: never a stale LDS word as an address)
#define S4_BATCH_LOAD(buf)                                                                \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                       \
        const bool ok_ = j + u < j1;                                                      \
        const int jj_ = ok_ ? j + u : 0;                                                  \
        const int cc_ = s_col[buf][jj_];                                                  \
        c[u] = ok_ ? (IDX ? row + cc_ : cc_) : 0; v[u] = s_val[buf][jj_];                 \
    }                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) xv[u] = a.x[c[u]];
#define S4_BATCH_FMA()                                                                    \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                       \
        if (j + u < j1) chain_step<EPI>(c[u], row, v[u], xv[u], acc, acc2, diag);         \
    }

#define S4_BODY(R, buf)                                                                                   \
    {                                                                                                     \
        const int4 d = dA;                                                                                \
        const int4 dD = S4_DESC(it + 3);                                                                  \
        const int nr = d.y & 0xffff, nbat = (((d.y >> 16) + 7) >> 3);                                     \
        const int row = d.x + t;                                                                          \
        const bool lng = d.w - d.z > SPMV_NNZ_BIG;                                                        \
        if (!lng) S4_COMMIT(R, d, buf);                                                                   \
        __syncthreads();                                                                                  \
        if (pend_row >= 0) { a.y[pend_row] = pend_val; pend_row = -1; }                                   \
        double e0 = 0.0, e1 = 0.0, e2 = 0.0;                                                              \
        if (t < nr) epi_operands<EPI, DOT>(a, row, e0, e1, e2);                                           \
        double acc = 0.0, acc2 = 0.0, diag = 0.0;                                                         \
        int c[8]; double v[8], xv[8];                                                                     \
        int j = 0, j1 = 0;                                                                                \
        if (!lng) {                                                                                       \
            if (t < nr) { const int base = d.z & ~3; j = s_rp[buf][t] - base; j1 = s_rp[buf][t + 1] - base; } \
            _Pragma("unroll 1") for (int bb = 1; bb < nbat; ++bb) {                                       \
                S4_BATCH_LOAD(buf) S4_BATCH_FMA() j += 8;                                                 \
            }                                                                                             \
            S4_BATCH_LOAD(buf)                                                                            \
        } else {                                                                                          \
            /* one row longer than the LDS capacity: stream it through the buffer in chunks, lane 0 */    \
            /* keeps the in-order chain (rare: power-law tails) */                                        \
            for (int c0 = d.z; c0 < d.w; c0 += SPMV_NNZ_BIG) {                                            \
                const int c1 = min(d.w, c0 + SPMV_NNZ_BIG);                                               \
                const int base = c0 & ~3;                                                                 \
                __syncthreads();                                                                          \
                for (int i = base + 4 * t; i < c1; i += 4 * ITS_BLOCK) {                                  \
                    const int o = i - base;                                                               \
                    *reinterpret_cast<int4 *>(&s_col[buf][o]) = *reinterpret_cast<const int4 *>(col + i); \
                    *reinterpret_cast<double2 *>(&s_val[buf][o]) = *reinterpret_cast<const double2 *>(val + i); \
                    *reinterpret_cast<double2 *>(&s_val[buf][o + 2]) = *reinterpret_cast<const double2 *>(val + i + 2); \
                }                                                                                         \
                __syncthreads();                                                                          \
                if (t == 0) {                                                                             \
                    for (int q = c0 - base; q < c1 - base; ++q) {                                         \
                        const int cc = s_col[buf][q];                                                     \
                        chain_step<EPI>(cc, row, s_val[buf][q], a.x[cc], acc, acc2, diag);                \
                    }                                                                                     \
                }                                                                                         \
            }                                                                                             \
            __syncthreads();                                                                              \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { c[u] = 0; v[u] = 0.0; xv[u] = 0.0; }          \
        }                                                                                                 \
        S4_ISSUE(R, dC);                                                                                  \
        S4_BATCH_FMA()                                                                                    \
        if (t < nr) {                                                                                     \
            const double outv = row_finish<EPI, DOT>(acc, acc2, diag, e0, e1, e2, lane_sum, lane_max);    \
            if (a.y) { pend_val = outv; pend_row = row; }   /* y == nullptr: only the norm is wanted */   \
        }                                                                                                 \
        dA = dB; dB = dC; dC = dD;                                                                        \
        ++it;                                                                                             \
    }

    if (niter > 0) {
        StageRegs P, Q;
        int4 dA = S4_DESC(0), dB = S4_DESC(1), dC = S4_DESC(2);
        S4_ISSUE(P, dA);
        S4_ISSUE(Q, dB);
        const int niter2 = (niter + 1) & ~1;
        int it = 0;
        while (it < niter2) {
            S4_BODY(P, 0)
            S4_BODY(Q, 1)
        }
        if (pend_row >= 0) a.y[pend_row] = pend_val;
    }
    if (DOT || EpiKind<EPI>::norm) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) {
            if (a.part_accum) a.part_sum[blockIdx.x] += bs;
            else {
                a.part_sum[blockIdx.x] = bs;
                for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_sum[z] = 0.0;
            }
        }
        if (EpiKind<EPI>::norm && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (t == 0) {
                if (a.part_accum) a.part_max[blockIdx.x] = fmax(a.part_max[blockIdx.x], bm);
                else {
                    a.part_max[blockIdx.x] = bm;
                    for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_max[z] = 0.0;
                }
            }
        }
    }
}


// --------------------------------------------------------------------------------------
// v5 of the stream kernel (rows of at most 8 entries): a THREE-stage software pipeline.
// One trip s of a workgroup
//   1. commits row block s+1 (register set issued two trips ago) to LDS, barrier,
//   2. reads row block s+1 out of LDS and ISSUES its x gathers (into the gather set of that parity),
//   3. issues the col/val/row_ptr loads of row block s+3,
//   4. runs the multiply-add chain of row block s, whose gathers went out ONE TRIP AGO, and parks the result
//      (stored at step 2 of the next trip).
// In v4 every trip waits for its own gathers (an L2 or HBM round trip with nothing else of that workgroup in
// flight); here the round trip is hidden behind the next trip's LDS work and load issue, so a trip costs its
// instruction issue, not a memory latency.  Same arithmetic and order as v4: bit-exact.
// --------------------------------------------------------------------------------------
struct GatherRegs { int c[8]; double v[8], x[8]; int j, j1, row, nr; double e0, e1, e2; };

template <int EPI, bool DOT, int CMP>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_stream5_kernel(
    int nb, const int4 *__restrict__ blk_desc, const int *__restrict__ rp,
    const int *__restrict__ col, const double *__restrict__ val, int n_rows,
    const unsigned char *__restrict__ cidx, const int *__restrict__ off_dict,
    const unsigned char *__restrict__ vidx, const double *__restrict__ val_dict, SpmvArgs a)
{
    constexpr bool IDX = CMP >= 1, VD = CMP == 2;
    __shared__ double s_vdict[VD ? 256 : 1];
    __shared__ __attribute__((aligned(16))) double s_val[2][S2_CAP];
    __shared__ __attribute__((aligned(16))) int s_col[2][S2_CAP];    // IDX: col - row, decoded at commit
    __shared__ int s_rp[2][ITS_BLOCK + 4];
    __shared__ int s_dict[IDX ? 256 : 1];
    __shared__ double s_red[4];

    const int t = threadIdx.x;
    const int G = gridDim.x;
    if (IDX) {
        s_dict[t] = off_dict[t];
        if (VD) s_vdict[t] = val_dict[t];
        __syncthreads();
    }
    int rb0, rb1, rstep;
    if (a.map_mode == 0) {
        const int slot = xcd_slot(blockIdx.x, G);
        rb0 = (int)(((int64_t)slot * nb) / G);
        rb1 = (int)(((int64_t)(slot + 1) * nb) / G);
        rstep = 1;
    } else {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        const int X0 = (int)(((int64_t)xcd * nb) / 8), X1 = (int)(((int64_t)(xcd + 1) * nb) / 8);
        rb0 = X0 + jx; rb1 = X1; rstep = G >> 3;
    }
    const int niter = rb0 < rb1 ? (rb1 - rb0 + rstep - 1) / rstep : 0;

    double lane_sum = 0.0, lane_max = 0.0;
    double pend_val = 0.0;
    int pend_row = -1;

    // LDS reads of row block d (committed to `buf`) and its gathers + epilogue loads into gather set Gs
#define S5_LOADGATHER(Gs, d, buf)                                                         \
    do {                                                                                  \
        Gs.nr = (d).y & 0xffff; Gs.row = (d).x + t;                                       \
        Gs.j = 0; Gs.j1 = 0; Gs.e0 = 0.0; Gs.e1 = 0.0; Gs.e2 = 0.0;                       \
        if (t < Gs.nr) {                                                                  \
            const int base_ = (d).z & ~3;                                                 \
            Gs.j = s_rp[buf][t] - base_; Gs.j1 = s_rp[buf][t + 1] - base_;                \
            epi_operands<EPI, DOT>(a, Gs.row, Gs.e0, Gs.e1, Gs.e2);                       \
        }                                                                                 \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                   \
            const bool ok_ = Gs.j + u < Gs.j1;                                            \
            const int jj_ = ok_ ? Gs.j + u : 0;                                           \
            const int cc_ = s_col[buf][jj_];                                              \
            Gs.c[u] = ok_ ? (IDX ? Gs.row + cc_ : cc_) : 0; Gs.v[u] = s_val[buf][jj_];    \
        }                                                                                 \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) Gs.x[u] = a.x[Gs.c[u]];             \
    } while (0)
    // the in-order chain of the row block whose gathers sit in Gs, its epilogue, result parked
#define S5_FINISH(Gs)                                                                     \
    do {                                                                                  \
        double acc = 0.0, acc2 = t < Gs.nr ? chain2_start<EPI>(a, Gs.row) : 0.0, diag = 0.0; \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                     \
            if (Gs.j + u < Gs.j1) chain_step<EPI>(Gs.c[u], Gs.row, Gs.v[u], Gs.x[u], acc, acc2, diag); \
        if (t < Gs.nr) {                                                                  \
            const double outv = row_finish<EPI, DOT>(acc, acc2, diag, Gs.e0, Gs.e1, Gs.e2, lane_sum, lane_max); \
            if (a.y) { pend_val = outv; pend_row = Gs.row; }                              \
            sweep_note<EPI>(a, Gs.row, acc);                                              \
        }                                                                                 \
    } while (0)
    // one trip: R = register set of row block it+1, GN = its gather set, GO = gather set of row block it
#define S5_STEP(R, GN, GO, buf)                                                           \
    {                                                                                     \
        const int4 dD = S4_DESC(it + 3);                                                  \
        S4_COMMIT(R, dB, buf);                                                            \
        __syncthreads();                                                                  \
        if (pend_row >= 0) { a.y[pend_row] = pend_val; pend_row = -1; }                   \
        S5_LOADGATHER(GN, dB, buf);                                                       \
        S4_ISSUE(R, dD);                                                                  \
        __builtin_amdgcn_sched_barrier(0);   /* the chain below must not move above the loads just issued */ \
        S5_FINISH(GO);                                                                    \
        dB = dC; dC = dD;                                                                 \
        ++it;                                                                             \
    }

    if (niter > 0) {
        StageRegs P, Q;
        GatherRegs GA, GB;
        const int4 dA = S4_DESC(0);
        int4 dB = S4_DESC(1), dC = S4_DESC(2);
        int it = 0;
        S4_ISSUE(P, dA);
        S4_ISSUE(Q, dB);
        S4_COMMIT(P, dA, 0);
        __syncthreads();
        S5_LOADGATHER(GA, dA, 0);
        S4_ISSUE(P, dC);
        // trips come in pairs (block it+1 odd -> set Q, LDS buffer 1, gather set B; then even -> P, 0, A)
        const int niter2 = (niter + 1) & ~1;
        while (it < niter2) {
            S5_STEP(Q, GB, GA, 1)
            S5_STEP(P, GA, GB, 0)
        }
        if (pend_row >= 0) a.y[pend_row] = pend_val;
    }
#undef S5_LOADGATHER
#undef S5_FINISH
#undef S5_STEP
    if (DOT || EpiKind<EPI>::norm) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) {
            if (a.part_accum) a.part_sum[blockIdx.x] += bs;
            else {
                a.part_sum[blockIdx.x] = bs;
                for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_sum[z] = 0.0;
            }
        }
        if (EpiKind<EPI>::norm && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (t == 0) {
                if (a.part_accum) a.part_max[blockIdx.x] = fmax(a.part_max[blockIdx.x], bm);
                else {
                    a.part_max[blockIdx.x] = bm;
                    for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_max[z] = 0.0;
                }
            }
        }
    }
}
#undef S4_DESC
#undef S4_ISSUE
#undef S4_DECODE
#undef S4_VDEC0
#undef S4_VDEC1
#undef S4_COMMIT
#undef S4_BATCH_LOAD
#undef S4_BATCH_FMA
#undef S4_BODY

// --------------------------------------------------------------------------------------
// Exact SpMV for IRREGULAR row lengths (power-law matrices): gather and arithmetic are split.
//   phase A: all 256 lanes stream the row block's col/val slice (16 B per lane, coalesced) and
//            gather x[col] for the entries THEY loaded — perfectly parallel whatever the rows
//            look like — and park val, x[col] (and col) in LDS;
//   phase B: one lane per row runs the in-order multiply-add chain out of LDS only, so a long
//            row costs LDS latency per entry, not a memory round trip per 8 entries.
// Same arithmetic as spmv_crs_cpu (kernels.hpp:380-409): bit-exact.  The next row block's slice
// is prefetched into registers before phase B.  Chosen at launch when the matrix's longest row
// exceeds IRR_ROW_LEN (CsrDev::max_row_len); rows longer than the LDS capacity are streamed in
// chunks with lane 0 carrying the chain across chunks.
// --------------------------------------------------------------------------------------
// 5 workgroups per CU (LDS footprint) = 5 waves per SIMD: keep the register count within 512 / 5
template <int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK, 5) void spmv_irr_kernel(
    int nb, const int4 *__restrict__ blk_desc, const int *__restrict__ rp,
    const int *__restrict__ col, const double *__restrict__ val, int n_rows, SpmvArgs a)
{
    __shared__ __attribute__((aligned(16))) double s_val[S2_CAP];
    __shared__ __attribute__((aligned(16))) double s_xg[S2_CAP];
    __shared__ __attribute__((aligned(16))) int s_col[S2_CAP];
    __shared__ int s_rp[ITS_BLOCK + 4];
    __shared__ double s_red[4];

    const int t = threadIdx.x;
    const int G = gridDim.x;
    int rb0, rb1, rstep;
    if (a.map_mode == 0) {
        const int slot = xcd_slot(blockIdx.x, G);
        rb0 = (int)(((int64_t)slot * nb) / G);
        rb1 = (int)(((int64_t)(slot + 1) * nb) / G);
        rstep = 1;
    } else {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        const int X0 = (int)(((int64_t)xcd * nb) / 8), X1 = (int)(((int64_t)(xcd + 1) * nb) / 8);
        rb0 = X0 + jx; rb1 = X1; rstep = G >> 3;
    }
    double lane_sum = 0.0, lane_max = 0.0;

    // staging registers of the slice in flight (two sub-slices of 4 entries per lane)
    int4 rcA = make_int4(0, 0, 0, 0), rcB = rcA;
    double2 rvA0 = make_double2(0, 0), rvA1 = rvA0, rvB0 = rvA0, rvB1 = rvA0;
    bool have = false;

#define IRR_ISSUE(c1v, basev)                                                             \
    do {                                                                                  \
        const int iA_ = (basev) + 4 * t, iB_ = iA_ + 4 * ITS_BLOCK;                       \
        if (iA_ < (c1v)) {                                                                \
            rcA = ld_i4<0>(col + iA_); rvA0 = ld_d2<0>(val + iA_); rvA1 = ld_d2<0>(val + iA_ + 2); \
        }                                                                                 \
        if (iB_ < (c1v)) {                                                                \
            rcB = ld_i4<0>(col + iB_); rvB0 = ld_d2<0>(val + iB_); rvB1 = ld_d2<0>(val + iB_ + 2); \
        }                                                                                 \
    } while (0)
    // gather x for the staged entries and park everything in LDS.  Staged columns are real matrix
    // columns or the zero padding behind the arrays, so every gather address is inside x.
#define IRR_PARK(c1v, basev)                                                              \
    do {                                                                                  \
        const int oA_ = 4 * t, oB_ = oA_ + 4 * ITS_BLOCK;                                 \
        const bool okA_ = (basev) + oA_ < (c1v), okB_ = (basev) + oB_ < (c1v);            \
        double gA0 = 0, gA1 = 0, gA2 = 0, gA3 = 0, gB0 = 0, gB1 = 0, gB2 = 0, gB3 = 0;    \
        if (okA_) { gA0 = a.x[rcA.x]; gA1 = a.x[rcA.y]; gA2 = a.x[rcA.z]; gA3 = a.x[rcA.w]; } \
        if (okB_) { gB0 = a.x[rcB.x]; gB1 = a.x[rcB.y]; gB2 = a.x[rcB.z]; gB3 = a.x[rcB.w]; } \
        if (okA_) {                                                                       \
            *reinterpret_cast<int4 *>(&s_col[oA_]) = rcA;                                 \
            *reinterpret_cast<double2 *>(&s_val[oA_]) = rvA0;                             \
            *reinterpret_cast<double2 *>(&s_val[oA_ + 2]) = rvA1;                         \
            *reinterpret_cast<double2 *>(&s_xg[oA_]) = make_double2(gA0, gA1);            \
            *reinterpret_cast<double2 *>(&s_xg[oA_ + 2]) = make_double2(gA2, gA3);        \
        }                                                                                 \
        if (okB_) {                                                                       \
            *reinterpret_cast<int4 *>(&s_col[oB_]) = rcB;                                 \
            *reinterpret_cast<double2 *>(&s_val[oB_]) = rvB0;                             \
            *reinterpret_cast<double2 *>(&s_val[oB_ + 2]) = rvB1;                         \
            *reinterpret_cast<double2 *>(&s_xg[oB_]) = make_double2(gB0, gB1);            \
            *reinterpret_cast<double2 *>(&s_xg[oB_ + 2]) = make_double2(gB2, gB3);        \
        }                                                                                 \
    } while (0)
    // the chain: LDS reads of 8 entries go out together, then the 8 multiply-adds run in CRS order
#define IRR_CHAIN(jb, je)                                                                 \
    {                                                                                     \
        int j = (jb);                                                                     \
        const int je_ = (je);                                                             \
        for (; j + 8 <= je_; j += 8) {                                                    \
            double vv[8], xx[8]; int cc[8];                                               \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                               \
                vv[u] = s_val[j + u]; xx[u] = s_xg[j + u];                                \
                cc[u] = EpiKind<EPI>::needs_col ? s_col[j + u] : 0;                       \
            }                                                                             \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                 \
                chain_step<EPI>(cc[u], row, vv[u], xx[u], acc, acc2, diag);               \
        }                                                                                 \
        for (; j < je_; ++j)                                                              \
            chain_step<EPI>(EpiKind<EPI>::needs_col ? s_col[j] : 0, row, s_val[j], s_xg[j], acc, acc2, diag); \
    }

    for (int rb = rb0; rb < rb1; rb += rstep) {
        const int4 d = blk_desc[rb];
        const int nr = d.y & 0xffff;
        const int row = d.x + t;
        const int rbn = rb + rstep;
        int4 dn = make_int4(0, 0, 0, 0);
        if (rbn < rb1) dn = blk_desc[rbn];
        double acc = 0.0, acc2 = t < nr ? chain2_start<EPI>(a, row) : 0.0, diag = 0.0;
        int my_rp = 0;
        if (t < nr) my_rp = rp[d.x + t];
        for (int c0 = d.z; ; c0 += SPMV_NNZ_MAX) {
            const int c1 = min(d.w, c0 + SPMV_NNZ_MAX);
            const int base = c0 & ~3;
            if (!have) IRR_ISSUE(c1, base);
            have = false;
            __syncthreads();                       // phase B of the previous chunk / block is done with LDS
            IRR_PARK(c1, base);
            if (c0 == d.z) { if (t < nr) s_rp[t] = my_rp; if (t == 0) s_rp[nr] = d.w; }
            __syncthreads();
            // prefetch: the next chunk of this long row, else the first chunk of the next row block
            if (c1 < d.w) { IRR_ISSUE(min(d.w, c1 + SPMV_NNZ_MAX), c1 & ~3); have = true; }
            else if (rbn < rb1) { IRR_ISSUE(min(dn.w, dn.z + SPMV_NNZ_MAX), dn.z & ~3); have = true; }
            if (d.w - d.z <= SPMV_NNZ_MAX) {
                if (t < nr) { IRR_CHAIN(s_rp[t] - base, s_rp[t + 1] - base) }
            } else if (t == 0) {
                IRR_CHAIN(c0 - base, c1 - base)
            }
            if (c1 >= d.w) break;
        }
        if (t < nr) {
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
            epi_operands<EPI, DOT>(a, row, e0, e1, e2);
            const double outv = row_finish<EPI, DOT>(acc, acc2, diag, e0, e1, e2, lane_sum, lane_max);
            if (a.y) a.y[row] = outv;                        // y == nullptr: only the norm is wanted
            sweep_note<EPI>(a, row, acc);
        }
    }
#undef IRR_ISSUE
#undef IRR_PARK
#undef IRR_CHAIN
    if (DOT || EpiKind<EPI>::norm) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) {
            if (a.part_accum) a.part_sum[blockIdx.x] += bs;
            else {
                a.part_sum[blockIdx.x] = bs;
                for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_sum[z] = 0.0;
            }
        }
        if (EpiKind<EPI>::norm && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (t == 0) {
                if (a.part_accum) a.part_max[blockIdx.x] = fmax(a.part_max[blockIdx.x], bm);
                else {
                    a.part_max[blockIdx.x] = bm;
                    for (int z = blockIdx.x + G; z < ITS_NPART; z += G) a.part_max[z] = 0.0;
                }
            }
        }
    }
}

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// launch tunables of the exact SpMV (its_set_option "spmv_grid" / "spmv_map" / ...)
// defaults measured on MI355X, 256^3 7-pt: v4 with 2 workgroups per CU (0.369 ms) vs v2 with 4 (0.376 ms)
static int g_spmv_variant = env_int("ITS_SPMV_VARIANT", 4);
static int g_spmv_grid = env_int("ITS_SPMV_GRID", g_spmv_variant >= 4 ? 512 : 1024);
static int g_spmv_map = env_int("ITS_SPMV_MAP", 1);
static int g_spmv_idx = env_int("ITS_SPMV_IDX", 1);             // 0 = never use the 1-byte column-offset stream
static int g_spmv_valdict = env_int("ITS_SPMV_VALDICT", 0);     // opt-in: 1-byte value-dictionary stream
static int g_spmv_vd_grid = env_int("ITS_SPMV_VD_GRID", 768);   // its grid (3 workgroups per CU measured best)
static int g_spmv_irr = env_int("ITS_SPMV_IRR", 1);             // 0 = never use the irregular-row kernel
static int g_spmv_irr_grid = env_int("ITS_SPMV_IRR_GRID", 1280);
static int g_spmv_irr_split = env_int("ITS_SPMV_IRR_SPLIT", 1);     // rows beyond a row block on the wave-per-row launch
static int g_spmv_irr_len = env_int("ITS_SPMV_IRR_LEN", IRR_ROW_LEN);   // longest row above which it is used
static int g_spmv_bigplan = env_int("ITS_SPMV_BIGPLAN", 1);     // 0 = spmv_stream4_kernel on the ordinary plan too
static int g_mixed_ell = env_int("ITS_SPMV_MIXED_ELL", 1);       // 0 = the CRS split even where the tiered ELL image exists
// the opt-in value-dictionary stream (2 bytes per nonzero) moves fewer bytes than the ELL image: it keeps its kernel
bool spmv_valdict_selected(const CsrDev &A) { return g_spmv_valdict && g_spmv_idx && A.n_off > 0 && A.n_vals > 0 && A.max_row_len <= 8; }
int spmv_tune(const char *name, int value)
{
    if (!strcmp(name, "spmv_grid")) { if (value < 8 || value > ITS_NPART || (value & 7)) return -1; g_spmv_grid = value; return 0; }
    if (!strcmp(name, "spmv_map")) { if (value < 0 || value > 1) return -1; g_spmv_map = value; return 0; }
    if (!strcmp(name, "spmv_variant")) { if (value != 2 && value != 4 && value != 5) return -1; g_spmv_variant = value; return 0; }
    if (!strcmp(name, "spmv_irr_split")) { if (value < 0 || value > 1) return -1; g_spmv_irr_split = value; return 0; }
    if (!strcmp(name, "spmv_idx")) { if (value < 0 || value > 1) return -1; g_spmv_idx = value; return 0; }
    if (!strcmp(name, "spmv_valdict")) { if (value < 0 || value > 1) return -1; g_spmv_valdict = value; return 0; }
    if (!strcmp(name, "spmv_irr")) { if (value < 0 || value > 1) return -1; g_spmv_irr = value; return 0; }
    if (!strcmp(name, "spmv_irr_len")) { if (value < 1) return -1; g_spmv_irr_len = value; return 0; }
    if (!strcmp(name, "spmv_irr_grid")) { if (value < 8 || value > ITS_NPART || (value & 7)) return -1; g_spmv_irr_grid = value; return 0; }
    if (!strcmp(name, "spmv_bigplan")) { if (value < 0 || value > 1) return -1; g_spmv_bigplan = value; return 0; }
    if (!strcmp(name, "spmv_mixed_ell")) { if (value < 0 || value > 1) return -1; g_mixed_ell = value; return 0; }
    return -2;
}

void launch_spmv_exact_desc(const CsrDev &A, const int4 *desc, int n_desc, const SpmvArgs &a_in, int epi,
                            int accumulate, hipStream_t s)
{
    int g2v = g_spmv_grid;
    const int g2v_long = g_spmv_variant >= 4 ? 1024 : g2v;   // v2's own best grid (4 workgroups per CU)
    const int mm = g_spmv_map;
    if (g2v < 8 || g2v > ITS_NPART || (g2v & 7)) g2v = 1024;
    dim3 g2(g2v), b(ITS_BLOCK);
    SpmvArgs a = a_in;
    a.map_mode = mm;
    a.part_accum = accumulate;
#define ITS_LAUNCH2(E, D)                                                                                     \
    do {                                                                                                      \
        if (A.n_off > 0 && g_spmv_idx && A.blk_desc_big)                                                      \
            hipLaunchKernelGGL((spmv_stream2_kernel<E, D, 2>), dim3(g2v_long), b, 0, s, n_desc, desc, A.row_ptr, A.col, \
                               A.val, A.n_rows, A.cidx, A.off_dict, a);                                       \
        else if (A.n_off > 0 && g_spmv_idx)                                                                   \
            hipLaunchKernelGGL((spmv_stream2_kernel<E, D, 1>), dim3(g2v_long), b, 0, s, n_desc, desc, A.row_ptr, A.col, \
                               A.val, A.n_rows, A.cidx, A.off_dict, a);                                       \
        else                                                                                                  \
            hipLaunchKernelGGL((spmv_stream2_kernel<E, D, 0>), dim3(g2v_long), b, 0, s, n_desc, desc, A.row_ptr, A.col, \
                               A.val, A.n_rows, A.cidx, A.off_dict, a);                                       \
    } while (0)
    // irregular rows: the two-phase kernel (5 workgroups per CU by its LDS footprint)
    if (A.max_row_len > g_spmv_irr_len && g_spmv_irr) {
        dim3 gi(g_spmv_irr_grid);
#define ITS_LAUNCHI(E, D) \
    hipLaunchKernelGGL((spmv_irr_kernel<E, D>), gi, b, 0, s, n_desc, desc, A.row_ptr, A.col, A.val, A.n_rows, a)
        switch (epi) {
        case EPI_STORE: if (a.dotv) ITS_LAUNCHI(EPI_STORE, true); else ITS_LAUNCHI(EPI_STORE, false); break;
        case EPI_RESIDUAL: ITS_LAUNCHI(EPI_RESIDUAL, false); break;
        case EPI_JACOBI_SHIPPED: ITS_LAUNCHI(EPI_JACOBI_SHIPPED, false); break;
        case EPI_JACOBI_REF: ITS_LAUNCHI(EPI_JACOBI_REF, false); break;
        case EPI_GS_RHS: ITS_LAUNCHI(EPI_GS_RHS, false); break;
        case EPI_JACOBI_FUSED: ITS_LAUNCHI(EPI_JACOBI_FUSED, false); break;
        case EPI_GS_FUSED: ITS_LAUNCHI(EPI_GS_FUSED, false); break;
        case EPI_SWEEP: ITS_LAUNCHI(EPI_SWEEP, false); break;
        }
#undef ITS_LAUNCHI
        return;
    }
#define ITS_LAUNCH4(E, D)                                                                                     \
    do {                                                                                                      \
        if (A.n_off > 0 && g_spmv_idx && A.n_vals > 0 && g_spmv_valdict)                                      \
            hipLaunchKernelGGL((spmv_stream4_kernel<E, D, 2>), dim3(g_spmv_vd_grid), b, 0, s, n_desc, desc, A.row_ptr, A.col,   \
                               A.val, A.n_rows, A.cidx, A.off_dict, A.vidx, A.val_dict, a);                   \
        else if (A.n_off > 0 && g_spmv_idx)                                                                   \
            hipLaunchKernelGGL((spmv_stream4_kernel<E, D, 1>), g2, b, 0, s, n_desc, desc, A.row_ptr, A.col,   \
                               A.val, A.n_rows, A.cidx, A.off_dict, A.vidx, A.val_dict, a);                   \
        else                                                                                                  \
            hipLaunchKernelGGL((spmv_stream4_kernel<E, D, 0>), g2, b, 0, s, n_desc, desc, A.row_ptr, A.col,   \
                               A.val, A.n_rows, A.cidx, A.off_dict, A.vidx, A.val_dict, a);                   \
    } while (0)
    // v4 keeps one batch of 8 gathers per row in flight; rows of 9..32 entries (27-pt) run faster in v2,
    // whose per-lane batch loop does not wait for the longest row of the block (0.169 vs 0.213 ms, 27-pt 128^3)
    // v5 (gathers one trip ahead of their multiply-adds) wins only once the streams are tiny: it is used for the
    // opt-in value-dictionary mode (0.183 vs 0.205 ms) and on request (variant 5); otherwise v4 (0.315 vs 0.320 ms)
    const int cmp = (A.n_off > 0 && g_spmv_idx) ? ((A.n_vals > 0 && g_spmv_valdict) ? 2 : 1) : 0;
    if (A.max_row_len <= 8 && (g_spmv_variant == 5 || (g_spmv_variant == 4 && cmp == 2))) {
        if (cmp == 2) g2 = dim3(g_spmv_vd_grid);
#define ITS_LAUNCH5C(E, D, C)                                                                                 \
    hipLaunchKernelGGL((spmv_stream5_kernel<E, D, C>), g2, b, 0, s, n_desc, desc, A.row_ptr, A.col, A.val,    \
                       A.n_rows, A.cidx, A.off_dict, A.vidx, A.val_dict, a)
#define ITS_LAUNCH5(E, D) do { if (cmp == 2) ITS_LAUNCH5C(E, D, 2); else if (cmp == 1) ITS_LAUNCH5C(E, D, 1); else ITS_LAUNCH5C(E, D, 0); } while (0)
        switch (epi) {
        case EPI_STORE: if (a.dotv) ITS_LAUNCH5(EPI_STORE, true); else ITS_LAUNCH5(EPI_STORE, false); break;
        case EPI_RESIDUAL: ITS_LAUNCH5(EPI_RESIDUAL, false); break;
        case EPI_JACOBI_SHIPPED: ITS_LAUNCH5(EPI_JACOBI_SHIPPED, false); break;
        case EPI_JACOBI_REF: ITS_LAUNCH5(EPI_JACOBI_REF, false); break;
        case EPI_GS_RHS: ITS_LAUNCH5(EPI_GS_RHS, false); break;
        case EPI_JACOBI_FUSED: ITS_LAUNCH5(EPI_JACOBI_FUSED, false); break;
        case EPI_GS_FUSED: ITS_LAUNCH5(EPI_GS_FUSED, false); break;
        case EPI_SWEEP: ITS_LAUNCH5(EPI_SWEEP, false); break;
        }
#undef ITS_LAUNCH5
#undef ITS_LAUNCH5C
        return;
    }
    // measured (256^3 7-pt, offset stream): v4 wins inside the CG loop (0.511 vs 0.522 ms per trip), v2 wins for
    // every other epilogue (plain 0.303 vs 0.316 ms, Jacobi trip 0.620 vs 0.672 ms, norm-only residual 0.245 vs 0.299)
    if (g_spmv_variant == 4 && A.max_row_len <= 8 && epi == EPI_STORE && a.dotv) {
        ITS_LAUNCH4(EPI_STORE, true);
        return;
    }
#undef ITS_LAUNCH4
    switch (epi) {
    case EPI_STORE: if (a.dotv) ITS_LAUNCH2(EPI_STORE, true); else ITS_LAUNCH2(EPI_STORE, false); break;
    case EPI_RESIDUAL: ITS_LAUNCH2(EPI_RESIDUAL, false); break;
    case EPI_JACOBI_SHIPPED: ITS_LAUNCH2(EPI_JACOBI_SHIPPED, false); break;
    case EPI_JACOBI_REF: ITS_LAUNCH2(EPI_JACOBI_REF, false); break;
    case EPI_GS_RHS: ITS_LAUNCH2(EPI_GS_RHS, false); break;
    case EPI_JACOBI_FUSED: ITS_LAUNCH2(EPI_JACOBI_FUSED, false); break;
    case EPI_GS_FUSED: ITS_LAUNCH2(EPI_GS_FUSED, false); break;
    case EPI_SWEEP: ITS_LAUNCH2(EPI_SWEEP, false); break;
    }
#undef ITS_LAUNCH2
}

// the CG form on spmv_stream4_kernel walks the plan with the larger row blocks (same test as the dispatch above)
bool spmv_exact_takes_big_plan(const CsrDev &A, const SpmvArgs &a, int epi)
{
    if (!g_spmv_bigplan || !A.blk_desc_big || (A.max_row_len > g_spmv_irr_len && g_spmv_irr) || g_spmv_variant == 5) return false;
    const int cmp = (A.n_off > 0 && g_spmv_idx) ? ((A.n_vals > 0 && g_spmv_valdict) ? 2 : 1) : 0;
    if (cmp == 2) return false;                                  // spmv_stream5_kernel: small LDS buffers
    if (g_spmv_variant == 4 && A.max_row_len <= 8 && epi == EPI_STORE && a.dotv) return true;   // spmv_stream4_kernel
    return cmp == 1;                                             // spmv_stream2_kernel<.., IDX = 2>
}

// which kernel family a whole-matrix launch of the exact SpMV takes (same tests as the dispatch), and the bytes
// it is asked to stream per launch (matrix arrays in the form that kernel reads + x once + y once)
const char *spmv_exact_kernel_name(const CsrDev &A, const SpmvArgs &a, int epi, int64_t *streamed)
{
    const int64_t n = A.n_rows, nnz = A.nnz;
    const int cmp = (A.n_off > 0 && g_spmv_idx) ? ((A.n_vals > 0 && g_spmv_valdict) ? 2 : 1) : 0;
    const int64_t csr = (cmp == 2 ? 2 : cmp == 1 ? 9 : 12) * nnz + 4 * (n + 1) + 16 * n;
    if (streamed) *streamed = csr;
    if (spmv_ell_eligible(A, a, epi)) {
        if (A.ell_kind == 3) {
            const int64_t slices = (int64_t)A.ell_blocks * (ELL_RB / ELL_S);
            if (streamed) *streamed = (int64_t)A.ell_blocks * ELL_RB * 8 * A.ell_w + 4 * A.ellg_idx_words + slices * (4 + 4 * A.ell_w) + 16 * n;
            return "spmv_ellg_kernel";
        }
        if (A.ell_kind == 2) {
            if (streamed) *streamed = (int64_t)A.ell_blocks * ELL_RB * (8 * A.ell_w + 4 * ((A.ell_w + 3) / 4)) + 16 * n;
            return "spmv_ellw_kernel";
        }
        if (streamed) *streamed = (int64_t)A.ell_blocks * ELL_RB * (8 * A.ell_w + 4) + 16 * n;
        return "spmv_ell_kernel";
    }
    if (spmv_panel_eligible(A, a, epi)) {
        // image entries at 12 bytes, chunk descriptors, the long rows in CRS form, the long-row flags, x and y once
        if (streamed) *streamed = 12 * A.pan_nnz + 8 * (int64_t)A.pan_chunks + 12 * (nnz - A.pan_nnz) + 12 * (int64_t)A.pan_n_long + n + 16 * n;
        return "spmv_panel_kernel";
    }
    if (A.max_row_len > g_spmv_irr_len && g_spmv_irr) { if (streamed) *streamed = 12 * nnz + 4 * (n + 1) + 16 * n; return "spmv_irr_kernel"; }
    if (A.max_row_len <= 8 && (g_spmv_variant == 5 || (g_spmv_variant == 4 && cmp == 2))) return "spmv_stream5_kernel";
    if (g_spmv_variant == 4 && A.max_row_len <= 8 && epi == EPI_STORE && a.dotv) return "spmv_stream4_kernel";
    return "spmv_stream2_kernel";
}

void launch_spmv_exact(const CsrDev &A, const SpmvArgs &a_in, int epi, hipStream_t s)
{
    if (spmv_ell_eligible(A, a_in, epi)) { launch_spmv_ell(A, a_in, epi, s); return; }
    if (spmv_panel_eligible(A, a_in, epi)) { launch_spmv_panel(A, a_in, epi, s); return; }
    // a row that does not fit a row block is walked by one lane, chunk by chunk between workgroup barriers: 16 ns per
    // entry, and a 6000-entry hub row IS the product of a 20 000-row power-law matrix (0.105 ms).  Such rows go to the
    // wave-per-row launch instead (spmv_long_kernel), the plan without them to the kernel below.
    if (g_spmv_irr_split && A.irr_n_long > 0 && epi != EPI_SWEEP && a_in.row_end < 0) {
        SpmvArgs a = a_in;
        launch_spmv_exact_desc(A, A.blk_desc_short, A.n_blocks_short, a, epi, 0, s);
        launch_spmv_long(A, A.irr_long_rows, A.irr_n_long, a, epi, s);
        return;
    }
    if (spmv_exact_takes_big_plan(A, a_in, epi)) launch_spmv_exact_desc(A, A.blk_desc_big, A.n_blocks_big, a_in, epi, 0, s);
    else launch_spmv_exact_desc(A, A.blk_desc, A.n_blocks, a_in, epi, 0, s);
}

// ======================================================================================
// SpMV, vector-CSR: LPR = 2^k lanes cooperate on one row (chosen from the row-length
// histogram), partial sums combined with a butterfly inside the lane group.  Direct
// coalesced loads, no LDS.  Reordered summation: 1e-12 relative to the exact variant.
// ======================================================================================
// Round 2: with short rows (LPR <= 8) a lane works on two rows per trip (row groups grp, grp + 1) and has the first two
// LPR-strided entries of each in flight before the first gather — four (index, value) loads, then four gathers, instead of one dependent
// chain of three round trips per entry — and streams the 1-byte column-offset index (CsrDev::cidx, dictionary in LDS)
// instead of the 4-byte column where the matrix has one: 9 bytes per entry.  Every lane still adds its entries in
// ascending j, so the values are those of the one-row-at-a-time loop.
template <int LPR, int EPI, bool DOT, bool IDX>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_vector_kernel(
    int n_rows, const int *__restrict__ rp, const int *__restrict__ col, const unsigned char *__restrict__ cidx,
    const int *__restrict__ dict, const double *__restrict__ val, SpmvArgs a)
{
    __shared__ double s_red[4];
    __shared__ int s_dict[256];
    constexpr int ROWS_PER_BLOCK = ITS_BLOCK / LPR;
    const int t = threadIdx.x;
    if (IDX) { s_dict[t] = dict[t]; __syncthreads(); }
    const int sub = t & (LPR - 1);
    const int G = gridDim.x;
    const int slot = xcd_slot(blockIdx.x, G);
    const int64_t n_groups = ((int64_t)n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    const int64_t g0 = (slot * n_groups) / G, g1 = ((slot + 1) * n_groups) / G;
    double lane_sum = 0.0, lane_max = 0.0;
    auto column = [&](int j, int row) { return IDX ? row + s_dict[cidx[j]] : col[j]; };
    auto finish = [&](int row, bool ok, double acc) {
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, ITS_WAVE);
        if (ok && sub == 0) {
            if (EPI == EPI_STORE) {
                a.y[row] = acc;
                if (DOT) lane_sum = fma(acc, a.dotv[row], lane_sum);
            } else if (EPI == EPI_RESIDUAL) {
                const double r = a.b[row] - acc;
                if (a.y) a.y[row] = r;                       // y == nullptr: only the norm is wanted
                lane_sum = fma(r, r, lane_sum);
                lane_max = fmax(lane_max, fabs(r));
            }
        }
    };
    constexpr int R = LPR <= 8 ? 2 : 1;         // rows per lane per trip (4: 0.383 ms on the 7-pt 256^3 matrix, 2: 0.372, 1: 0.45)
    for (int64_t grp = g0; grp < g1; grp += R) {
        int row[R], j0[R], j1[R], c0[R], c1[R]; bool ok[R];
        double v0[R], v1[R], x0[R], x1[R], acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            row[r] = (int)((grp + r) * ROWS_PER_BLOCK) + t / LPR;
            ok[r] = grp + r < g1 && row[r] < n_rows;
            j0[r] = j1[r] = 0;
            if (ok[r]) { j0[r] = rp[row[r]] + sub; j1[r] = rp[row[r] + 1]; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            c0[r] = c1[r] = 0; v0[r] = v1[r] = 0.0;
            if (j0[r] < j1[r]) { c0[r] = column(j0[r], row[r]); v0[r] = val[j0[r]]; }
            if (j0[r] + LPR < j1[r]) { c1[r] = column(j0[r] + LPR, row[r]); v1[r] = val[j0[r] + LPR]; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            x0[r] = j0[r] < j1[r] ? a.x[c0[r]] : 0.0;
            x1[r] = j0[r] + LPR < j1[r] ? a.x[c1[r]] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = 0.0;
            if (j0[r] < j1[r]) acc[r] = fma(v0[r], x0[r], acc[r]);
            if (j0[r] + LPR < j1[r]) acc[r] = fma(v1[r], x1[r], acc[r]);
            for (int j = j0[r] + 2 * LPR; j < j1[r]; j += LPR) acc[r] = fma(val[j], a.x[column(j, row[r])], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) finish(row[r], ok[r], acc[r]);
    }
    if (DOT || EPI == EPI_RESIDUAL) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) a.part_sum[blockIdx.x] = bs;
        if (EPI == EPI_RESIDUAL && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (t == 0) a.part_max[blockIdx.x] = bm;
        }
    }
}

template <int LPR, bool IDX>
static void launch_vec_lpr(const CsrDev &A, const SpmvArgs &a, int epi, hipStream_t s)
{
    dim3 g(ITS_NPART), b(ITS_BLOCK);
    if (epi == EPI_RESIDUAL)
        hipLaunchKernelGGL((spmv_vector_kernel<LPR, EPI_RESIDUAL, false, IDX>), g, b, 0, s, A.n_rows, A.row_ptr, A.col, A.cidx, A.off_dict, A.val, a);
    else if (a.dotv)
        hipLaunchKernelGGL((spmv_vector_kernel<LPR, EPI_STORE, true, IDX>), g, b, 0, s, A.n_rows, A.row_ptr, A.col, A.cidx, A.off_dict, A.val, a);
    else
        hipLaunchKernelGGL((spmv_vector_kernel<LPR, EPI_STORE, false, IDX>), g, b, 0, s, A.n_rows, A.row_ptr, A.col, A.cidx, A.off_dict, A.val, a);
}
template <int LPR>
static void launch_vec_lpr(const CsrDev &A, const SpmvArgs &a, int epi, hipStream_t s)
{
    if (A.n_off > 0 && A.cidx && g_spmv_idx) launch_vec_lpr<LPR, true>(A, a, epi, s);
    else launch_vec_lpr<LPR, false>(A, a, epi, s);
}

void launch_spmv_vector(const CsrDev &A, const SpmvArgs &a, int epi, hipStream_t s)
{
    switch (A.lanes_per_row) {
    case 2: launch_vec_lpr<2>(A, a, epi, s); break;
    case 4: launch_vec_lpr<4>(A, a, epi, s); break;
    case 8: launch_vec_lpr<8>(A, a, epi, s); break;
    case 16: launch_vec_lpr<16>(A, a, epi, s); break;
    case 32: launch_vec_lpr<32>(A, a, epi, s); break;
    default: launch_vec_lpr<64>(A, a, epi, s); break;
    }
}

// ======================================================================================
// SpMV, mixed precision storage (ITS_FMT_MIXED; the reference's USE_AP "ap[dp_sp]" idea, main.cpp:171-184): the
// entries at or above the threshold as an fp64 CRS, the rest as an fp32 CRS (8 instead of 12 bytes per entry);
// x, y and the accumulation stay fp64.  LPR lanes per row as in the vector kernel.  Not bit-exact by design.
// ======================================================================================
template <int LPR, int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_mixed_kernel(
    int n_rows, const int *__restrict__ rp_d, const int *__restrict__ col_d, const double *__restrict__ val_d,
    const int *__restrict__ rp_s, const int *__restrict__ col_s, const float *__restrict__ val_s,
    const int *__restrict__ rp_h, const int *__restrict__ col_h, const _Float16 *__restrict__ val_h, SpmvArgs a)
{
    __shared__ double s_red[4];
    constexpr int ROWS_PER_BLOCK = ITS_BLOCK / LPR;
    const int t = threadIdx.x;
    const int sub = t & (LPR - 1);
    const int G = gridDim.x;
    const int slot = xcd_slot(blockIdx.x, G);
    const int64_t n_groups = ((int64_t)n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    const int64_t g0 = (slot * n_groups) / G, g1 = ((slot + 1) * n_groups) / G;
    double lane_sum = 0.0, lane_max = 0.0;
    for (int64_t grp = g0; grp < g1; ++grp) {
        const int row = (int)(grp * ROWS_PER_BLOCK) + t / LPR;
        double acc = 0.0;
        if (row < n_rows) {
            for (int j = rp_d[row] + sub, j1 = rp_d[row + 1]; j < j1; j += LPR) acc = fma(val_d[j], a.x[col_d[j]], acc);
            for (int j = rp_s[row] + sub, j1 = rp_s[row + 1]; j < j1; j += LPR) acc = fma((double)val_s[j], a.x[col_s[j]], acc);
            if (rp_h) for (int j = rp_h[row] + sub, j1 = rp_h[row + 1]; j < j1; j += LPR) acc = fma((double)(float)val_h[j], a.x[col_h[j]], acc);
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, ITS_WAVE);
        if (row < n_rows && sub == 0) {
            if (EPI == EPI_STORE) {
                a.y[row] = acc;
                if (DOT) lane_sum = fma(acc, a.dotv[row], lane_sum);
            } else {
                const double r = a.b[row] - acc;
                if (a.y) a.y[row] = r;
                lane_sum = fma(r, r, lane_sum);
                lane_max = fmax(lane_max, fabs(r));
            }
        }
    }
    if (DOT || EPI == EPI_RESIDUAL) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) a.part_sum[blockIdx.x] = bs;
        if (EPI == EPI_RESIDUAL && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (t == 0) a.part_max[blockIdx.x] = bm;
        }
    }
}

template <int LPR>
static void launch_mixed_lpr(const CsrDev &A, const SpmvArgs &a, int epi, hipStream_t s)
{
    dim3 g(ITS_NPART), b(ITS_BLOCK);
#define ITS_LAUNCHM(E, D) \
    hipLaunchKernelGGL((spmv_mixed_kernel<LPR, E, D>), g, b, 0, s, A.n_rows, A.mix_rp_d, A.mix_col_d, A.mix_val_d, A.mix_rp_s, \
                       A.mix_col_s, A.mix_val_s, A.mix_nnz_h ? A.mix_rp_h : nullptr, A.mix_col_h, reinterpret_cast<const _Float16 *>(A.mix_val_h), a)
    if (epi == EPI_RESIDUAL) ITS_LAUNCHM(EPI_RESIDUAL, false);
    else if (a.dotv) ITS_LAUNCHM(EPI_STORE, true);
    else ITS_LAUNCHM(EPI_STORE, false);
#undef ITS_LAUNCHM
}

void launch_spmv_mixed(const CsrDev &A, const SpmvArgs &a, int epi, hipStream_t s)
{
    if (g_mixed_ell && spmv_ellm_eligible(A, a, epi)) { launch_spmv_ellm(A, a, epi, s); return; }
    switch (A.lanes_per_row) {
    case 2: launch_mixed_lpr<2>(A, a, epi, s); break;
    case 4: launch_mixed_lpr<4>(A, a, epi, s); break;
    case 8: launch_mixed_lpr<8>(A, a, epi, s); break;
    case 16: launch_mixed_lpr<16>(A, a, epi, s); break;
    case 32: launch_mixed_lpr<32>(A, a, epi, s); break;
    default: launch_mixed_lpr<64>(A, a, epi, s); break;
    }
}

// ======================================================================================
// SpMV, SELL-64-sigma: one wavefront per chunk of 64 (sorted) rows, column-major inside the
// chunk (values[chunk_ptr + j*64 + lane], CUDA_tests/helpers/utilities.cpp:321-337), so every
// load is one fully coalesced 512-byte / 256-byte wave access.  perm[] maps chunk row -> row.
// ======================================================================================
constexpr int SELL_COOP = 512;   // chunks wider than this are shared by the four waves of the workgroup

template <int EPI, bool DOT>
__global__ __launch_bounds__(ITS_BLOCK) void spmv_sell_kernel(
    int n_rows, int n_chunks, const long long *__restrict__ chunk_ptr, const int *__restrict__ chunk_len,
    const int *__restrict__ scol, const double *__restrict__ sval, const int *__restrict__ perm,
    const int *__restrict__ grp_start, SpmvArgs a)
{
    __shared__ double s_red[4];
    __shared__ double s_part[4][ITS_WAVE];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int slot = xcd_slot(blockIdx.x, gridDim.x);
    const int g0 = grp_start[slot], g1 = grp_start[slot + 1];
    double lane_sum = 0.0, lane_max = 0.0;
    // columns [ja, jb) of one chunk: loads of 8 columns go out together, multiply-adds in column order
#define SELL_RUN(cs, ja, jb, acc)                                                         \
    {                                                                                     \
        int j = (ja);                                                                     \
        for (; j + 8 <= (jb); j += 8) {                                                   \
            int c[8]; double v[8], xv[8];                                                 \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                               \
                const int64_t idx = (cs) + (int64_t)(j + u) * 64 + lane;                  \
                c[u] = scol[idx]; v[u] = sval[idx];                                       \
            }                                                                             \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) xv[u] = a.x[c[u]];              \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) acc = fma(v[u], xv[u], acc);    \
        }                                                                                 \
        if (j < (jb)) {        /* the last 1..7 columns as one guarded batch (not one dependent round trip each) */ \
            int c[8]; double v[8], xv[8];                                                 \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                               \
                const int64_t idx = (cs) + (int64_t)(j + u) * 64 + lane;                  \
                c[u] = 0; v[u] = 0.0;                                                     \
                if (j + u < (jb)) { c[u] = scol[idx]; v[u] = sval[idx]; }                 \
            }                                                                             \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) xv[u] = (j + u < (jb)) ? a.x[c[u]] : 0.0; \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) if (j + u < (jb)) acc = fma(v[u], xv[u], acc); \
        }                                                                                 \
    }
#define SELL_EPILOGUE(ch, acc)                                                            \
    {                                                                                     \
        const int row = perm[(ch) * 64 + lane];                                           \
        if (row >= 0 && row < n_rows) {                                                   \
            if (EPI == EPI_STORE) {                                                       \
                a.y[row] = acc;                                                           \
                if (DOT) lane_sum = fma(acc, a.dotv[row], lane_sum);                      \
            } else if (EPI == EPI_RESIDUAL) {                                             \
                const double r = a.b[row] - acc;                                          \
                if (a.y) a.y[row] = r;                                                    \
                lane_sum = fma(r, r, lane_sum);                                           \
                lane_max = fmax(lane_max, fabs(r));                                       \
            }                                                                             \
        }                                                                                 \
    }
    for (int grp = g0; grp < g1; ++grp) {
        for (int cc = 0; cc < 4; ++cc) {
            const int ch = grp * 4 + cc;
            if (ch >= n_chunks) break;
            const int len = chunk_len[ch];
            const int64_t cs = chunk_ptr[ch];
            if (len > SELL_COOP) {
                // wide chunk: wave w takes a quarter of the columns, partial sums combined in wave order
                const int q = (len + 3) >> 2;
                const int ja = min(len, w * q), jb = min(len, ja + q);
                double acc = 0.0;
                SELL_RUN(cs, ja, jb, acc)
                s_part[w][lane] = acc;
                __syncthreads();
                if (w == 0) {
                    acc = ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
                    SELL_EPILOGUE(ch, acc)
                }
                __syncthreads();
            } else if (cc == w) {
                double acc = 0.0;
                SELL_RUN(cs, 0, len, acc)
                SELL_EPILOGUE(ch, acc)
            }
        }
    }
#undef SELL_RUN
#undef SELL_EPILOGUE
    if (DOT || EPI == EPI_RESIDUAL) {
        const double bs = block_sum(lane_sum, s_red);
        if (t == 0 && a.part_sum) a.part_sum[blockIdx.x] = bs;
        if (EPI == EPI_RESIDUAL && a.part_max) {
            const double bm = block_max(lane_max, s_red);
            if (t == 0) a.part_max[blockIdx.x] = bm;
        }
    }
}

// The image's col / val arrays from the CRS arrays already on the device: a lane per stored row, entry j of row
// perm[c * 64 + r] to chunk_ptr[c] + j * 64 + r (column-major inside the chunk, so the 64 lanes of a wave write 64
// consecutive elements per step); the arrays were zeroed, which is the padding (col 0, val 0.0).
__global__ __launch_bounds__(ITS_BLOCK) void sell_fill_kernel(
    int n_rows, int n_chunks, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
    const long long *__restrict__ chunk_ptr, const int *__restrict__ perm, int *__restrict__ scol, double *__restrict__ sval)
{
    const int64_t i = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (i >= (int64_t)n_chunks * 64) return;
    const int row = perm[i];
    if (row < 0 || row >= n_rows) return;
    const int64_t base = chunk_ptr[i >> 6] + (i & 63);
    const int lo = rp[row], hi = rp[row + 1];
    for (int j = lo; j < hi; ++j) {
        const int64_t idx = base + (int64_t)(j - lo) * 64;
        scol[idx] = col[j]; sval[idx] = val[j];
    }
}

void launch_sell_fill(const CsrDev &A, hipStream_t s)
{
    const int64_t rows = (int64_t)A.sell_chunks * 64;
    if (rows == 0) return;
    hipLaunchKernelGGL(sell_fill_kernel, dim3((unsigned)((rows + ITS_BLOCK - 1) / ITS_BLOCK)), dim3(ITS_BLOCK), 0, s, A.n_rows, A.sell_chunks,
                       A.row_ptr, A.col, A.val, A.sell_chunk_ptr, A.sell_perm, A.sell_col, A.sell_val);
}

void launch_spmv_sell(const CsrDev &A, const SpmvArgs &a, int epi, hipStream_t s)
{
    dim3 g(ITS_NPART), b(ITS_BLOCK);
#define ITS_LAUNCH(E, D) \
    hipLaunchKernelGGL((spmv_sell_kernel<E, D>), g, b, 0, s, A.n_rows, A.sell_chunks, A.sell_chunk_ptr, \
                       A.sell_chunk_len, A.sell_col, A.sell_val, A.sell_perm, A.sell_grp_start, a)
    if (epi == EPI_RESIDUAL) ITS_LAUNCH(EPI_RESIDUAL, false);
    else if (a.dotv) ITS_LAUNCH(EPI_STORE, true);
    else ITS_LAUNCH(EPI_STORE, false);
#undef ITS_LAUNCH
}

// ======================================================================================
// BLAS-1.  Grid-stride over pairs (16 B per lane) when the operands are 16-byte aligned,
// scalar tail otherwise.  Reductions: thread sum -> wave butterfly -> 4-wave LDS sum ->
// partial[blockIdx]; the order is fixed, so results are reproducible run to run.
// ======================================================================================
__device__ __forceinline__ bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ __launch_bounds__(ITS_BLOCK) void dot_partials_kernel(
    const double *__restrict__ a, const double *__restrict__ b, int64_t n, double *__restrict__ part)
{
    __shared__ double s_red[4];
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    double s0 = 0.0, s1 = 0.0;
    int64_t n2 = (al16(a) && al16(b)) ? n / 2 : 0;
    const double2 *a2 = reinterpret_cast<const double2 *>(a), *b2 = reinterpret_cast<const double2 *>(b);
    for (int64_t i = tid; i < n2; i += nth) {
        const double2 u = a2[i], v = b2[i];
        s0 = fma(u.x, v.x, s0);
        s1 = fma(u.y, v.y, s1);
    }
    for (int64_t i = 2 * n2 + tid; i < n; i += nth) s0 = fma(a[i], b[i], s0);
    const double bs = block_sum(s0 + s1, s_red);
    store_partial(part, bs);
}

__global__ __launch_bounds__(ITS_BLOCK) void absmax_partials_kernel(
    const double *__restrict__ v, int64_t n, double *__restrict__ part)
{
    __shared__ double s_red[4];
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    double m = 0.0;
    for (int64_t i = tid; i < n; i += nth) m = fmax(m, fabs(v[i]));
    const double bm = block_max(m, s_red);
    store_partial(part, bm);
}

__global__ __launch_bounds__(ITS_BLOCK) void reduce_finalize_kernel(
    const double *__restrict__ part, double *__restrict__ out, int mode)
{
    __shared__ double s_red[4];
    double r;
    if (mode == FIN_MAX) r = max_partials(part, s_red);
    else {
        r = sum_partials(part, s_red);
        if (mode == FIN_SQRT) r = sqrt(r);
    }
    if (threadIdx.x == 0) out[0] = r;
}

__global__ void sqrt_scalar_kernel(const double *in, double *out) { out[0] = sqrt(in[0]); }

__global__ __launch_bounds__(ITS_BLOCK) void axpby_kernel(
    double *__restrict__ out, const double *__restrict__ a, const double *__restrict__ b, int64_t n, double sc)
{
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    int64_t n2 = (al16(a) && al16(b) && al16(out)) ? n / 2 : 0;
    const double2 *a2 = reinterpret_cast<const double2 *>(a), *b2 = reinterpret_cast<const double2 *>(b);
    double2 *o2 = reinterpret_cast<double2 *>(out);
    for (int64_t i = tid; i < n2; i += nth) {
        const double2 u = a2[i], v = b2[i];
        double2 o; o.x = fma(sc, v.x, u.x); o.y = fma(sc, v.y, u.y);
        o2[i] = o;
    }
    for (int64_t i = 2 * n2 + tid; i < n; i += nth) out[i] = fma(sc, b[i], a[i]);
}

__global__ __launch_bounds__(ITS_BLOCK) void scale_kernel(
    double *__restrict__ out, const double *__restrict__ v, double sc, int64_t n)
{
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    for (int64_t i = tid; i < n; i += nth) out[i] = v[i] * sc;
}

__global__ __launch_bounds__(ITS_BLOCK) void fill_kernel(double *__restrict__ out, double v, int64_t n)
{
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    for (int64_t i = tid; i < n; i += nth) out[i] = v;
}

// vector kernels: at least 4 elements per thread, at most ITS_NPART workgroups.  Every consumer of a reduction
// re-sums all ITS_NPART partials in each of its workgroups, which costs 16 KB of L2 reads per workgroup: on small
// systems that re-sum, not the vectors, is the kernel (64 K rows: 5 us per launch on the full grid)
static inline dim3 vgrid(int64_t n)
{
    int64_t g = (n + 4 * ITS_BLOCK - 1) / (4 * ITS_BLOCK);
    return dim3((unsigned)std::min<int64_t>(std::max<int64_t>(g, 8), ITS_NPART));
}

void launch_dot_partials(const double *a, const double *b, int64_t n, double *part, hipStream_t s)
{ hipLaunchKernelGGL(dot_partials_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, a, b, n, part); }
void launch_absmax_partials(const double *v, int64_t n, double *part, hipStream_t s)
{ hipLaunchKernelGGL(absmax_partials_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, v, n, part); }
void launch_reduce_finalize(const double *part, double *out, int mode, hipStream_t s)
{ hipLaunchKernelGGL(reduce_finalize_kernel, dim3(1), dim3(ITS_BLOCK), 0, s, part, out, mode); }
void launch_sqrt_scalar(const double *in, double *out, hipStream_t s)
{ hipLaunchKernelGGL(sqrt_scalar_kernel, dim3(1), dim3(1), 0, s, in, out); }
void launch_axpby(double *out, const double *a, const double *b, int64_t n, double sc, hipStream_t s)
{ hipLaunchKernelGGL(axpby_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, out, a, b, n, sc); }
// a sweep level whose rows have no entries in L: x_i = (rhs_i - 0.0) / D_i, the value the chain kernels give for an empty
// row (rhs - (+0.0) == rhs bit for bit)
__global__ __launch_bounds__(ITS_BLOCK) void sweep_empty_kernel(double *x, const double *rhs, const double *__restrict__ D, int64_t n, double *acc_out)
{
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    for (int64_t i = tid; i < n; i += nth) {
        x[i] = (rhs[i] - 0.0) / D[i];
        if (acc_out) acc_out[i] = 0.0;               // the chain of an empty row (SpmvArgs::acc_out)
    }
}
void launch_sweep_empty(double *x, const double *rhs, const double *D, int64_t n, hipStream_t s, double *acc_out)
{ hipLaunchKernelGGL(sweep_empty_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, x, rhs, D, n, acc_out); }

void launch_scale(double *out, const double *v, double sc, int64_t n, hipStream_t s)
{ hipLaunchKernelGGL(scale_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, out, v, sc, n); }
void launch_copy(double *out, const double *v, int64_t n, hipStream_t s)
{ (void)hipMemcpyAsync(out, v, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s); }
void launch_fill(double *out, double v, int64_t n, hipStream_t s)
{ hipLaunchKernelGGL(fill_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, out, v, n); }

// ======================================================================================
// CG vector updates (methods/cg.hpp:204-230) with the reductions fused in.
// ======================================================================================
// Cache policy of the vector passes.  A CG trip moves 2.5 GB through a 256 MB Infinity Cache and 32 MB of L2, so
// what a pass leaves behind decides what the next pass finds.  A vector's LAST read of a trip, and every access to
// x (touched once per trip, by the x/p pass only), carry the non-temporal hint, so they do not push out p (read by
// the next SpMV), q and r (read by the next pass).  Measured on the 256^3 CG loop, cumulative: x loads + stores
// 2041 -> 2115 it/s (x/p pass 108.6 -> 99.3 us, SpMV 318 -> 309 us), q and r loads of the r pass -> 2128, r load
// of the x/p pass -> 2174.  NOT helped: the hint on the old-p load of the x/p pass (2154), on the matrix stream
// of the SpMV (2104), on the stores of q (2174), r (2159) and p (2173).
constexpr bool CG_NT_R = true;      // r pass: r read (last read of the old r)
constexpr bool CG_NT_Q = true;      // r pass: q read (q is dead afterwards)
constexpr bool CG_NT_XR = true;     // x/p pass: r read (last read of the trip)
constexpr bool CG_NT_XP = false;    // x/p pass: old p read
// Pass 1 of a CG trip: alpha = rr / (p.q); r -= alpha q; partial sums of r.r.  (x += alpha p needs p
// once more and moves into pass 2, which reads p anyway: 64n bytes per trip instead of 72n; the
// per-element operations and their order are those of cg.hpp:181-202, so the bits are unchanged.)
__global__ __launch_bounds__(ITS_BLOCK) void cg_update_r_kernel(
    double *__restrict__ r, const double *__restrict__ q, int64_t n, const double *__restrict__ part_pq,
    int pq_is_scalar, const double *__restrict__ rr_in, double *__restrict__ sc, double *__restrict__ part_rr)
{
    __shared__ double s_red[4];
    const double pq = pq_is_scalar ? part_pq[0] : sum_partials(part_pq, s_red);
    const double alpha = rr_in[0] / pq;                        // cg.hpp:162
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    double s0 = 0.0, s1 = 0.0;
    const int64_t n2 = n / 2;                                  // solver vectors are 256-byte aligned
    double2 *r2 = reinterpret_cast<double2 *>(r);
    const double2 *q2 = reinterpret_cast<const double2 *>(q);
    for (int64_t i = tid; i < n2; i += nth) {
        double2 rv = CG_NT_R ? ld_d2<1>(r + 2 * i) : r2[i];
        const double2 qv = CG_NT_Q ? ld_d2<1>(q + 2 * i) : q2[i];
        rv.x = fma(-alpha, qv.x, rv.x); rv.y = fma(-alpha, qv.y, rv.y);    // cg.hpp:182
        r2[i] = rv;
        s0 = fma(rv.x, rv.x, s0); s1 = fma(rv.y, rv.y, s1);                // cg.hpp:186
    }
    for (int64_t i = 2 * n2 + tid; i < n; i += nth) {
        const double rv = fma(-alpha, q[i], r[i]);
        r[i] = rv;
        s0 = fma(rv, rv, s0);
    }
    const double bs = block_sum(s0 + s1, s_red);
    store_partial(part_rr, bs);
    if (threadIdx.x == 0 && blockIdx.x == 0) { sc[SC_PQ] = pq; sc[SC_ALPHA] = alpha; }
}

// Pass 2: beta = rr_new / rr_old; x += alpha p (alpha from pass 1, sc[SC_ALPHA]); p = r + beta p.
__global__ __launch_bounds__(ITS_BLOCK) void cg_update_xp_kernel(
    double *__restrict__ x, double *__restrict__ p, const double *__restrict__ r, int64_t n,
    const double *__restrict__ part_rr, int rr_is_scalar, const double *__restrict__ rr_old_p,
    double *__restrict__ rr_new_p, double *__restrict__ sc, double *__restrict__ res_out)
{
    __shared__ double s_red[4];
    const double rr_new = rr_is_scalar ? part_rr[0] : sum_partials(part_rr, s_red);
    const double rr_old = rr_old_p[0];
    const double beta = rr_new / rr_old;                       // cg.hpp:188
    const double alpha = sc[SC_ALPHA];
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    const int64_t n2 = n / 2;
    double2 *p2 = reinterpret_cast<double2 *>(p), *x2 = reinterpret_cast<double2 *>(x);
    const double2 *r2 = reinterpret_cast<const double2 *>(r);
    for (int64_t i = tid; i < n2; i += nth) {
        double2 pv = CG_NT_XP ? ld_d2<1>(p + 2 * i) : p2[i];
        const v2d_t xl = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(x2 + i));
        double2 xv = make_double2(xl.x, xl.y);
        const double2 rv = CG_NT_XR ? ld_d2<1>(r + 2 * i) : r2[i];
        xv.x = fma(alpha, pv.x, xv.x); xv.y = fma(alpha, pv.y, xv.y);      // cg.hpp:181
        pv.x = fma(beta, pv.x, rv.x); pv.y = fma(beta, pv.y, rv.y);        // cg.hpp:202
        v2d_t xs; xs.x = xv.x; xs.y = xv.y;
        __builtin_nontemporal_store(xs, reinterpret_cast<v2d_t *>(x2 + i));
        p2[i] = pv;
    }
    for (int64_t i = 2 * n2 + tid; i < n; i += nth) {
        const double pv = p[i];
        x[i] = fma(alpha, pv, x[i]);
        p[i] = fma(beta, pv, r[i]);
    }
    // rr_new goes to the OTHER rr slot (rr_new_p != rr_old_p): other workgroups may still be
    // reading rr_old.  sc[SC_ALPHA] is only rewritten by the next trip's pass 1 (a later kernel).
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        rr_new_p[0] = rr_new; sc[SC_BETA] = beta;
        if (res_out) res_out[0] = sqrt(rr_new);                // utility_funcs.hpp:483
    }
}

void launch_cg_update_r(double *r, const double *q, int64_t n, const double *part_pq, int pq_is_scalar,
                        const double *rr_in, double *sc, double *part_rr, hipStream_t s)
{ hipLaunchKernelGGL(cg_update_r_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, r, q, n, part_pq, pq_is_scalar, rr_in, sc, part_rr); }

void launch_cg_update_xp(double *x, double *p, const double *r, int64_t n, const double *part_rr, int rr_is_scalar,
                         const double *rr_old, double *rr_new, double *sc, double *res_out, hipStream_t s)
{ hipLaunchKernelGGL(cg_update_xp_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, x, p, r, n, part_rr, rr_is_scalar, rr_old, rr_new, sc, res_out); }

// ======================================================================================
// GMRES: modified Gram-Schmidt as fused "finish previous axpy + start next dot" steps.
// ======================================================================================
// v_prev is read for the last time in this trip: the non-temporal hint keeps it from pushing w and v_next out
// of the caches (27-pt 128^3 GMRES(50): 0.503 -> 0.451 ms per trip; two elements per lane before that: 0.540 -> 0.503)
constexpr bool MGS_NT_PREV = true;
__global__ __launch_bounds__(ITS_BLOCK) void mgs_step_kernel(
    double *__restrict__ w, const double *__restrict__ v_prev, const double *__restrict__ part_prev,
    int prev_is_scalar, const double *__restrict__ v_next, double *__restrict__ part_next,
    double *__restrict__ h_out, int64_t n)
{
    __shared__ double s_red[4];
    double h = 0.0;
    if (v_prev) h = prev_is_scalar ? part_prev[0] : sum_partials(part_prev, s_red);
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    double s0 = 0.0;
    // two elements per lane (solver vectors are 256-byte aligned, ldv even); v_prev is read for the last time
    // this trip: non-temporal (MGS_NT_PREV)
    const int64_t n2 = (al16(w) && (!v_prev || al16(v_prev)) && (!v_next || al16(v_next))) ? n / 2 : 0;
    double2 *w2 = reinterpret_cast<double2 *>(w);
    const double2 *vn2 = reinterpret_cast<const double2 *>(v_next);
    double s1 = 0.0;
    for (int64_t i = tid; i < n2; i += nth) {
        double2 wi = w2[i];
        if (v_prev) {
            const double2 vp = MGS_NT_PREV ? ld_d2<1>(v_prev + 2 * i) : *reinterpret_cast<const double2 *>(v_prev + 2 * i);
            wi.x = fma(-h, vp.x, wi.x); wi.y = fma(-h, vp.y, wi.y);        // gmres.hpp:314
            w2[i] = wi;
        }
        const double2 vn = v_next ? vn2[i] : wi;
        s0 = fma(wi.x, vn.x, s0); s1 = fma(wi.y, vn.y, s1);                // gmres.hpp:286 / :359
    }
    for (int64_t i = 2 * n2 + tid; i < n; i += nth) {
        double wi = w[i];
        if (v_prev) { wi = fma(-h, v_prev[i], wi); w[i] = wi; }
        s0 = fma(wi, v_next ? v_next[i] : wi, s0);
    }
    const double bs = block_sum(s0 + s1, s_red);
    store_partial(part_next, bs);
    if (threadIdx.x == 0 && blockIdx.x == 0 && h_out && v_prev) h_out[0] = h;
}

__global__ __launch_bounds__(ITS_BLOCK) void mgs_normalize_kernel(
    double *__restrict__ v_out, const double *__restrict__ w, const double *__restrict__ part_ww,
    int ww_is_scalar, double *__restrict__ h_out, int64_t n)
{
    __shared__ double s_red[4];
    const double nrm = sqrt(ww_is_scalar ? part_ww[0] : sum_partials(part_ww, s_red));
    const double inv = 1.0 / nrm;                                          // gmres.hpp:389
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    const int64_t n2 = (al16(w) && al16(v_out)) ? n / 2 : 0;
    double2 *o2 = reinterpret_cast<double2 *>(v_out);
    for (int64_t i = tid; i < n2; i += nth) {
        const double2 wv = ld_d2<1>(w + 2 * i);                             // last read of w
        o2[i] = make_double2(wv.x * inv, wv.y * inv);
    }
    for (int64_t i = 2 * n2 + tid; i < n; i += nth) v_out[i] = w[i] * inv;
    if (blockIdx.x == 0 && threadIdx.x == 0 && h_out) h_out[0] = nrm;
}

__global__ __launch_bounds__(ITS_BLOCK) void gmres_update_x_kernel(
    double *__restrict__ x, const double *__restrict__ x0, const double *__restrict__ V, int64_t ldv,
    const double *__restrict__ y, int m, int64_t n)
{
    const int64_t tid = (int64_t)blockIdx.x * ITS_BLOCK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * ITS_BLOCK;
    for (int64_t i = tid; i < n; i += nth) {
        double t = 0.0;
        for (int k = 0; k < m; ++k) t = fma(V[(int64_t)k * ldv + i], y[k], t);   // kernels.hpp:511-513
        x[i] = x0[i] + t;                                                   // gmres.hpp:972
    }
}

void launch_mgs_step(double *w, const double *v_prev, const double *part_prev, int prev_is_scalar,
                     const double *v_next, double *part_next, double *h_out, int64_t n, hipStream_t s)
{ hipLaunchKernelGGL(mgs_step_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, w, v_prev, part_prev, prev_is_scalar, v_next, part_next, h_out, n); }
void launch_mgs_normalize(double *v_out, const double *w, const double *part_ww, int ww_is_scalar,
                          double *h_out, int64_t n, hipStream_t s)
{ hipLaunchKernelGGL(mgs_normalize_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, v_out, w, part_ww, ww_is_scalar, h_out, n); }
void launch_gmres_update_x(double *x, const double *x0, const double *V, int64_t ldv, const double *y_dev,
                           int m, int64_t n, hipStream_t s)
{ hipLaunchKernelGGL(gmres_update_x_kernel, vgrid(n), dim3(ITS_BLOCK), 0, s, x, x0, V, ldv, y_dev, m, n); }

// ======================================================================================
// Triangular sweep, one dependency level (or colour) per launch: each listed row computes
// x_i = (rhs_i - sum_j L_ij x_j) / D_i with the reference's in-order fused chain
// (kernels.hpp:460-472).  Rows of one level never read each other.
// ======================================================================================
// Short rows (<= SWEEP_SHORT entries, listed first) take one lane each, gathers eight at a time
// (sweep_short_kernel).  Every longer row gets a whole wave (sweep_long_kernel): SWEEP_CHUNK entries of
// col/val are loaded coalesced, x gathered in parallel, (val, x) parked in a wave-private LDS strip, and the
// chain runs over the strip in CRS order — the same operations in the same order as the reference, whatever
// the row length; the level's long rows run side by side.
constexpr int SWEEP_LPL = 8;                        // entries of a long row per lane per chunk
constexpr int SWEEP_CHUNK = SWEEP_LPL * ITS_WAVE;   // entries of a long row staged per trip
constexpr int SWEEP_LW_M = 16, SWEEP_LW_MIN = 2048;  // the register walk of the longest rows: entries per lane, shortest row that takes it
constexpr int SWEEP_MERGE_ROWS = 1 << 18;           // levels up to this size with both kinds of rows go out as one launch

__device__ __forceinline__ void sweep_short_rows(
    int bid, int nb, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
    const int *__restrict__ rows, int n_short, double *x, const double *rhs, const double *__restrict__ D, double *acc_out)
{
    {
        const int64_t nth = (int64_t)nb * ITS_BLOCK;
        for (int64_t k = (int64_t)bid * ITS_BLOCK + threadIdx.x; k < n_short; k += nth) {
            const int row = rows ? rows[k] : (int)k;
            const int j0 = rp[row], j1 = rp[row + 1];
            double s = 0.0;
            for (int j = j0; j < j1; j += 8) {
                int c[8]; double v[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = j + u < j1;
                    const int jj = ok ? j + u : j;
                    c[u] = col[jj]; v[u] = val[jj];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = x[c[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j + u < j1) s = fma(v[u], xv[u], s);
            }
            x[row] = (rhs[row] - s) / D[row];
            if (acc_out) acc_out[row] = s;
        }
    }
}

__global__ __launch_bounds__(ITS_BLOCK) void sweep_short_kernel(
    const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
    const int *__restrict__ rows, int n_short, double *x, const double *rhs, const double *__restrict__ D, double *acc_out)
{ sweep_short_rows(blockIdx.x, gridDim.x, rp, col, val, rows, n_short, x, rhs, D, acc_out); }

__device__ __forceinline__ void sweep_long_rows(
    int bid, int nb, double2 (*s_vx)[SWEEP_CHUNK + 8], const int *__restrict__ rp, const int *__restrict__ col,
    const double *__restrict__ val, const int *__restrict__ rows, int n_level_rows, int n_short, double *x,
    const double *rhs, const double *__restrict__ D, double *acc_out)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    // Long rows.  The chain is serial by definition (one fused multiply-add per entry, in CRS order), so the
    // level cannot finish before its longest row has walked its entries: everything else is kept off that path.
    //   loads    two chunks ahead: col/val of chunk c+2 and the x gathers of chunk c+1 (whose columns arrived a
    //            trip ago) are in flight while chunk c is walked, so no trip waits for a memory round trip
    //   chain    reads LDS in batches of 16 entries
    const int64_t wstride = (int64_t)nb * (ITS_BLOCK / ITS_WAVE);
    for (int64_t k = n_short + (int64_t)bid * (ITS_BLOCK / ITS_WAVE) + w; k < n_level_rows; k += wstride) {
        const int row = rows[k];
        const int rj0 = rp[row], rj1 = rp[row + 1];
        double acc = 0.0;
        if (rj1 - rj0 >= SWEEP_LW_MIN) {
            // The longest rows: the chain walks through the lanes' own registers instead of LDS (spmv_long_kernel,
            // its_panel.hip: lane j holds SWEEP_LW_M consecutive entries, every lane runs its multiply-adds from the running
            // sum, lane j's result is handed to all lanes for step j + 1).  Same entries, same order, same bits.
            constexpr int M = SWEEP_LW_M, CH = M * ITS_WAVE;
            int cn[M]; double vn[M], v1[M], x1[M], vv[M], xg[M];
#define SWEEP_LW_LOAD(c0v)                                                                \
            _Pragma("unroll") for (int u = 0; u < M; ++u) {                               \
                const int e_ = (c0v) + l * M + u;                                         \
                cn[u] = e_ < rj1 ? col[e_] : 0;                                           \
                vn[u] = e_ < rj1 ? val[e_] : 0.0;                                         \
            }
#define SWEEP_LW_GATHER(c0v)                                                              \
            _Pragma("unroll") for (int u = 0; u < M; ++u) {                               \
                const int e_ = (c0v) + l * M + u;                                         \
                v1[u] = vn[u]; x1[u] = e_ < rj1 ? x[cn[u]] : 0.0;                         \
            }
            SWEEP_LW_LOAD(rj0)
            SWEEP_LW_GATHER(rj0)
            SWEEP_LW_LOAD(rj0 + CH)
            for (int c0 = rj0; c0 < rj1; c0 += CH) {
#pragma unroll
                for (int u = 0; u < M; ++u) { vv[u] = v1[u]; xg[u] = x1[u]; }
                SWEEP_LW_GATHER(c0 + CH)
                SWEEP_LW_LOAD(c0 + 2 * CH)
                const int nsteps = (min(CH, rj1 - c0) + M - 1) / M;
                for (int j = 0; j < nsteps; ++j) {
                    double s_ = acc;
#pragma unroll
                    for (int u = 0; u < M; ++u) s_ = fma(vv[u], xg[u], s_);
                    acc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s_), j),
                                           __builtin_amdgcn_readlane(__double2loint(s_), j));
                }
            }
#undef SWEEP_LW_LOAD
#undef SWEEP_LW_GATHER
            if (l == 0) { x[row] = (rhs[row] - acc) / D[row]; if (acc_out) acc_out[row] = acc; }
            continue;
        }
        int cn[SWEEP_LPL];                    // columns of the chunk after next
        double vn[SWEEP_LPL];                 // its values
        double vv[SWEEP_LPL], xg[SWEEP_LPL];  // (val, x[col]) of the next chunk
        // entry e of a chunk starting at c0 sits in lane (e - c0) % 64, slot (e - c0) / 64
#define SWEEP_LOAD_CV(c0v)                                                                \
        _Pragma("unroll") for (int u = 0; u < SWEEP_LPL; ++u) {                           \
            const int e_ = (c0v) + l + ITS_WAVE * u;                                      \
            cn[u] = e_ < rj1 ? col[e_] : 0;                                               \
            vn[u] = e_ < rj1 ? val[e_] : 0.0;                                             \
        }
#define SWEEP_GATHER()                                                                    \
        _Pragma("unroll") for (int u = 0; u < SWEEP_LPL; ++u) { vv[u] = vn[u]; xg[u] = x[cn[u]]; }
        SWEEP_LOAD_CV(rj0)
        SWEEP_GATHER()
        SWEEP_LOAD_CV(rj0 + SWEEP_CHUNK)
        for (int c0 = rj0; c0 < rj1; c0 += SWEEP_CHUNK) {
#pragma unroll
            for (int u = 0; u < SWEEP_LPL; ++u) s_vx[w][l + ITS_WAVE * u] = make_double2(vv[u], xg[u]);
            SWEEP_GATHER()                                  // chunk c+1: its columns were loaded a trip ago
            SWEEP_LOAD_CV(c0 + 2 * SWEEP_CHUNK)             // chunk c+2
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int ne = min(SWEEP_CHUNK, rj1 - c0);
            {
                // the walk: batches of 16 entries — their LDS reads go out together, the multiply-adds follow them in (one
                // read latency per batch).  (Two alternating register sets of 8, the next one read during the current
                // one's multiply-adds, compiled into 16 register copies per trip and waits on reads just issued: ~30 cycles
                // per entry instead of ~12; see spmv_long_kernel, its_panel.hip.)
#define SWEEP_RD(S, N, qv) _Pragma("unroll") for (int u = 0; u < (N); ++u) S[u] = s_vx[w][(qv) + u];
#define SWEEP_FMA(S, N, cnt) _Pragma("unroll") for (int u = 0; u < (N); ++u) if (u < (cnt)) acc = fma(S[u].x, S[u].y, acc);
                int q = 0;
                for (; q + 16 <= ne; q += 16) {
                    double2 pw[16];
                    SWEEP_RD(pw, 16, q)
                    SWEEP_FMA(pw, 16, 16)
                }
                for (; q < ne; q += 8) {
                    double2 p8[8];
                    SWEEP_RD(p8, 8, q)                 // (the strip is 8 entries longer than a chunk)
                    SWEEP_FMA(p8, 8, ne - q)
                }
#undef SWEEP_RD
#undef SWEEP_FMA
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
#undef SWEEP_LOAD_CV
#undef SWEEP_GATHER
        if (l == 0) { x[row] = (rhs[row] - acc) / D[row]; if (acc_out) acc_out[row] = acc; }
    }
}

// (a kernel of its own for levels that only have long rows: its staging registers would halve the occupancy of the
// latency-bound short rows of big levels)
__global__ __launch_bounds__(ITS_BLOCK) void sweep_long_kernel(
    const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
    const int *__restrict__ rows, int n_level_rows, int n_short, double *x,
    const double *rhs, const double *__restrict__ D, double *acc_out)
{
    __shared__ double2 s_vx[ITS_BLOCK / ITS_WAVE][SWEEP_CHUNK + 8];   // + 8: the walk's last batch may read past the chunk
    sweep_long_rows(blockIdx.x, gridDim.x, s_vx, rp, col, val, rows, n_level_rows, n_short, x, rhs, D, acc_out);
}

// A level that has both kinds of rows and is small (a natural-order sweep of an irregular matrix: two dozen levels of a
// few thousand rows, each with a handful of hub rows) as ONE launch: the first gl workgroups take the long rows — they
// last longest — the others the short ones.  Two launches per level were 14 + 6 us on the 200 000-row power-law matrix.
__global__ __launch_bounds__(ITS_BLOCK) void sweep_level_kernel(
    const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
    const int *__restrict__ rows, int n_level_rows, int n_short, int gl, double *x,
    const double *rhs, const double *__restrict__ D, double *acc_out)
{
    __shared__ double2 s_vx[ITS_BLOCK / ITS_WAVE][SWEEP_CHUNK + 8];
    if ((int)blockIdx.x < gl) sweep_long_rows(blockIdx.x, gl, s_vx, rp, col, val, rows, n_level_rows, n_short, x, rhs, D, acc_out);
    else sweep_short_rows(blockIdx.x - gl, gridDim.x - gl, rp, col, val, rows, n_short, x, rhs, D, acc_out);
}

// --------------------------------------------------------------------------------------
// Level-ordered copy of L for the natural-order sweep of matrices with short rows (stencils: 766 dependency levels
// of scattered rows at 256^3).  The level-list kernel above walks rows[k] -> row_ptr -> col/val -> x: four dependent
// memory round trips per level, and a level is little else.  Here every position p of the level list carries its
// row, its diagonal and its (<= W) entries of L, stored slot-major over the whole list (sw_col[s*n + p]), so a
// level reads them with coalesced loads at once and only the gathers of x / rhs depend on them: two round trips.
// Same entries in the same (CRS) order: bit-identical sweep.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(ITS_BLOCK) void sweep_ell_build_kernel(
    int n, int W, const int *__restrict__ rows, const int *__restrict__ rp, const int *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ D, int *__restrict__ sw_col, double *__restrict__ sw_val,
    double *__restrict__ sw_D)
{
    const int p = blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (p >= n) return;
    const int row = rows[p];
    const int j0 = rp[row], len = rp[row + 1] - j0;
    for (int s = 0; s < W; ++s) {
        sw_col[(size_t)s * n + p] = s < len ? col[j0 + s] : -1;
        sw_val[(size_t)s * n + p] = s < len ? val[j0 + s] : 0.0;
    }
    sw_D[p] = D[row];
}

template <int W>
__global__ __launch_bounds__(ITS_BLOCK) void sweep_ell_kernel(
    int n, int p0, int p1, const int *__restrict__ rows, const int *__restrict__ sw_col, const double *__restrict__ sw_val,
    const double *__restrict__ sw_D, double *x, const double *rhs, double *acc_out)
{
    const int p = p0 + blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (p >= p1) return;
    const int row = rows[p];
    int c[W]; double v[W], xv[W];
#pragma unroll
    for (int s = 0; s < W; ++s) { c[s] = sw_col[(size_t)s * n + p]; v[s] = sw_val[(size_t)s * n + p]; }
    const double d = sw_D[p];
    const double r = rhs[row];
#pragma unroll
    for (int s = 0; s < W; ++s) xv[s] = x[c[s] < 0 ? 0 : c[s]];
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < W; ++s) if (c[s] >= 0) acc = fma(v[s], xv[s], acc);
    x[row] = (r - acc) / d;
    if (acc_out) acc_out[row] = acc;
}

void launch_sweep_ell_build(int n, int W, const int *rows, const CsrDev &L, const double *D, int *sw_col, double *sw_val,
                            double *sw_D, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(sweep_ell_build_kernel, dim3((n + ITS_BLOCK - 1) / ITS_BLOCK), dim3(ITS_BLOCK), 0, s, n, W, rows,
                       L.row_ptr, L.col, L.val, D, sw_col, sw_val, sw_D);
}

void launch_sweep_ell_level(int n, int W, int p0, int p1, const int *rows, const int *sw_col, const double *sw_val,
                            const double *sw_D, double *x, const double *rhs, hipStream_t s, double *acc_out)
{
    if (p1 <= p0) return;
    dim3 g((p1 - p0 + ITS_BLOCK - 1) / ITS_BLOCK), b(ITS_BLOCK);
#define ITS_LAUNCHSW(WW) hipLaunchKernelGGL((sweep_ell_kernel<WW>), g, b, 0, s, n, p0, p1, rows, sw_col, sw_val, sw_D, x, rhs, acc_out)
    switch (W) {
    case 1: ITS_LAUNCHSW(1); break;
    case 2: ITS_LAUNCHSW(2); break;
    case 3: ITS_LAUNCHSW(3); break;
    case 4: ITS_LAUNCHSW(4); break;
    case 5: ITS_LAUNCHSW(5); break;
    case 6: ITS_LAUNCHSW(6); break;
    case 7: ITS_LAUNCHSW(7); break;
    case 8: ITS_LAUNCHSW(8); break;
    case 9: ITS_LAUNCHSW(9); break;
    case 10: ITS_LAUNCHSW(10); break;
    case 11: ITS_LAUNCHSW(11); break;
    case 12: ITS_LAUNCHSW(12); break;
    case 13: ITS_LAUNCHSW(13); break;
    case 14: ITS_LAUNCHSW(14); break;
    case 15: ITS_LAUNCHSW(15); break;
    default: ITS_LAUNCHSW(16); break;
    }
#undef ITS_LAUNCHSW
}

// A RUN of consecutive small levels (each of at most sweep_run_rows(W) list positions: the natural-order sweep of a 2-D
// stencil has nothing else — 511 levels of <= 256 rows on a 256 x 256 grid — and a 3-D one starts and ends with them) in
// ONE launch of ONE workgroup: a lane owns one list position per level, the levels are separated by a workgroup barrier
// instead of a kernel boundary (all waves of a workgroup share the CU's L1, so a barrier makes the level's x visible),
// and the next level's row, diagonal, right-hand side and entries — which do not depend on x — are loaded while the
// current level's gathers are in flight: one dependent round trip per level.  A launch per level cost 3.2 us
// (its_solve -gs on the 256 x 256 Laplacian: 1.62 ms per trip).  Same entries in the same order: bit-identical.
// (1024 lanes for copies of up to 8 entries per row, 512 for wider ones: 3 W registers for the level in hand and as many
// for the next one)
template <int W, int NT>
__global__ __launch_bounds__(NT) void sweep_ell_run_kernel(
    int n, int l0, int l1, const int *__restrict__ lptr, const int *__restrict__ rows, const int *__restrict__ sw_col,
    const double *__restrict__ sw_val, const double *__restrict__ sw_D, double *x, const double *rhs, double *acc_out)
{
    const int t = threadIdx.x;
    int row = 0, c[W]; double v[W], d = 1.0, r = 0.0; bool have = false;
    int rown = 0, cn[W]; double vn[W], dn = 1.0, rn = 0.0; bool haven = false;
    {
        const int p = lptr[l0] + t;
        have = p < lptr[l0 + 1];
#pragma unroll
        for (int s = 0; s < W; ++s) { c[s] = -1; v[s] = 0.0; }
        if (have) {
            row = rows[p];
#pragma unroll
            for (int s = 0; s < W; ++s) { c[s] = sw_col[(size_t)s * n + p]; v[s] = sw_val[(size_t)s * n + p]; }
            d = sw_D[p]; r = rhs[row];
        }
    }
    for (int l = l0; l < l1; ++l) {
        double xv[W];
#pragma unroll
        for (int s = 0; s < W; ++s) xv[s] = (have && c[s] >= 0) ? x[c[s]] : 0.0;
        // the next level's operands: nothing of them is written by this level (a row is written once per sweep, by the
        // lane that owns it; rhs may alias x — the preconditioner sweeps in place — and rhs[row] of a LATER level is still
        // the right-hand side)
        haven = false;
#pragma unroll
        for (int s = 0; s < W; ++s) { cn[s] = -1; vn[s] = 0.0; }
        if (l + 1 < l1) {
            const int p = lptr[l + 1] + t;
            haven = p < lptr[l + 2];
            if (haven) {
                rown = rows[p];
#pragma unroll
                for (int s = 0; s < W; ++s) { cn[s] = sw_col[(size_t)s * n + p]; vn[s] = sw_val[(size_t)s * n + p]; }
                dn = sw_D[p]; rn = rhs[rown];
            }
        }
        if (have) {
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < W; ++s) if (c[s] >= 0) acc = fma(v[s], xv[s], acc);
            x[row] = (r - acc) / d;
            if (acc_out) acc_out[row] = acc;     // (may alias rhs: rhs[row] was read a level ago, by this lane)
        }
        __syncthreads();
        have = haven; row = rown; d = dn; r = rn;
#pragma unroll
        for (int s = 0; s < W; ++s) { c[s] = cn[s]; v[s] = vn[s]; }
    }
}

void launch_sweep_ell_run(int n, int W, int l0, int l1, const int *level_ptr_dev, const int *rows, const int *sw_col,
                          const double *sw_val, const double *sw_D, double *x, const double *rhs, hipStream_t s, double *acc_out)
{
    if (l1 <= l0) return;
    dim3 g(1);
#define ITS_LAUNCHSR(WW, NT) hipLaunchKernelGGL((sweep_ell_run_kernel<WW, NT>), g, dim3(NT), 0, s, n, l0, l1, level_ptr_dev, rows, sw_col, sw_val, sw_D, x, rhs, acc_out)
    switch (W) {
    case 1: ITS_LAUNCHSR(1, 1024); break;
    case 2: ITS_LAUNCHSR(2, 1024); break;
    case 3: ITS_LAUNCHSR(3, 1024); break;
    case 4: ITS_LAUNCHSR(4, 1024); break;
    case 5: ITS_LAUNCHSR(5, 1024); break;
    case 6: ITS_LAUNCHSR(6, 1024); break;
    case 7: ITS_LAUNCHSR(7, 1024); break;
    case 8: ITS_LAUNCHSR(8, 1024); break;
    case 9: ITS_LAUNCHSR(9, 512); break;
    case 10: ITS_LAUNCHSR(10, 512); break;
    case 11: ITS_LAUNCHSR(11, 512); break;
    case 12: ITS_LAUNCHSR(12, 512); break;
    case 13: ITS_LAUNCHSR(13, 512); break;
    case 14: ITS_LAUNCHSR(14, 512); break;
    case 15: ITS_LAUNCHSR(15, 512); break;
    default: ITS_LAUNCHSR(16, 512); break;
    }
#undef ITS_LAUNCHSR
}
int sweep_run_rows(int W) { return W <= 8 ? 1024 : 512; }

void launch_sweep_level(const CsrDev &L, const int *rows, int n_level_rows, int n_short, double *x,
                        const double *rhs, const double *D, hipStream_t s, double *acc_out)
{
    if (n_level_rows <= 0) return;
    if (!rows) n_short = n_level_rows;                       // identity order has no long-row list
    int gs = (n_short + ITS_BLOCK - 1) / ITS_BLOCK;
    if (gs > 4 * ITS_NPART) gs = 4 * ITS_NPART;
    const int n_long = n_level_rows - n_short;
    int gl = (n_long + 3) / 4;
    if (gl > 4 * ITS_NPART) gl = 4 * ITS_NPART;
    // rows of a level are independent, so the order of the two launches does not matter
    if (gs > 0 && gl > 0 && n_level_rows <= SWEEP_MERGE_ROWS) {
        hipLaunchKernelGGL(sweep_level_kernel, dim3(gl + gs), dim3(ITS_BLOCK), 0, s, L.row_ptr, L.col, L.val, rows, n_level_rows,
                           n_short, gl, x, rhs, D, acc_out);
        return;
    }
    if (gs > 0)
        hipLaunchKernelGGL(sweep_short_kernel, dim3(gs), dim3(ITS_BLOCK), 0, s, L.row_ptr, L.col, L.val, rows, n_short, x, rhs, D, acc_out);
    if (gl > 0)
        hipLaunchKernelGGL(sweep_long_kernel, dim3(gl), dim3(ITS_BLOCK), 0, s, L.row_ptr, L.col, L.val, rows, n_level_rows,
                           n_short, x, rhs, D, acc_out);
}

}  // namespace its
