// its_mgs.hip — the modified Gram-Schmidt sweep of a GMRES trip (methods/gmres.hpp:269-329,359,389) as ONE persistent
// launch that keeps w in registers.
//
// Why.  As one launch per basis vector (mgs_step_kernel, its_kernels.hip) a step moves 32 n bytes — read w, v_{j-1},
// v_j, write w — and runs at the bandwidth ceiling of that form: 10.9 us at n = 128^3, k + 2 launches per trip, 0.29 of
// the 0.42 ms trip of GMRES(50) on the 27-point matrix.  The arithmetic only needs v_j once: with w held in registers
// across the steps a step reads 8 n bytes.
// How.  One workgroup per CU (15 data waves + 1 polling wave), every data lane owns EP pairs of elements of w for the
// whole launch.  Step j: partial sums of w . v_j -> one 8-byte SLOT per workgroup and step, preset to a sentinel
// (all ones: there are two sets of slots, a launch uses one and presets the other for the launch after it — a memset
// per trip cost 4.8 us); the polling wave of workgroup 0 gathers the G slots in a fixed order
// and publishes h_j in MGS_NB copies (watched by 1 / MGS_NB of the other workgroups' polling waves each: 256 pollers on
// one word cost 0.5 us of the 3.7).  Data and arrival flag
// are the same word, stores and loads are agent-scope relaxed atomics (write-through / L2 bypass): no fence, no counter.
// The loads of v_{j+1} are issued before the wait, so the stream of the next vector hides behind the reduction.  Then
// w -= h_j v_j from the registers that still hold v_j, and after the last vector ||w||, v_{k+1} = w / ||w||.
// Measured (tools/mgslab.hip, n = 128^3, 27 steps): 5.4 us per step, of which 3.3 us is the reduction (two trips to
// memory and back; gathering all-to-all instead of through workgroup 0: 4.3 us, through 8 workgroups: 4.5 us).
// The same operations on every element in the same order as the per-step kernels (fma(-h, v, w), w * (1 / nrm)); the
// dot products are summed in another (fixed) order, so H differs from theirs in the last bits — as the reference's
// own OpenMP reductions do from run to run.
// Every wait is bounded by wall-clock time: a grid that is not co-resident (another process on the card) ends, writes
// nothing but slots, and the caller falls back to the per-step kernels for that trip — w in memory is never modified.
// How the host learns of it: H[k+1,k] (h_out[nv]) stays armed.  A workgroup that gives up at a step j < nv never
// publishes its slot of step j + 1, so workgroup 0 gives up there.  At the LAST step there is no next step: a follower
// that runs out of time there claims its copy of the published sum with a compare-and-swap (sentinel -> MGS_GAVEUP), and
// workgroup 0 publishes that step with compare-and-swaps too — exactly one of the two wins a word, so either the
// follower gets the sum after all or workgroup 0 learns that somebody left and does not write h_out[nv] (round 3; before,
// such a follower returned without its slice of v_{k+1} and the host accepted the trip).  Followers also wait twice as
// long as workgroup 0, so a sum published within its limit is seen.
#include "its_kernels.h"

namespace its {

constexpr unsigned long long MGS_SENT = ~0ull;
constexpr unsigned long long MGS_QNAN = 0x7FF8000000000000ull;
constexpr int MGS_NWD = 15;                       // data waves per workgroup (+ 1 polling wave)
constexpr int MGS_NT = MGS_NWD * 64;
constexpr int MGS_KQ = 8;                         // slots per polling lane: G <= 512
constexpr int MGS_NB = 16;                        // copies of the published sum (slot layout: MGS_NB x bst words, then G per step)
constexpr unsigned long long MGS_GAVEUP = ~0ull - 1;  // left in a published-sum word by a follower that ran out of time at the last step
constexpr long long MGS_WAIT_TICKS = 2000000;     // 20 ms of the 100 MHz wall clock (workgroup 0; the followers wait twice as long)

__device__ __forceinline__ double mgs_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void mgs_publish(unsigned long long *slot, double v)
{
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits == MGS_SENT || bits == MGS_GAVEUP) bits = MGS_QNAN;        // a NaN with that payload would read as "not there yet" / "gave up"
    __hip_atomic_store(slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the last step's sum: published only into words nobody has given up on; false when a follower was there first
__device__ __forceinline__ bool mgs_publish_claim(unsigned long long *slot, double v)
{
    unsigned long long bits = (unsigned long long)__double_as_longlong(v), expect = MGS_SENT;
    if (bits == MGS_SENT || bits == MGS_GAVEUP) bits = MGS_QNAN;
    return __hip_atomic_compare_exchange_strong(slot, &expect, bits, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int EP>
__global__ __launch_bounds__(MGS_NT + 64) void mgs_persist_kernel(
    const double *__restrict__ w_in, const double *__restrict__ V, int64_t ld, int nv,
    double *__restrict__ v_out, double *__restrict__ h_out, unsigned long long *slots, unsigned long long *__restrict__ arm,
    int rows, int bst, int64_t n, int inject_fail)
{
    __shared__ double s_red[MGS_NWD];
    __shared__ double s_h;
    __shared__ int s_fail;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const bool poller = wv == MGS_NWD;
    const int G = gridDim.x;
    const int64_t npair = poller ? 0 : (n + 1) / 2;
    const int64_t tail = (n & 1) ? n / 2 : -1;                  // the pair whose second element is past the end
    const int64_t stride = (int64_t)G * MGS_NT;
    const int64_t g = (int64_t)blockIdx.x * MGS_NT + t;
    auto ld2 = [&](const double *base, int64_t i) {
        double2 d = make_double2(0.0, 0.0);
        if (i < npair) { d = reinterpret_cast<const double2 *>(base)[i]; if (i == tail) d.y = 0.0; }
        return d;
    };
    double2 w[EP], vc[EP], vn[EP];
    if (t == 0) s_fail = 0;
    if (poller) {
        // preset the OTHER set of slots for the next launch (nobody reads it during this one; the kernel boundary
        // makes the stores visible): this workgroup's slot of every step, workgroup 0 also the published copies
        for (int r = lane; r < rows; r += 64) arm[(size_t)MGS_NB * bst + (size_t)r * G + blockIdx.x] = MGS_SENT;
        if (blockIdx.x == 0) for (int i = lane; i < MGS_NB * bst; i += 64) arm[i] = MGS_SENT;
    }
#pragma unroll
    for (int u = 0; u < EP; ++u) { w[u] = ld2(w_in, g + u * stride); vn[u] = nv > 0 ? ld2(V, g + u * stride) : make_double2(0.0, 0.0); }
    for (int j = 0; j <= nv; ++j) {
        // j < nv: h_j = w . v_j (gmres.hpp:286);  j == nv: ||w||^2 (gmres.hpp:359)
        if (!poller) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int u = 0; u < EP; ++u) {
                vc[u] = j < nv ? vn[u] : w[u];
                s0 = fma(w[u].x, vc[u].x, s0); s1 = fma(w[u].y, vc[u].y, s1);
            }
            if (j + 1 < nv) {
                const double *vp = V + (int64_t)(j + 1) * ld;
#pragma unroll
                for (int u = 0; u < EP; ++u) vn[u] = ld2(vp, g + u * stride);
            }
            const double ws = mgs_wave_sum(s0 + s1);
            if (lane == 0) s_red[wv] = ws;
        }
        __syncthreads();
        unsigned long long *sl = slots + (size_t)MGS_NB * bst + (size_t)j * G;      // this step's partial sums
        unsigned long long *bc = slots + j;                                         // copy c of its sum: bc[c * bst]
        if (poller) {
            if (lane == 0) {
                double bs = 0.0;
#pragma unroll
                for (int q = 0; q < MGS_NWD; ++q) bs += s_red[q];
                mgs_publish(sl + blockIdx.x, bs);
            }
            double acc = 0.0; bool fail = false;
            const long long t0 = wall_clock64();
            if (blockIdx.x == 0 && inject_fail == 1) {
                fail = true;                                     // test hook: workgroup 0 never publishes, the others run into their time limit
            } else if (blockIdx.x == 0) {
                // all of a lane's slots are asked for at once; only those still pending are asked for again
                unsigned long long b[MGS_KQ];
#pragma unroll
                for (int u = 0; u < MGS_KQ; ++u) b[u] = MGS_SENT;
                for (int spins = 0;; ++spins) {
#pragma unroll
                    for (int u = 0; u < MGS_KQ; ++u) {
                        const int q = lane + u * 64;
                        if (q < G && b[u] == MGS_SENT) b[u] = __hip_atomic_load(sl + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    bool pend = false;
#pragma unroll
                    for (int u = 0; u < MGS_KQ; ++u) pend |= (lane + u * 64 < G) && b[u] == MGS_SENT;
                    if (!__any(pend)) break;
                    if ((spins & 63) == 63 && wall_clock64() - t0 > MGS_WAIT_TICKS) { fail = true; break; }
                }
#pragma unroll
                for (int u = 0; u < MGS_KQ; ++u) if (lane + u * 64 < G && b[u] != MGS_SENT) acc += __longlong_as_double((long long)b[u]);
                acc = mgs_wave_sum(acc);
                if (inject_fail == 2 && j == nv) while (wall_clock64() - t0 < 20000) { }      // test hook: 0.2 ms late, the follower below is first
                if (!fail && lane < MGS_NB) {
                    if (j < nv) mgs_publish(bc + (size_t)lane * bst, acc);
                    else if (!mgs_publish_claim(bc + (size_t)lane * bst, acc)) fail = true;   // a follower has left: no H[k+1,k]
                }
                fail = __any(fail);
            } else {
                unsigned long long b;
                unsigned long long *mine = bc + (size_t)(blockIdx.x % MGS_NB) * bst;
                bool late = inject_fail == 2 && j == nv && blockIdx.x == 1;                   // test hook: gives up at the last step at once
                if (!late)
                    for (int spins = 0; (b = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == MGS_SENT; ++spins)
                        if ((spins & 63) == 63 && wall_clock64() - t0 > 2 * MGS_WAIT_TICKS) { late = true; break; }
                if (late && j == nv) {
                    // the last step has no next one in which workgroup 0 would miss this workgroup: claim the word
                    b = MGS_SENT;
                    late = __hip_atomic_compare_exchange_strong(mine, &b, MGS_GAVEUP, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }                                                                             // (lost: b is what was there — the sum, or another's MGS_GAVEUP)
                fail = late || b == MGS_GAVEUP;
                acc = __longlong_as_double((long long)b);
            }
            if (lane == 0) { s_h = acc; if (fail) s_fail = 1; }
        }
        __syncthreads();
        if (s_fail) return;                                      // nothing but slots and entries of h_out has been written
        const double h = s_h;
        if (j < nv) {
#pragma unroll
            for (int u = 0; u < EP; ++u) { w[u].x = fma(-h, vc[u].x, w[u].x); w[u].y = fma(-h, vc[u].y, w[u].y); }   // gmres.hpp:314
            if (blockIdx.x == 0 && t == 0) h_out[j] = h;
        } else {
            const double nrm = sqrt(h), inv = 1.0 / nrm;         // gmres.hpp:359,389
#pragma unroll
            for (int u = 0; u < EP; ++u) {
                const int64_t i = g + u * stride;
                if (i < npair) {
                    if (i == tail) v_out[2 * i] = w[u].x * inv;
                    else reinterpret_cast<double2 *>(v_out)[i] = make_double2(w[u].x * inv, w[u].y * inv);
                }
            }
            if (blockIdx.x == 0 && t == 0) h_out[nv] = nrm;
        }
    }
}

typedef void (*mgs_kernel_t)(const double *, const double *, int64_t, int, double *, double *, unsigned long long *, unsigned long long *, int, int, int64_t, int);
static mgs_kernel_t mgs_kernel_for(int ep)
{
    switch (ep) {
    case 1: return mgs_persist_kernel<1>; case 2: return mgs_persist_kernel<2>; case 3: return mgs_persist_kernel<3>;
    case 4: return mgs_persist_kernel<4>; case 5: return mgs_persist_kernel<5>; case 6: return mgs_persist_kernel<6>;
    }
    return nullptr;                               // 7 and up spill: 12 registers per pair on top of ~50, 128 per lane at 16 waves per CU
}
static int mgs_ep(int64_t n, int G) { return (int)(((n + 1) / 2 + (int64_t)G * MGS_NT - 1) / ((int64_t)G * MGS_NT)); }

// Workgroups the launch would use on this device for vectors of n doubles (one per CU), 0 when the vectors do not fit the
// registers of one resident grid (n <= 12 * 960 * CUs: 2.9 M on 256 CUs) or the kernel cannot have one workgroup per CU.
int mgs_persist_grid(int64_t n, int device)
{
    hipDeviceProp_t prop;
    if (n < 1 || hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    // one workgroup per CU for long vectors; for short ones only as many as one pair per lane needs (a multiple of 8, at
    // least 8): the grid-wide sum is cheaper with fewer participants (n = 65 536, 52 steps: 64 workgroups 2.6 us per step,
    // 256: 3.1 us; a launch per basis vector: 4.4 us + the gap between launches)
    const int gmax = std::min(prop.multiProcessorCount, 64 * MGS_KQ);
    const int64_t want = (((n + 1) / 2 + MGS_NT - 1) / MGS_NT + 7) / 8 * 8;
    const int G = (int)std::max<int64_t>(std::min<int64_t>(want, gmax), std::min(8, gmax));
    const int ep = mgs_ep(n, G);
    mgs_kernel_t k = mgs_kernel_for(ep);
    if (!k) return 0;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, MGS_NT + 64, 0) != hipSuccess || occ < 1) return 0;
    return G;
}
static int mgs_bst(int m) { return (m + 2 + 31) & ~31; }
static size_t mgs_set_words(int m, int G) { return (size_t)MGS_NB * mgs_bst(m) + (size_t)(m + 2) * (size_t)G; }
size_t mgs_persist_slot_words(int m, int G) { return 2 * mgs_set_words(m, G); }
// both sets to the sentinel: at set-up and after a launch that gave up
hipError_t mgs_persist_reset(unsigned long long *slots, int m, int G, hipStream_t s)
{ return hipMemsetAsync(slots, 0xFF, mgs_persist_slot_words(m, G) * sizeof(unsigned long long), s); }

// w_in: A v_k (not modified).  V: basis, nv = k + 1 vectors of leading dimension ld.  v_out: V[k+1].  h_out: nv + 1 words
// (the Hessenberg column).  slots: mgs_persist_slot_words(m, G) words, all ones before the first launch
// (mgs_persist_reset); m (the restart length) >= nv; parity: 0 / 1, alternating from launch to launch.
hipError_t launch_mgs_persist(const double *w_in, const double *V, int64_t ld, int nv, double *v_out, double *h_out,
                              unsigned long long *slots, int parity, int m, int G, int64_t n, hipStream_t s, int inject_fail)
{
    mgs_kernel_t k = mgs_kernel_for(mgs_ep(n, G));
    if (!k) return hipErrorInvalidValue;
    if (nv > m) return hipErrorInvalidValue;
    const int bst = mgs_bst(m);
    unsigned long long *use = slots + (parity & 1) * mgs_set_words(m, G), *arm = slots + ((parity & 1) ^ 1) * mgs_set_words(m, G);
    hipLaunchKernelGGL(k, dim3(G), dim3(MGS_NT + 64), 0, s, w_in, V, ld, nv, v_out, h_out, use, arm, m + 2, bst, n, inject_fail);
    return hipGetLastError();
}

}  // namespace its
