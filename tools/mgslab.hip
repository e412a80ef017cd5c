// mgslab — can the modified Gram-Schmidt sweep of a GMRES trip run as ONE persistent launch that keeps w in registers?
// Not part of the library.
//
// Today (csrc/its_kernels.hip, mgs_step_kernel): one launch per basis vector, 32 n bytes each (read w, v_{j-1}, v_j, write
// w) — 10.9 us per step at n = 128^3, the bandwidth ceiling of that form.  With w in registers a step only has to read
// v_j (8 n bytes, 2.7 us), but it ends in a grid-wide reduction.  Here that reduction is one 8-byte slot per workgroup and
// step, preset to a sentinel: a workgroup publishes its partial sum with one agent-scope store and polls the other
// workgroups' slots with agent-scope loads — data and arrival flag are the same word, no fence, no counter — while the
// loads of v_{j+1} are already in flight.  Every poll loop is bounded: a grid that is not co-resident ends with an error
// flag instead of hanging.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mgslab tools/mgslab.hip && ./tools/mgslab [n] [nv] [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr unsigned long long SENT = ~0ull;
constexpr int SPIN_MAX = 1 << 18;          // polls of ~0.2 us each: ~50 ms, then give up

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// NT threads per workgroup, EP element PAIRS per thread; grid G workgroups, G <= NT
template <int NT, int EP>
__global__ __launch_bounds__(NT) void mgs_persist_kernel(const double *__restrict__ w_in, const double *__restrict__ V, int64_t ld,
                                                         int nv, double *__restrict__ v_out, double *__restrict__ h_out,
                                                         unsigned long long *slots, int *err, int64_t n)
{
    constexpr int NW = NT / 64;
    __shared__ double s_red[NW];
    __shared__ double s_h;
    __shared__ int s_fail;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int G = gridDim.x;
    const int64_t n2 = n / 2;                       // lab: n even
    const int64_t stride = (int64_t)G * NT;
    const int64_t g = (int64_t)blockIdx.x * NT + t;
    double2 w[EP], vc[EP], vn[EP];
    if (t == 0) s_fail = 0;
#pragma unroll
    for (int u = 0; u < EP; ++u) {
        const int64_t i = g + u * stride;
        w[u] = i < n2 ? reinterpret_cast<const double2 *>(w_in)[i] : make_double2(0.0, 0.0);
        vn[u] = (i < n2 && nv > 0) ? reinterpret_cast<const double2 *>(V)[i] : make_double2(0.0, 0.0);
    }
    for (int j = 0; j <= nv; ++j) {
        // j < nv: h_j = w . v_j;  j == nv: ||w||^2
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int u = 0; u < EP; ++u) {
            vc[u] = j < nv ? vn[u] : w[u];
            s0 = fma(w[u].x, vc[u].x, s0); s1 = fma(w[u].y, vc[u].y, s1);
        }
        if (j + 1 < nv) {
            const double2 *vp = reinterpret_cast<const double2 *>(V + (int64_t)(j + 1) * ld);
#pragma unroll
            for (int u = 0; u < EP; ++u) { const int64_t i = g + u * stride; vn[u] = i < n2 ? vp[i] : make_double2(0.0, 0.0); }
        }
        double ws = wave_sum(s0 + s1);
        if (lane == 0) s_red[wv] = ws;
        __syncthreads();
        unsigned long long *sl = slots + (size_t)j * G;
        if (t == 0) {
            double bs = 0.0;
#pragma unroll
            for (int q = 0; q < NW; ++q) bs += s_red[q];
            unsigned long long bits = __double_as_longlong(bs);
            if (bits == SENT) bits = 0x7FF8000000000000ull;
            __hip_atomic_store(sl + blockIdx.x, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // wave 0 gathers all G partials (G <= 64 * 16) in a fixed order
        if (wv == 0) {
            double acc = 0.0; bool fail = false;
            for (int q = lane; q < G; q += 64) {
                unsigned long long b; int spins = 0;
                while ((b = __hip_atomic_load(sl + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == SENT) {
                    if (++spins > SPIN_MAX) { fail = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                acc += __longlong_as_double(b);
            }
            acc = wave_sum(acc);
            if (__any(fail)) { if (lane == 0) { s_fail = 1; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
            if (lane == 0) s_h = acc;
        }
        __syncthreads();
        if (s_fail) return;
        const double h = s_h;
        if (j < nv) {
#pragma unroll
            for (int u = 0; u < EP; ++u) { w[u].x = fma(-h, vc[u].x, w[u].x); w[u].y = fma(-h, vc[u].y, w[u].y); }
            if (blockIdx.x == 0 && t == 0) h_out[j] = h;
        } else {
            const double nrm = sqrt(h), inv = 1.0 / nrm;
#pragma unroll
            for (int u = 0; u < EP; ++u) {
                const int64_t i = g + u * stride;
                if (i < n2) reinterpret_cast<double2 *>(v_out)[i] = make_double2(w[u].x * inv, w[u].y * inv);
            }
            if (blockIdx.x == 0 && t == 0) h_out[nv] = nrm;
        }
    }
}

// variant B: a dedicated polling wave (the last one) that owns no elements, so its polls do not queue behind the
// prefetch of v_{j+1};  TWO = 1: two-level gather — workgroup 0 sums the partials and publishes h in one more slot
template <int NWD, int EP, int TWO, int THR = 0, int NB = 1, int BST = 512>
__global__ __launch_bounds__((NWD + 1) * 64) void mgs_persist2_kernel(const double *__restrict__ w_in, const double *__restrict__ V, int64_t ld,
                                                         int nv, double *__restrict__ v_out, double *__restrict__ h_out,
                                                         unsigned long long *slots, int *err, int64_t n)
{
    constexpr int NT = NWD * 64;
    constexpr int KQ = 8;                           // G <= 512
    __shared__ double s_red[NWD];
    __shared__ double s_h;
    __shared__ int s_fail;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const bool poller = wv == NWD;
    const int G = gridDim.x;
    const int64_t n2 = poller ? 0 : n / 2;
    const int64_t stride = (int64_t)G * NT;
    const int64_t g = (int64_t)blockIdx.x * NT + t;
    double2 w[EP], vc[EP], vn[EP];
    if (t == 0) s_fail = 0;
#pragma unroll
    for (int u = 0; u < EP; ++u) {
        const int64_t i = g + u * stride;
        w[u] = i < n2 ? reinterpret_cast<const double2 *>(w_in)[i] : make_double2(0.0, 0.0);
        vn[u] = (i < n2 && nv > 0) ? reinterpret_cast<const double2 *>(V)[i] : make_double2(0.0, 0.0);
    }
    for (int j = 0; j <= nv; ++j) {
        double s0 = 0.0, s1 = 0.0;
        if (!poller) {
#pragma unroll
            for (int u = 0; u < EP; ++u) {
                vc[u] = j < nv ? vn[u] : w[u];
                s0 = fma(w[u].x, vc[u].x, s0); s1 = fma(w[u].y, vc[u].y, s1);
            }
            if (j + 1 < nv) {
                const double2 *vp = reinterpret_cast<const double2 *>(V + (int64_t)(j + 1) * ld);
#pragma unroll
                for (int u = 0; u < EP; ++u) { const int64_t i = g + u * stride; vn[u] = i < n2 ? vp[i] : make_double2(0.0, 0.0); if (THR) __builtin_amdgcn_s_sleep(THR); }
            }
            const double ws = wave_sum(s0 + s1);
            if (lane == 0) s_red[wv] = ws;
        }
        __syncthreads();
        unsigned long long *sl = TWO == 3 ? slots + (size_t)j * NB * G : slots + (size_t)j * (G + 8);
        if (poller) {
            if (TWO == 3) {
                double bs = 0.0;
#pragma unroll
                for (int q = 0; q < NWD; ++q) bs += s_red[q];
                unsigned long long bits = __double_as_longlong(bs);
                if (bits == SENT) bits = 0x7FF8000000000000ull;
                if (lane < NB) __hip_atomic_store(sl + (size_t)lane * G + blockIdx.x, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sl += (size_t)(blockIdx.x % NB) * G;
            } else
            if (lane == 0) {
                double bs = 0.0;
#pragma unroll
                for (int q = 0; q < NWD; ++q) bs += s_red[q];
                unsigned long long bits = __double_as_longlong(bs);
                if (bits == SENT) bits = 0x7FF8000000000000ull;
                __hip_atomic_store(sl + blockIdx.x, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            double acc = 0.0; bool fail = false;
            if (!TWO || TWO == 3 || blockIdx.x == 0 || (TWO == 2 && blockIdx.x < 8)) {
                // KQ slots per lane, all loads in flight at once; only the still-pending ones are asked for again
                unsigned long long b[KQ];
#pragma unroll
                for (int u = 0; u < KQ; ++u) b[u] = SENT;
                int spins = 0;
                for (;;) {
                    bool pend = false;
#pragma unroll
                    for (int u = 0; u < KQ; ++u) {
                        const int q = lane + u * 64;
                        if (q < G && b[u] == SENT) b[u] = __hip_atomic_load(sl + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int u = 0; u < KQ; ++u) pend |= (lane + u * 64 < G) && b[u] == SENT;
                    if (!__any(pend)) break;
                    if (++spins > SPIN_MAX) { fail = true; break; }
                }
#pragma unroll
                for (int u = 0; u < KQ; ++u) if (lane + u * 64 < G) acc += __longlong_as_double(b[u]);
                acc = wave_sum(acc);
                if (TWO == 3) {
                } else if (TWO && NB > 1) {
                    unsigned long long bits = __double_as_longlong(acc);
                    if (bits == SENT) bits = 0x7FF8000000000000ull;
                    if (lane < NB) __hip_atomic_store(slots + (size_t)64 * (G + 8) + (size_t)lane * BST + j, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (TWO && lane == 0) {
                    unsigned long long bits = __double_as_longlong(acc);
                    if (bits == SENT) bits = 0x7FF8000000000000ull;
                    __hip_atomic_store(sl + G + (TWO == 2 ? blockIdx.x : 0), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                unsigned long long b; int spins = 0;
                const unsigned long long *bw = NB > 1 ? slots + (size_t)64 * (G + 8) + (size_t)(blockIdx.x % NB) * BST + j : sl + G + (TWO == 2 ? (blockIdx.x & 7) : 0);
                while ((b = __hip_atomic_load(bw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == SENT) {
                    if (++spins > SPIN_MAX) { fail = true; break; }
                }
                acc = __longlong_as_double(b);
            }
            if (__any(fail)) { if (lane == 0) { s_fail = 1; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
            if (lane == 0) s_h = acc;
        }
        __syncthreads();
        if (s_fail) return;
        const double h = s_h;
        if (j < nv) {
#pragma unroll
            for (int u = 0; u < EP; ++u) { w[u].x = fma(-h, vc[u].x, w[u].x); w[u].y = fma(-h, vc[u].y, w[u].y); }
            if (blockIdx.x == 0 && t == 0) h_out[j] = h;
        } else {
            const double nrm = sqrt(h), inv = 1.0 / nrm;
#pragma unroll
            for (int u = 0; u < EP; ++u) {
                const int64_t i = g + u * stride;
                if (i < n2) reinterpret_cast<double2 *>(v_out)[i] = make_double2(w[u].x * inv, w[u].y * inv);
            }
            if (blockIdx.x == 0 && t == 0) h_out[nv] = nrm;
        }
    }
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 128ll * 128 * 128;
    const int nv = argc > 2 ? atoi(argv[2]) : 26;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    const int64_t ld = (n + 31) & ~31ll;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("n = %lld, nv = %d, CUs = %d\n", (long long)n, nv, cus);
    std::vector<double> hV((size_t)ld * (nv + 1)), hw(n);
    uint64_t z = 12345;
    auto rnd = [&]() { z = z * 6364136223846793005ull + 1442695040888963407ull; return ((z >> 11) * (1.0 / 9007199254740992.0)) - 0.5; };
    for (auto &v : hV) v = rnd() * 2e-3;
    for (auto &v : hw) v = rnd();
    double *V, *w, *hout; unsigned long long *slots; int *err;
    CK(hipMalloc(&V, hV.size() * 8)); CK(hipMalloc(&w, n * 8)); CK(hipMalloc(&hout, (nv + 2) * 8));
    const size_t slot_bytes = (size_t)(nv + 1) * 64 * 512 * 8 + ((size_t)64 * 1032 + 64 * 512) * 8;
    CK(hipMalloc(&slots, slot_bytes)); CK(hipMalloc(&err, 4));
    CK(hipMemcpy(V, hV.data(), hV.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(err, 0, 4));
    // host reference (same MGS, plain double)
    std::vector<double> rw(hw), rh(nv + 1);
    for (int j = 0; j < nv; ++j) {
        long double s = 0; const double *vj = hV.data() + (size_t)j * ld;
        for (int64_t i = 0; i < n; ++i) s += (long double)rw[i] * vj[i];
        rh[j] = (double)s;
        for (int64_t i = 0; i < n; ++i) rw[i] -= rh[j] * vj[i];
    }
    { long double s = 0; for (int64_t i = 0; i < n; ++i) s += (long double)rw[i] * rw[i]; rh[nv] = std::sqrt((double)s); }

    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int NT, int EP, int G, const char *name, int extra = 0) {
        if ((int64_t)G * NT * EP * 2 < n) { printf("%-28s does not hold n\n", name); return; }
        int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NT + extra, 0));
        if ((int64_t)occ * cus < G) { printf("%-28s grid %d not co-resident (occ %d)\n", name, G, occ); return; }
        double *vout = V + (size_t)nv * ld;
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipMemsetAsync(slots, 0xFF, slot_bytes, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(kern, dim3(G), dim3(NT + extra), 0, st, w, V, ld, nv, vout, hout, slots, err, n);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) { best = std::min(best, ms); sum += ms; }
        }
        int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        std::vector<double> gh(nv + 1); CK(hipMemcpy(gh.data(), hout, (nv + 1) * 8, hipMemcpyDeviceToHost));
        double maxrel = 0; for (int j = 0; j <= nv; ++j) maxrel = std::max(maxrel, std::fabs(gh[j] - rh[j]) / (std::fabs(rh[j]) + 1e-300));
        std::vector<double> gv(n); CK(hipMemcpy(gv.data(), vout, n * 8, hipMemcpyDeviceToHost));
        double maxv = 0; for (int64_t i = 0; i < n; ++i) maxv = std::max(maxv, std::fabs(gv[i] - rw[i] / rh[nv]));
        printf("%-28s G %4d: avg %.1f us  best %.1f us  = %.2f us/step (nv+1 = %d)  err %d  max rel dh %.2e  max dv %.2e\n", name, G,
               sum / reps * 1e3, best * 1e3, sum / reps * 1e3 / (nv + 1), nv + 1, herr, maxrel, maxv);
    };
    run(mgs_persist_kernel<1024, 4>, 1024, 4, cus, "persist<1024,4>");
    run(mgs_persist_kernel<512, 8>, 512, 8, cus, "persist<512,8>");
    run(mgs_persist_kernel<256, 4>, 256, 4, cus * 4, "persist<256,4> x4");
    run(mgs_persist_kernel<512, 4>, 512, 4, cus * 2, "persist<512,4> x2");
    run(mgs_persist2_kernel<15, 5, 0>, 960, 5, cus, "persist2<15,5> all-to-all", 64);
    run(mgs_persist2_kernel<15, 5, 1>, 960, 5, cus, "persist2<15,5> two-level", 64);
    if (n <= 131072) {
        // few workgroups: is the grid-wide sum cheaper with fewer participants (short vectors)?
        run(mgs_persist2_kernel<15, 5, 1, 0, 4>, 960, 5, 8, "persist2 G=8 nb4", 64);
        run(mgs_persist2_kernel<15, 5, 1, 0, 8>, 960, 5, 16, "persist2 G=16 nb8", 64);
        run(mgs_persist2_kernel<15, 2, 1, 0, 16>, 960, 2, 32, "persist2 G=32 nb16", 64);
        run(mgs_persist2_kernel<15, 1, 1, 0, 16>, 960, 1, 64, "persist2 G=64 nb16", 64);
        run(mgs_persist2_kernel<15, 1, 1, 0, 16>, 960, 1, 128, "persist2 G=128 nb16", 64);
    }
    run(mgs_persist2_kernel<15, 5, 3, 0, 8>, 960, 5, cus, "persist2<15,5> all-to-all r8", 64);
    run(mgs_persist2_kernel<15, 5, 3, 0, 16>, 960, 5, cus, "persist2<15,5> all-to-all r16", 64);
    run(mgs_persist2_kernel<15, 5, 3, 0, 32>, 960, 5, cus, "persist2<15,5> all-to-all r32", 64);
    run(mgs_persist2_kernel<15, 5, 3, 0, 64>, 960, 5, cus, "persist2<15,5> all-to-all r64", 64);
    run(mgs_persist2_kernel<15, 5, 1, 0, 4>, 960, 5, cus, "persist2<15,5> two-level nb4", 64);
    run(mgs_persist2_kernel<15, 5, 1, 0, 8>, 960, 5, cus, "persist2<15,5> two-level nb8", 64);
    run(mgs_persist2_kernel<15, 5, 1, 0, 16>, 960, 5, cus, "persist2<15,5> two-level nb16", 64);
    run(mgs_persist2_kernel<15, 5, 1, 0, 32>, 960, 5, cus, "persist2<15,5> two-level nb32", 64);
    run(mgs_persist2_kernel<15, 5, 1, 0, 64>, 960, 5, cus, "persist2<15,5> two-level nb64", 64);
    run(mgs_persist2_kernel<15, 5, 1, 0, 16, 64>, 960, 5, cus, "persist2<15,5> 2-lvl nb16 st64", 64);
    run(mgs_persist2_kernel<8, 9, 0>, 512, 9, cus, "persist2<8,9> all-to-all", 64);
    run(mgs_persist2_kernel<8, 9, 1>, 512, 9, cus, "persist2<8,9> two-level", 64);
    run(mgs_persist2_kernel<7, 5, 0>, 448, 5, 2 * cus, "persist2<7,5> x2 all-to-all", 64);
    run(mgs_persist2_kernel<7, 5, 1>, 448, 5, 2 * cus, "persist2<7,5> x2 two-level", 64);
    return 0;
}
