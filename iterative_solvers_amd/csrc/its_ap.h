// its_ap.h — adaptive-precision storage tiers (ITS_FMT_MIXED): which precision an entry of the matrix is
// STORED in, decided by its magnitude against one or two thresholds as the reference's USE_AP modes do (config.mk:56-63:
// AP_VALUE_TYPE "ap[dp_sp]" / "ap[dp_hp]" / "ap[sp_hp]" / "ap[dp_sp_hp]", AP_THRESHOLD_1, AP_THRESHOLD_2;
// main.cpp:171-184 hands them to partition_precisions, whose source — the USpMV library — is not in the reference tree:
// the split below follows the parameter names and the dp_percent / sp_percent / hp_percent report of main.cpp:186-190,
// parity unpinned).  Shared by the host passes (its_host.cpp, g++) and the device builders (hipcc).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ITS_AP_FN __host__ __device__ inline
#else
#define ITS_AP_FN inline
#endif

namespace its_ap {

enum { TIER_FP64 = 0, TIER_FP32 = 1, TIER_FP16 = 2, TIER_NONE = 3 };

// mode: ITS_AP_DP_SP 0, ITS_AP_DP_HP 1, ITS_AP_SP_HP 2, ITS_AP_DP_SP_HP 3 (include/its_hip.h).  An entry at or above
// AP_THRESHOLD_1 takes the mode's first type; below it the second; in the three-type mode below AP_THRESHOLD_2 the
// third.  An entry the chosen type cannot hold (beyond 65504 for fp16, beyond FLT_MAX for fp32; NaN) stays one type up.
ITS_AP_FN int tier(double v, int mode, double t1, double t2)
{
    const double a = fabs(v);
    int t;
    if (!(a < t1)) t = mode == 2 ? TIER_FP32 : TIER_FP64;                 // (NaN lands here)
    else if (mode == 0) t = TIER_FP32;
    else if (mode == 1 || mode == 2) t = TIER_FP16;
    else t = !(a < t2) ? TIER_FP32 : TIER_FP16;
    if (t == TIER_FP16 && !(a <= 65504.0)) t = TIER_FP32;
    if (t == TIER_FP32 && !(a <= 3.4028234663852886e38)) t = TIER_FP64;
    return t;
}

// float -> IEEE binary16 bits, round to nearest even (what v_cvt_f16_f32 does): the host's counterpart of (_Float16)f
ITS_AP_FN uint16_t half_bits(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));   // inf / NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);              // rounds to beyond 65504: inf
    if (x < 0x33000001u) return (uint16_t)sign;                           // below half of the smallest subnormal: 0
    const int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;                                                            // bits dropped from the 24-bit significand
    uint32_t base;
    if (e >= -14) { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    else { shift = 13 + (-14 - e); base = 0; }
    uint32_t h = base + (m >> shift);
    const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (h & 1u))) ++h;               // (a carry into the exponent is the right answer)
    return (uint16_t)(sign | h);
}

// binary16 bits -> double, exactly
ITS_AP_FN double half_value(uint16_t hb)
{
    const int s = hb >> 15, e = (hb >> 10) & 31, m = hb & 1023;
    double v;
    if (e == 0) v = ldexp((double)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexp((double)(m | 1024), e - 25);
    return s ? -v : v;
}

// the value an entry contributes after storage in its tier: fp32 = (float)v; fp16 = the float (float)v rounded to binary16
ITS_AP_FN double stored_value(double v, int t)
{
    if (t == TIER_FP32) return (double)(float)v;
    if (t == TIER_FP16) return half_value(half_bits((float)v));
    return v;
}

}  // namespace its_ap
