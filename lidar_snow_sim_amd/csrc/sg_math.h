// sg_math.h -- device math that has to agree bit-for-bit with what NumPy/glibc compute on the host.
// Compiled with -ffp-contract=off: NumPy never fuses a multiply into an add, so neither may we.
#pragma once
#include <hip/hip_runtime.h>
#include "sg_common.h"

// ---------------------------------------------------------------------------------------------
// float32 atan2: the fdlibm algorithm glibc 2.35 ships as atan2f (e_atan2f.c / s_atanf.c).
// simulation.py:91 calls np.arctan2 on float32 columns; NumPy's portable loop is glibc's atan2f,
// and one float32 ULP of beam azimuth (2.4e-7 rad) is enough to move occlusion ratios by 1e-4 of
// a beam -- so this is restated operation by operation instead of calling OCML's atan2f.
// Verified against glibc on 6e7 inputs (0 mismatches) in the build container.
//
// The constants and the branch structure of sg_atanf / sg_atan2f, and the __kernel_sin / __kernel_cos polynomials used by
// sg_tan_0_2pi below, are fdlibm's (a third-party algorithm, not part of the reference), whose notice reads:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
__device__ __forceinline__ float sg_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    int hx = __float_as_int(x);
    int ix = hx & 0x7fffffff;
    int id;
    float hi = 0.f, lo = 0.f;
    if (ix >= 0x4c000000) { /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) { /* |x| < 0.4375 */
        if (ix < 0x31000000) return x; /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); hi = atanhi[0]; lo = atanlo[0]; }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); hi = atanhi[1]; lo = atanlo[1]; }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); hi = atanhi[2]; lo = atanlo[2]; }
            else { id = 3; x = -1.0f / x; hi = atanhi[3]; lo = atanlo[3]; }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -z : z;
}

__device__ __forceinline__ float sg_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    int hx = __float_as_int(x), hy = __float_as_int(y);
    int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return sg_atanf(y);
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        const float pi_o_4 = 7.8539818525e-01f;
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = sg_atanf(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return __int_as_float(__float_as_int(z) ^ 0x80000000);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// ---------------------------------------------------------------------------------------------
// np.add.reduce over a short float64 vector, streamed: 0.0 + pairwise_sum(a, n) for n <= 128
// (numpy/_core/src/umath/loops_utils.h.src).  For n < 8 that is a left-to-right sum; for
// 8 <= n <= 128 the first 8*floor(n/8) values go to eight interleaved accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the remaining n%8 values are added left to right.
// The reference reaches it through `diffs[assignment == j].sum()` (simulation.py:289, :292).
// The values of the current, not yet complete block of 8 sit in a shift register so that every
// register index is static (no scratch).
struct SgNpSum {
    double r0, r1, r2, r3, r4, r5, r6, r7;
    double v0, v1, v2, v3, v4, v5, v6, v7;   // v0 = newest
    int c;        // values in the open block
    int blocks;   // completed blocks
    __device__ __forceinline__ void reset() { c = 0; blocks = 0; }
    __device__ __forceinline__ void push(double v)
    {
        v7 = v6; v6 = v5; v5 = v4; v4 = v3; v3 = v2; v2 = v1; v1 = v0; v0 = v;
        if (++c == 8) {
            if (blocks == 0) { r0 = v7; r1 = v6; r2 = v5; r3 = v4; r4 = v3; r5 = v2; r6 = v1; r7 = v0; }
            else { r0 += v7; r1 += v6; r2 += v5; r3 += v4; r4 += v3; r5 += v2; r6 += v1; r7 += v0; }
            c = 0;
            ++blocks;
        }
    }
    __device__ __forceinline__ double result() const
    {
        double res = blocks ? ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) : -0.0;
        if (c >= 7) res += v6;
        if (c >= 6) res += v5;
        if (c >= 5) res += v4;
        if (c >= 4) res += v3;
        if (c >= 3) res += v2;
        if (c >= 2) res += v1;
        if (c >= 1) res += v0;
        return 0.0 + res;
    }
};

__device__ __forceinline__ double sg_clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }

// geometry.py:68-70, :219-221
__device__ __forceinline__ bool sg_forward(double ray, double centre)
{
    double d = ray - centre;
    return (fabs(d) < SG_PI / 2) || (fabs(d - SG_TWO_PI) < SG_PI / 2) || (fabs(d + SG_TWO_PI) < SG_PI / 2);
}

// simulation.py:553-569; the float32 twin follows NumPy-2 scalar promotion (NEP 50): Python floats
// next to an np.float32 range are computed in float32.
__device__ __forceinline__ double sg_xsi(double R)
{
    if (R <= 0.9) return 0.0;
    if (R >= 1.0) return 1.0;
    double m = (1 - 0) / (1.0 - 0.9);
    double b = 0 - (m * 0.9);
    return m * R + b;
}
__device__ __forceinline__ double sg_xsi(float R)
{
    if (R <= (float)0.9) return 0.0;
    if (R >= (float)1.0) return 1.0;
    double m = (1 - 0) / (1.0 - 0.9);
    double b = 0 - (m * 0.9);
    float y = (float)m * R;
    y = y + (float)b;
    return (double)y;
}

// ---------------------------------------------------------------------------------------------
// tan(theta) for theta in [0, 2 pi] (beam-limit slopes, geometry.py:94): quadrant reduction against pi/2 in
// three parts, sine and cosine polynomials on [-pi/4, pi/4] (fdlibm's kernel coefficients), one division.
// About 40 float64 instructions instead of the device math library's general-range tan, within 2 ULP of it;
// the slope only enters the point-line distance test (geometry.py:131-135), where a last-bit difference moves the
// distance by 1e-16 relative -- the same order as glibc's tan vs any other libm.
__device__ __forceinline__ double sg_tan_0_2pi(double theta)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;   // fdlibm split of pi/2
    const double PIO2_2 = 6.07710050630396597660e-11, PIO2_2T = 2.02226624879595063154e-21;
    const double fn = rint(theta * TWO_OVER_PI);                 // 0 .. 4
    const int n = (int)fn;
    double r = theta - fn * PIO2_1;
    double w = fn * PIO2_1T;
    double x = r - w;
    // second step keeps ~118 bits of pi/2: enough for every theta this close to a multiple of pi/2
    {
        const double t = r;
        w = fn * PIO2_2;
        r = t - w;
        w = fn * PIO2_2T - ((t - r) - w);
        x = r - w;
    }
    const double tail = (r - x) - w;
    const double z = x * x;
    // __kernel_sin
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * x;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sn = x - ((z * (0.5 * tail - v * rs) - tail) - v * S1);
    // __kernel_cos
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double ww = 1.0 - hz;
    const double cs = ww + (((1.0 - ww) - hz) + (z * rc - x * tail));
    // tan(x + n pi/2) = sin/cos for even n, -cos/sin for odd n
    return (n & 1) ? -(cs / sn) : (sn / cs);
}

// sin and cos of theta in [0, 2 pi] to ~2e-16 absolute (two-part quadrant reduction, the same kernel coefficients, no
// tail terms): the FAST form of the beam-limit test (sg_beam.h: sg_flake_hits) needs the limit rays' directions only to
// 1e-13; whatever it cannot decide within that goes to the reference's own expression.
__device__ __forceinline__ void sg_sincos_0_2pi(double theta, double &s, double &c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
    const double fn = rint(theta * TWO_OVER_PI);
    const int n = (int)fn;
    const double x = (theta - fn * PIO2_1) - fn * PIO2_1T;
    const double z = x * x;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double sn = x + (x * z) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double cs = (1.0 - 0.5 * z) + (z * z) * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double a = (n & 1) ? cs : sn, b = (n & 1) ? sn : cs;      // sin(x + n pi/2), cos(x + n pi/2)
    s = (n & 2) ? -a : a;
    c = ((n + 1) & 2) ? -b : b;
}
