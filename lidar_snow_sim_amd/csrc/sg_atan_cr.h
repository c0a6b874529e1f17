// sg_atan_cr.h -- float64 atan2 / atan rounded correctly (to nearest), for the two places where the reference's float64
// arctan2 / arctan feed a decision: the beam azimuth of float64 rows (simulation.py:91) and the per-flake azimuth and tangent
// angles of a table filed on the device (simulation.py:351; geometry.py:50).  NumPy's portable loops call glibc there, whose
// atan2 / atan are correctly rounded in practice (its IBM Accurate Mathematical Library code falls back to a second,
// ~100-bit stage whenever the first cannot decide the rounding).  The device math library's versions differ from it in the
// last bit for some inputs; this one evaluates atan in double-double arithmetic (~2^-100 relative) and rounds once, so it
// returns the same double as glibc wherever glibc is correctly rounded -- scripts/probe/atan_cr_check.cpp compares the two (2 * 10^8
// inputs once in the build container; 4 * 10^6 on every run of tests/test_host_logic.py::
// test_correctly_rounded_atan2_equals_glibc_where_glibc_is, which settles each mismatch with 70-digit arithmetic).
//
//   atan2(y, x): t = min(|y|, |x|) / max(|y|, |x|) in double-double; c = round(64 t) / 64; u = (t - c) / (1 + t c), |u| <= 2^-7;
//   atan(t) = atan(c) [table, double-double] + u (1 - u^2/3 + u^4/5 - u^6/7 [double-double] + u^8/9 - .. + u^16/17 [double]);
//   then the octant / quadrant reflections with pi/2 and pi as double-doubles.
// Host and device code alike (the check above compiles it with g++).
#pragma once
#include <math.h>
#include <stdint.h>
#include "sg_atan_table.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define SG_CR_FN __host__ __device__ static inline
#else
#define SG_CR_FN static inline
#endif

struct SgDD { double hi, lo; };

SG_CR_FN SgDD sg_dd_two_sum(double a, double b)            // a + b exactly
{
    const double s = a + b, bb = s - a;
    return SgDD{s, (a - (s - bb)) + (b - bb)};
}
SG_CR_FN SgDD sg_dd_quick(double a, double b)              // |a| >= |b|
{
    const double s = a + b;
    return SgDD{s, b - (s - a)};
}
SG_CR_FN SgDD sg_dd_two_prod(double a, double b)           // a * b exactly
{
    const double p = a * b;
    return SgDD{p, fma(a, b, -p)};
}
SG_CR_FN SgDD sg_dd_add(SgDD a, SgDD b)
{
    SgDD s = sg_dd_two_sum(a.hi, b.hi);
    const SgDD t = sg_dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = sg_dd_quick(s.hi, s.lo);
    s.lo += t.lo;
    return sg_dd_quick(s.hi, s.lo);
}
SG_CR_FN SgDD sg_dd_neg(SgDD a) { return SgDD{-a.hi, -a.lo}; }
SG_CR_FN SgDD sg_dd_mul(SgDD a, SgDD b)
{
    SgDD p = sg_dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return sg_dd_quick(p.hi, p.lo);
}
SG_CR_FN SgDD sg_dd_mul_d(SgDD a, double b)
{
    SgDD p = sg_dd_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return sg_dd_quick(p.hi, p.lo);
}
SG_CR_FN SgDD sg_dd_div(SgDD a, SgDD b)                    // a / b, three quotient digits
{
    const double q1 = a.hi / b.hi;
    SgDD r = sg_dd_add(a, sg_dd_neg(sg_dd_mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = sg_dd_add(r, sg_dd_neg(sg_dd_mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    SgDD q = sg_dd_quick(q1, q2);
    return sg_dd_add(q, SgDD{q3, 0.0});
}

// atan(t) for a double-double 0 <= t <= 1
SG_CR_FN SgDD sg_dd_atan01(SgDD t)
{
    const double tab[65][2] = SG_ATAN_TABLE_INIT;
    int k = (int)floor(t.hi * 64.0 + 0.5);
    if (k < 0) k = 0;
    if (k > 64) k = 64;
    const double c = (double)k * 0.015625;                  // exact
    SgDD u;
    if (k == 0) u = t;
    else {
        const SgDD num = sg_dd_add(t, SgDD{-c, 0.0});
        const SgDD den = sg_dd_add(sg_dd_mul_d(t, c), SgDD{1.0, 0.0});
        u = sg_dd_div(num, den);
    }
    const SgDD s = sg_dd_mul(u, u);
    const double z = s.hi;
    // u^8/9 - u^10/11 + u^12/13 - u^14/15 + u^16/17, relative to u^8: double is enough (|u| <= 2^-7)
    const double tail = 1.0 / 9.0 + z * (-1.0 / 11.0 + z * (1.0 / 13.0 + z * (-1.0 / 15.0 + z * (1.0 / 17.0))));
    SgDD p = sg_dd_add(SgDD{-SG_SEVENTH_HI, -SG_SEVENTH_LO}, sg_dd_mul_d(s, tail));
    p = sg_dd_add(SgDD{SG_FIFTH_HI, SG_FIFTH_LO}, sg_dd_mul(s, p));
    p = sg_dd_add(SgDD{-SG_THIRD_HI, -SG_THIRD_LO}, sg_dd_mul(s, p));
    p = sg_dd_mul(s, p);                                    // atan(u) / u - 1
    const SgDD au = sg_dd_add(u, sg_dd_mul(u, p));
    return sg_dd_add(SgDD{tab[k][0], tab[k][1]}, au);
}

SG_CR_FN double sg_atan2_cr(double y, double x)
{
    if (x != x || y != y) return x + y;
    const double ax = fabs(x), ay = fabs(y);
    const bool xneg = signbit(x);
    if (ay == 0.0) return xneg ? copysign(SG_PI_HI, y) : copysign(0.0, y);
    if (ax == 0.0) return copysign(SG_PIO2_HI, y);
    if (isinf(ax) || isinf(ay)) {
        if (isinf(ax) && isinf(ay)) return copysign(xneg ? 3 * (SG_PI_HI / 4) : SG_PI_HI / 4, y);
        if (isinf(ay)) return copysign(SG_PIO2_HI, y);
        return xneg ? copysign(SG_PI_HI, y) : copysign(0.0, y);
    }
    const bool swap = ay > ax;
    double num = swap ? ax : ay, den = swap ? ay : ax;
    // keep the quotient's remainder exact: scale both away from the subnormal / overflow ends (powers of two: exact)
    if (den > 0x1p+1000) { num *= 0x1p-64; den *= 0x1p-64; }
    if (den < 0x1p-900) { num *= 0x1p+128; den *= 0x1p+128; }
    const double q = num / den;
    if (q < 0x1p-1000) {                                   // atan(t) = t to far beyond double precision
        double r = swap ? SG_PIO2_HI : q;
        if (xneg) r = swap ? SG_PIO2_HI : SG_PI_HI;         // pi - tiny, pi/2 +- tiny: round to the constant
        if (!swap && !xneg) return copysign(q, y);
        return copysign(r, y);
    }
    const double rem = fma(-q, den, num);                   // exact
    const SgDD t = sg_dd_quick(q, rem / den);
    SgDD r = sg_dd_atan01(t);
    if (swap) r = sg_dd_add(SgDD{SG_PIO2_HI, SG_PIO2_LO}, sg_dd_neg(r));
    if (xneg) r = sg_dd_add(SgDD{SG_PI_HI, SG_PI_LO}, sg_dd_neg(r));
    const double v = r.hi + r.lo;
    return signbit(y) ? -v : v;
}

SG_CR_FN double sg_atan_cr(double x) { return sg_atan2_cr(x, 1.0); }
