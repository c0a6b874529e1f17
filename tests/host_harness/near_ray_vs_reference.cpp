// Host harness: the fast form of the beam-limit distance test (sg_beam.h: sg_near_ray -- |y cos - x sin| against the radius, the
// reference's expression only inside a narrow band around equality) against the reference's expression everywhere
// (sg_near_ray_reference: geometry.py:94-106, :131-135 -- slope by tangent, root, quotient), on flakes placed ADVERSARIALLY:
// at distance r (1 +- eps) from a limit ray, eps log-uniform over 1e-17 .. 1e-4, rays at random azimuths and at / next to the
// quadrant boundaries (pi/2, 3 pi/2: the vertical special case; 0, pi, 2 pi), float32-valued azimuths as the float32 rows give
// them and float64 ones, narrow and wide beams.  Every decision must be the reference expression's; the share of tests that
// needed the reference's expression is printed (the band must stay a rare path on ordinary input).
// usage: near_ray_vs_reference [tests per class]; exit status 1 on any mismatch.  Built and run by tests/test_kernel_math.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <cmath>
__host__ inline int __double2hiint(double x) { unsigned long long u; memcpy(&u, &x, 8); return (int)(u >> 32); }
__host__ inline double __hiloint2double(int hi, int lo) { unsigned long long u = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; double x; memcpy(&x, &u, 8); return x; }
__host__ inline int __float_as_int(float x) { int i; memcpy(&i, &x, 4); return i; }
__host__ inline float __int_as_float(int i) { float x; memcpy(&x, &i, 4); return x; }
__host__ inline unsigned __float_as_uint(float x) { unsigned i; memcpy(&i, &x, 4); return i; }
__host__ inline float __uint_as_float(unsigned i) { float x; memcpy(&x, &i, 4); return x; }
__host__ inline long long __double_as_longlong(double x) { long long i; memcpy(&i, &x, 8); return i; }
__host__ inline double __longlong_as_double(long long i) { double x; memcpy(&x, &i, 8); return x; }
__host__ inline int __double2loint(double x) { unsigned long long u; memcpy(&u, &x, 8); return (int)(u & 0xffffffffu); }
template <typename T> __host__ inline T __shfl(T v, int) { return v; }
template <typename T> __host__ inline T __shfl_up(T v, int) { return v; }
template <typename T> __host__ inline T __shfl_down(T v, int) { return v; }
template <typename T> __host__ inline T __shfl_xor(T v, int) { return v; }
__host__ inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
__host__ inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
__host__ inline int __ffsll(long long v) { return __builtin_ffsll(v); }
__host__ inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
__host__ inline int atomicOr(int *p, int v) { int o = *p; *p |= v; return o; }
struct { unsigned x = 0, y = 0, z = 0; } threadIdx_host;
#undef __device__
#define __device__
#include "sg_beam.h"

struct Tally { long tests = 0, mismatches = 0, band = 0, libm_differs = 0; double worst = 0.0; };

// the reference's quotient itself (what sg_near_ray_reference compares)
static double reference_quotient(double theta, double fx, double fy)
{
    double a, b;
    if (theta == SG_PI / 2 || theta == 3 * SG_PI / 2) { a = 1.0; b = 0.0; } else { a = -tan(theta); b = 1.0; }
    return fabs((fx * a + fy * b) + 0.0) / sqrt(a * a + b * b);
}

// one beam azimuth -> its two limit rays as the kernels derive them; flakes around both
static void one_beam(double theta_c, double div_deg, std::mt19937_64 &rng, int flakes, Tally &t, double eps_lo, double eps_hi)
{
    std::uniform_real_distribution<double> U(0.0, 1.0);
    SgBeamGeo g{};
    g.theta_c = theta_c;
    sg_geo_limits(g, div_deg);
    // the directions exactly as sg_beam_geometry makes them
    double sc, cc, sh, ch;
    sg_sincos_0_2pi(g.theta_c, sc, cc);
    const double h = (div_deg / 2) * (SG_PI / 180.0);
    if (h < 0.015625) {
        const double q = h * h;
        sh = h * (1.0 - q * (1.0 / 6) * (1.0 - q * (1.0 / 20) * (1.0 - q * (1.0 / 42))));
        ch = 1.0 - q * 0.5 * (1.0 - q * (1.0 / 12) * (1.0 - q * (1.0 / 30)));
    } else sg_sincos_0_2pi(h, sh, ch);
    g.sr = sc * ch - cc * sh; g.cr = cc * ch + sc * sh;
    g.sl = sc * ch + cc * sh; g.cl = cc * ch - sc * sh;
    for (int side = 0; side < 2; ++side) {
        const double th = side ? g.theta_l : g.theta_r, s = side ? g.sl : g.sr, c = side ? g.cl : g.cr;
        const long double ls = sinl((long double)th), lc = cosl((long double)th);
        for (int i = 0; i < flakes; ++i) {
            const double fr = std::exp(std::log(2e-5) + U(rng) * (std::log(0.05) - std::log(2e-5)));     // 20 um .. 5 cm
            long double along = 0.3L + 119.7L * U(rng);
            if (U(rng) < 0.1) along = -along;                                                           // behind the sensor: the test is about the LINE
            const double eps = std::exp(std::log(eps_lo) + U(rng) * (std::log(eps_hi) - std::log(eps_lo)));
            const long double off = (long double)fr * (1.0L + (U(rng) < 0.5 ? -1.0L : 1.0L) * (long double)eps) * (U(rng) < 0.5 ? -1.0L : 1.0L);
            const double fx = (double)(along * lc - off * ls), fy = (double)(along * ls + off * lc);
            bool und0 = false, und = false;
            const bool fast = sg_near_ray<false>(th, s, c, fx, fy, fr, false, und0);
            const bool deferred = sg_near_ray<true>(th, s, c, fx, fy, fr, false, und);      // the pass over all rows: decides, or reports `undecided`
            const bool ref = sg_near_ray_reference(th, fx, fy, fr, false);
            const bool ref_libm = sg_near_ray_reference(th, fx, fy, fr, true);
            const double dist = fabs(fy * c - fx * s);
            ++t.tests;
            const double gap = fabs(dist - reference_quotient(th, fx, fy)) / (fabs(fx) + fabs(fy));
            if (gap > t.worst) t.worst = gap;
            if (und) ++t.band;
            if (fast != ref || (!und && deferred != ref)) {
                if (t.mismatches < 10) printf("  MISMATCH theta %.17g fx %.17g fy %.17g r %.17g: fast %d reference %d (dist %.17g)\n", th, fx, fy, fr, (int)fast, (int)ref, dist);
                ++t.mismatches;
            }
            if (ref != ref_libm) ++t.libm_differs;
        }
    }
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 20000;
    std::mt19937_64 rng(20260927);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    int rc = 0;
    const double divs[3] = {0.1718873385392, 0.003, 44.0};
    // class 1: random azimuths (float32-valued as sg_beam_geometry<float> gives them, and float64), eps over the whole range
    // class 2: azimuths at and next to the quadrant boundaries
    // class 3: ordinary input -- flakes anywhere within 3 radii of the ray (how often does the band decide?)
    for (int cls = 1; cls <= 3; ++cls) {
        Tally t;
        for (long b = 0; b < n; ++b) {
            double th;
            if (cls == 2) {
                const double base = (SG_PI / 2) * (double)(rng() % 5);
                const int kind = (int)(rng() % 4);
                if (kind == 0) th = base;
                else if (kind == 1) th = (double)(float)base;
                else if (kind == 2) { th = base; for (int k = (int)(rng() % 40); k > 0; --k) th = nextafter(th, (rng() & 1) ? 10.0 : -10.0); }
                else th = (double)nextafterf((float)base, (rng() & 1) ? 10.0f : -10.0f);
                // ... and azimuths whose LIMIT ray lands on the boundary
                if (rng() % 3 == 0) th += ((rng() & 1) ? 1.0 : -1.0) * (divs[b % 3] / 2) * (SG_PI / 180.0);
                if (th < 0) th += SG_TWO_PI;
                if (th > SG_TWO_PI) th -= SG_TWO_PI;
            } else {
                th = U(rng) * SG_TWO_PI;
                if (b & 1) th = (double)(float)th;
                if (th > SG_TWO_PI) th = SG_TWO_PI;
            }
            if (cls == 3) one_beam(th, divs[0], rng, 8, t, 1e-3, 2.0);
            else one_beam(th, divs[b % 3], rng, 8, t, 1e-17, 1e-4);
        }
        printf("near<class %d>: %ld tests, %ld mismatches, %.3g of them decided by the reference's expression, libm tangent differs in %ld; largest |fast - quotient| / (|x| + |y|) = %.3g (band 1e-12)\n", cls, t.tests, t.mismatches,
               (double)t.band / (double)t.tests, t.libm_differs, t.worst);
        if (t.worst > 1e-14) { printf("  the two distances differ by more than 1e-14 (|x| + |y|): the band's margin is gone\n"); rc = 1; }
        if (t.mismatches) rc = 1;
        if (cls == 3 && t.band * 100000 > t.tests) { printf("  the band decided more than 1e-5 of ordinary tests\n"); rc = 1; }
    }
    return rc;
}
