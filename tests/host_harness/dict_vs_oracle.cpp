// Host harness: phase 2 of the kernels (sg_beam.h: sg_beam_dict, the occlusion dict every capacity tier builds) against the oracle's
// compute_occlusion_dict (oracle/snow_oracle.c: so_occlusion_dict, pinned to the reference's golden vectors) on random interval lists,
// bit for bit -- list capacities 4, 8, 16 and 63, wrapped wedges, shared endpoints, nested intervals, owners with eight and more slots
// (NumPy's blocked sum).  The device function is compiled for the host (hipcc --cuda-host-only; host overloads of the intrinsics
// below); the oracle is linked in as the checker.  usage: dict_vs_oracle [cases per capacity]; exit status 1 on any mismatch.
// Built and run by tests/test_kernel_math.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <cmath>
// host overloads of the two device intrinsics the list cells use
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
#include "sg_few.h"



extern "C" int so_occlusion_dict(double right_angle, double left_angle, const double *intervals, int64_t L, double current_range,
                                 double beam_div_deg, int64_t *key_out, double *rj_out, double *ratio_out);

template <int LMAX>
static long run_dict(long n, unsigned long long seed)
{
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double div = 0.1718873385392;
    long bad = 0, big_owner = 0, slots = 0;
    static double g_a1[LMAX + 2], g_a2[LMAX + 2], g_rho[LMAX + 2], g_ratio[LMAX + 2], iv[3 * (LMAX + 1)], rj[LMAX + 2], ra[LMAX + 2];
    static int64_t key[LMAX + 2];
    for (long it = 0; it < n; ++it) {
        const double d = 3.0 + 100.0 * U(rng) * U(rng);
        double tc = (double)(float)(U(rng) * SG_TWO_PI);
        if (it % 200 == 0) tc = (double)(float)(U(rng) * 0.002);
        if (it % 200 == 1) tc = (double)(float)(SG_TWO_PI - U(rng) * 0.002);
        double tr, tl; sg_beam_limits(tc, div, tr, tl);
        const double half = div / 2 * (SG_PI / 180.0);
        const int L = 1 + (int)(U(rng) * LMAX) % LMAX;
        const double wscale = U(rng) < 0.3 ? 0.05 : 0.7;          // many thin flakes: owners with many slots
        for (int j = 0; j < L; ++j) {
            double c = tc + (U(rng) * 2.4 - 1.2) * half, w = half * (0.005 + wscale * U(rng) * U(rng));
            double a1 = c - w, a2 = c + w;
            if (U(rng) < 0.2) a1 = tr;
            if (U(rng) < 0.2) a2 = tl;
            if (U(rng) < 0.1 && j > 0) a1 = g_a2[j - 1];
            if (U(rng) < 0.03 && j > 0) { a1 = g_a1[j - 1]; a2 = g_a2[j - 1]; }
            if (U(rng) < 0.05) { a1 = tc - 1.3 * half; a2 = tc + 1.3 * half; }     // a far flake behind everything
            if (a1 < 0) a1 += SG_TWO_PI; if (a2 < 0) a2 += SG_TWO_PI;
            if (a1 > SG_TWO_PI) a1 -= SG_TWO_PI; if (a2 > SG_TWO_PI) a2 -= SG_TWO_PI;
            g_a1[j] = a1; g_a2[j] = a2;
            g_rho[j] = d * (j + U(rng)) / (L + 1);                // near -> far
        }
        for (int j = 0; j < L; ++j) { iv[3 * j] = g_a1[j]; iv[3 * j + 1] = g_a2[j]; iv[3 * j + 2] = g_rho[j]; }
        const int n_or = so_occlusion_dict(tr, tl, iv, L, d, div, key, rj, ra);
        const int S = sg_beam_dict<LMAX, 1>(L, tc, d, div, g_a1, g_a2, g_rho, g_ratio, 0, 0, nullptr, nullptr, nullptr);
        bool ok = n_or == S + 1;
        for (int t = 0; ok && t <= S; ++t) ok = memcmp(&rj[t], &g_rho[t], 8) == 0 && memcmp(&ra[t], &g_ratio[t], 8) == 0;
        slots += 2 * L + 1;
        if (!ok) {
            if (bad < 10) printf("MISMATCH LMAX=%d it=%ld L=%d: entries %d/%d\n", LMAX, it, L, n_or, S + 1);
            ++bad;
        }
    }
    printf("dict<%d>: %ld cases, %ld mismatches\n", LMAX, n, bad);
    return bad;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 200000;
    long bad = 0;
    bad += run_dict<4>(n, 1);
    bad += run_dict<8>(n, 2);
    bad += run_dict<16>(n, 3);
    bad += run_dict<63>(n / 4, 4);
    return bad != 0;
}
