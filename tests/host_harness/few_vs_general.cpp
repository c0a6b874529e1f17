// Host harness: the few-flake register path of k_power_few (lidar_snow_sim_amd/csrc/sg_few.h) against the general per-lane path
// (sg_beam.h: sg_beam_dict + sg_beam_amp + sg_lane_power) on random beams, bit for bit.  Both are the kernels' own device functions,
// compiled for the host (hipcc --cuda-host-only; the handful of device intrinsics they use get host overloads below) -- no oracle,
// no GPU.  usage: few_vs_general [cases per N]; exit status 1 on any mismatch.  Built and run by tests/test_kernel_math.py.
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


template <int N>
static long run_few(long n, unsigned long long seed)
{
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    SgLasers las{};
    las.n = 64;
    for (int i = 0; i < 64; ++i) { las.max_i[i] = (i % 7 == 0) ? 230 : 255; las.min_i[i] = 0; las.focal_slope[i] = 0.001 * i; las.focal_offset[i] = 0.9; }
    const double div = 0.1718873385392;
    long bad = 0, s_hist[5] = {0, 0, 0, 0, 0}, bins = 0;
    for (long it = 0; it < n; ++it) {
        const float df = (float)(3.0 + 100.0 * U(rng) * U(rng));
        const double d = (double)df;
        double tc = (double)(float)(U(rng) * SG_TWO_PI);
        if (it % 500 == 0) tc = (double)(float)(U(rng) * 0.002);
        if (it % 500 == 1) tc = (double)(float)(SG_TWO_PI - U(rng) * 0.002);
        double tr, tl; sg_beam_limits(tc, div, tr, tl);
        const double half = div / 2 * (SG_PI / 180.0);
        const int L = 1 + (int)(U(rng) * N) % N;
        double a1[N], a2[N], rho[N];
        const double cluster = d * U(rng);
        for (int j = 0; j < N; ++j) {
            double c = tc + (U(rng) * 2.4 - 1.2) * half, w = half * (0.01 + 0.7 * U(rng) * U(rng));
            a1[j] = c - w; a2[j] = c + w;
            if (U(rng) < 0.3) a1[j] = tr;
            if (U(rng) < 0.3) a2[j] = tl;
            if (U(rng) < 0.1 && j > 0) a1[j] = a2[j - 1];       // shared endpoints
            if (U(rng) < 0.05 && j > 0) { a1[j] = a1[j - 1]; a2[j] = a2[j - 1]; }   // identical intervals
            if (a1[j] < 0) a1[j] += SG_TWO_PI; if (a2[j] < 0) a2[j] += SG_TWO_PI;
            if (a1[j] > SG_TWO_PI) a1[j] -= SG_TWO_PI; if (a2[j] > SG_TWO_PI) a2[j] -= SG_TWO_PI;
            const double u = U(rng);
            rho[j] = u < 0.4 ? d - 4.0 * U(rng) : (u < 0.7 ? cluster + 3.0 * U(rng) : d * U(rng));
            if (U(rng) < 0.05) rho[j] = d * std::sqrt(w / half) * (0.98 + 0.04 * U(rng));
            if (!(rho[j] > 0.05)) rho[j] = 0.05 + U(rng);
            if (!(rho[j] < d)) rho[j] = d * (0.999 - 0.0001 * j);
            if (U(rng) < 0.02 && j > 0) rho[j] = rho[j - 1];    // equal ranges
        }
        for (int i = 1; i < L; ++i)                              // near -> far, stable (the scan's order)
            for (int q = i; q > 0 && rho[q - 1] > rho[q]; --q) { std::swap(rho[q - 1], rho[q]); std::swap(a1[q - 1], a1[q]); std::swap(a2[q - 1], a2[q]); }
        const int ch = (int)(U(rng) * 64) & 63;
        // ---- general path, one lane
        double g_a1[8], g_a2[8], g_rho[8], g_ratio[8];
        for (int j = 0; j < L; ++j) { g_a1[j] = a1[j]; g_a2[j] = a2[j]; g_rho[j] = rho[j]; }
        SgBeamOut o1{};
        const int S1 = sg_beam_dict<4, 1>(L, tc, d, div, g_a1, g_a2, g_rho, g_ratio, 0, 0, nullptr, nullptr, nullptr);
        double best1 = 0.0; int k1 = 0;
        if (S1 > 0) {
            sg_beam_amp<float, 4, 1>(df, S1, ch, &las, g_a1, g_a2, g_rho, g_ratio, 0, o1, 0);
            sg_lane_power<1, false, 4, 4>(S1, nullptr, g_a1, g_a2, g_rho, g_ratio, 0, best1, k1);
        }
        // ---- few-flake path
        SgBeamOut o2{};
        SgFew<N> P{};
        const int S2 = sg_few_prep<float, N>(d, tc, L, a1, a2, rho, ch, &las, div, P, o2);
        double best2 = 0.0; int k2 = 0;
        if (S2) {
            int ka[N + 1], kb[N + 1];
            sg_few_zone<N, 0>(P, 0.0, ka[0], kb[0]);
            if constexpr (N >= 2) sg_few_zone<N, 1>(P, 0.0, ka[1], kb[1]);
            if constexpr (N >= 3) sg_few_zone<N, 2>(P, 0.0, ka[2], kb[2]);
            sg_few_zone<N, N>(P, 0.0, ka[N], kb[N]);
            for (int z = 0; z <= N; ++z)
                for (int k = ka[z]; k <= kb[z]; ++k) {
                    const double sm = sg_few_bin<false, N>(P.amp, P.rho, P.k0, P.k1, P.tamp, P.d, P.tk0, P.tk1, k, nullptr);
                    if (sm > best2 || (sm == best2 && k < k2)) { best2 = sm; k2 = k; }
                    ++bins;
                }
        }
        ++s_hist[S2];
        if (S1 != S2 || memcmp(&best1, &best2, 8) || k1 != k2 || o1.range_error != o2.range_error) {
            if (bad < 10) printf("MISMATCH N=%d it=%ld L=%d d=%.9g tc=%.9g: S %d/%d best %.17g/%.17g k %d/%d\n", N, it, L, d, tc, S1, S2, best1, best2, k1, k2);
            ++bad;
        }
    }
    printf("few<%d>: %ld cases, %ld mismatches; S histogram %ld %ld %ld %ld; %.2f bins per beam\n", N, n, bad, s_hist[0], s_hist[1], s_hist[2], s_hist[3], (double)bins / n);
    return bad;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 1000000;
    long bad = 0;
    bad += run_few<1>(n, 11);
    bad += run_few<2>(n, 22);
    bad += run_few<3>(n, 33);
    return bad != 0;
}
