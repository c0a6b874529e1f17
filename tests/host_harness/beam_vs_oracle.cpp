// Host harness: the per-beam chain of the kernels -- beam geometry and candidate scan over a table filed by the product's own host
// filing (sg_table_host.h), occlusion dict, amplitudes, received power with its exact pruning, first maximum, attenuate-or-scatter
// decision and the moved coordinates (sg_beam.h: sg_beam, sg_lane_power, sg_beam_decide, sg_scatter_scale) -- compiled for the host
// and run beam by beam against the oracle's process_single_channel (oracle/snow_oracle.c: so_process_channel_f32, pinned to the
// reference's golden vectors) on random float32 and float64 sweeps: the output rows must be the same bytes.  Both arithmetic modes: the kernels'
// own sine / tangent polynomials (default) and libm + the tabulated range grid (snowgpu_set_exact_math).
// usage: beam_vs_oracle [beams]; exit status 1 on any mismatch.  Built and run by tests/test_kernel_math.py.
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
#include "sg_table_host.h"




typedef struct { int32_t channel, min_intensity, max_intensity; double focal_slope, focal_offset; } so_laser;
extern "C" int so_process_channel_f32(const float *pts_in, int64_t M, const double *table_xyr, int64_t K, double beam_div_deg, const so_laser *las,
                                      const double *R, float *pts_out, double *diff_sum, int64_t *dump_count, int64_t *dump_key, double *dump_rj,
                                      double *dump_ratio, int64_t dump_cap, int64_t *dump_used);
extern "C" int so_process_channel_f64(const double *pts_in, int64_t M, const double *table_xyr, int64_t K, double beam_div_deg, const so_laser *las,
                                      const double *R, double *pts_out, double *diff_sum, int64_t *dump_count, int64_t *dump_key, double *dump_rj,
                                      double *dump_ratio, int64_t dump_cap, int64_t *dump_used);
static int oracle_channel(const float *in, int64_t M, const double *xyr, int64_t K, double div, const so_laser *l, const double *R, float *out, double *diff)
{ return so_process_channel_f32(in, M, xyr, K, div, l, R, out, diff, nullptr, nullptr, nullptr, nullptr, 0, nullptr); }
static int oracle_channel(const double *in, int64_t M, const double *xyr, int64_t K, double div, const so_laser *l, const double *R, double *out, double *diff)
{ return so_process_channel_f64(in, M, xyr, K, div, l, R, out, diff, nullptr, nullptr, nullptr, nullptr, 0, nullptr); }
static float row_norm(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }
static double row_norm(double x, double y, double z) { return sqrt((x * x + y * y) + z * z); }

template <typename T, bool EXACT>
static long run_beams(long M, unsigned long long seed, double flake_r, int K)
{
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double div = 0.1718873385392;
    // a table: K disks, uniform in area over 80 m, clear of the origin
    std::vector<double> xyr((size_t)K * 3);
    for (int i = 0; i < K; ++i) {
        const double rho = 1.0 + 79.0 * std::sqrt(U(rng)), phi = U(rng) * SG_TWO_PI;
        xyr[3 * i] = rho * std::cos(phi); xyr[3 * i + 1] = rho * std::sin(phi); xyr[3 * i + 2] = flake_r * (0.3 + 1.4 * U(rng));
    }
    std::vector<SgEntry> entries;
    std::vector<uint32_t> start;
    uint32_t max_bin = 0;
    int64_t bad = -1;
    if (sg_file_table_host(xyr.data(), K, entries, start, max_bin, &bad)) { printf("table filing failed at row %lld\n", (long long)bad); return 1; }
    SgTable tab{};
    tab.entries = entries.data(); tab.bin_start = start.data(); tab.bin_q = nullptr;
    tab.n_bins = SG_NBINS; tab.n_entries = (uint32_t)start[SG_NBINS]; tab.inv_bin_w = SG_NBINS / SG_TWO_PI; tab.n_flakes = (uint32_t)K; tab.max_bin = max_bin;
    const int ch = 17;
    SgLasers las{};
    las.n = 64;
    for (int i = 0; i < 64; ++i) { las.max_i[i] = 255; las.min_i[i] = 0; las.focal_slope[i] = 0.0; las.focal_offset[i] = 0.9; }
    las.max_i[ch] = 255; las.min_i[ch] = 2; las.focal_slope[ch] = 0.013; las.focal_offset[ch] = (1 - 900.0 / 13100) * (1 - 900.0 / 13100);
    so_laser ol{ch, las.min_i[ch], las.max_i[ch], las.focal_slope[ch], las.focal_offset[ch]};
    std::vector<double> R(SG_RBINS);
    for (int k = 0; k < SG_RBINS; ++k) R[k] = sg_range_bin(k);
    std::vector<T> in((size_t)M * 5), ref((size_t)M * 5), got((size_t)M * 5);
    for (long j = 0; j < M; ++j) {
        const double d = 3.0 + 72.0 * U(rng), az = U(rng) * SG_TWO_PI, el = (U(rng) - 0.7) * 0.4;
        in[5 * j] = (T)(d * std::cos(el) * std::cos(az)); in[5 * j + 1] = (T)(d * std::cos(el) * std::sin(az)); in[5 * j + 2] = (T)(d * std::sin(el));
        in[5 * j + 3] = (T)(int)(U(rng) * 255); in[5 * j + 4] = (T)ch;
    }
    double diff = 0;
    const int orc = oracle_channel(in.data(), M, xyr.data(), K, div, &ol, R.data(), ref.data(), &diff);
    if (orc) { printf("oracle returned %d\n", orc); return 1; }
    constexpr int LC = 63;
    static double s_a1[LC + 2], s_a2[LC + 2], s_rho[LC + 2], s_ratio[LC + 2];
    long badn = 0, labels[3] = {0, 0, 0}, over = 0;
    long long diff2 = 0;
    for (long j = 0; j < M; ++j) {
        const T px = in[5 * j], py = in[5 * j + 1], pz = in[5 * j + 2];
        T *o = &got[5 * j];
        memcpy(o, &in[5 * j], 5 * sizeof(T));
        SgBeamOut bo{};
        sg_beam<T, LC, 1>(px, py, pz, ch, tab, &las, div, s_a1, s_a2, s_rho, s_ratio, 0, bo, 0, nullptr, nullptr, nullptr, EXACT);
        if (bo.overflow) { ++over; memcpy(o, &ref[5 * j], 5 * sizeof(T)); continue; }      // (more than 63 flakes: the global-list tier's business)
        int label = 0;
        if (bo.has_power) {
            double best = 0.0;
            int k_best = 0;
            sg_lane_power<1, EXACT, 4, LC>(bo.n_flakes, R.data(), s_a1, s_a2, s_rho, s_ratio, 0, best, k_best);
            const T d_t = row_norm(px, py, pz);
            sg_beam_decide((double)d_t, ch, &las, best, k_best, bo);
            label = bo.label;
            if (label == 2) {                             // simulation.py:178-180: a column of the row dtype times a float64 scalar
                const double sc = sg_scatter_scale(bo.k_best, (double)d_t);
                o[0] = (T)((double)px * sc); o[1] = (T)((double)py * sc); o[2] = (T)((double)pz * sc);
            }
            o[3] = (T)bo.new_i;
            diff2 += (long long)bo.diff2;
        }
        o[4] = (T)label;
        ++labels[label];
        if (memcmp(o, &ref[5 * j], 5 * sizeof(T))) {
            if (badn < 10) printf("MISMATCH beam %ld: got (%.17g %.17g %.17g %g %g) oracle (%.17g %.17g %.17g %g %g)\n", j, (double)o[0], (double)o[1], (double)o[2],
                                  (double)o[3], (double)o[4], (double)ref[5 * j], (double)ref[5 * j + 1], (double)ref[5 * j + 2], (double)ref[5 * j + 3], (double)ref[5 * j + 4]);
            ++badn;
        }
    }
    if ((double)diff2 != 2.0 * diff) { printf("intensity-difference sum: %lld / 2 vs oracle %.17g\n", diff2, diff); ++badn; }
    printf("beams<%s, %s> r=%.3f K=%d: %ld beams, %ld mismatches; labels %ld %ld %ld, %ld beyond 63 flakes\n", sizeof(T) == 4 ? "float32" : "float64",
           EXACT ? "exact" : "default", flake_r, K, M, badn, labels[0], labels[1], labels[2], over);
    return badn;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 100000;
    long bad = 0;
    bad += run_beams<float, false>(n, 7, 0.004, 18000);       // the density of the 2.5 mm/h tables
    bad += run_beams<float, false>(n, 8, 0.02, 18000);        // many flakes per beam
    bad += run_beams<float, true>(n / 2, 9, 0.01, 18000);
    bad += run_beams<double, false>(n, 10, 0.01, 18000);      // float64 rows: the correctly rounded atan2 of the beam azimuth (sg_atan_cr.h)
    bad += run_beams<double, true>(n / 2, 11, 0.01, 18000);
    return bad != 0;
}
