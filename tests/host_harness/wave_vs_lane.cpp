// Host harness: the candidate scan of the pass over all rows (sg_beam.h: sg_wave_scan with the one-word lists, COMPACT, with and without
// deferred distance tests, with and without the tables' coarse range index) against the one-beam-per-lane scan (sg_beam_scan), which
// tests/host_harness/beam_vs_oracle.cpp holds to the oracle.  A "wave" of one lane (SG_PAIR_WINDOW = 1): the cross-lane reads return the lane's own values, so
// what runs is the scan's own logic -- counts per bin by the coarse index and the search, pair numbering, the records' first half for
// the decision and the second for the list, the word per flake and its resolution (sg_hit_word / sg_hit_angles), the sort by (range, scan
// order), the count beyond a full list.  Per beam: the same number of intersecting flakes and the same list (interval angles and ranges,
// bit for bit) -- unless a deferred test was undecided, which the scan must then report.
// usage: wave_vs_lane [beams]; exit status 1 on any mismatch.  Built and run by tests/test_kernel_math.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <cmath>
#include <vector>
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
#define SG_PAIR_WINDOW 1     /* a wave of one lane takes one pair per trip */
#include "sg_beam.h"
#include "sg_table_host.h"

template <typename T, bool DEFER>
static long run(long M, unsigned long long seed, double flake_r, int K, double div, bool coarse_index)
{
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
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
    std::vector<uint32_t> q;
    if (coarse_index) {                                   // snowgpu_tables.hip: k_table_index
        q.resize((size_t)SG_NBINS * SG_QSTEPS);
        for (int b = 0; b < SG_NBINS; ++b)
            for (int k = 0; k < SG_QSTEPS; ++k) {
                uint32_t lo = start[b], hi = start[b + 1];
                const double lim = SG_QSTEP_M * (double)k;
                while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (entries[m].rho < lim) lo = m + 1; else hi = m; }
                q[(size_t)b * SG_QSTEPS + k] = lo - start[b];
            }
    }
    SgTable tab{};
    tab.entries = entries.data(); tab.bin_start = start.data(); tab.bin_q = coarse_index ? q.data() : nullptr;
    tab.n_bins = SG_NBINS; tab.n_entries = (uint32_t)start[SG_NBINS]; tab.inv_bin_w = SG_NBINS / SG_TWO_PI; tab.n_flakes = (uint32_t)K; tab.max_bin = max_bin;
    constexpr int LMAX = 4;
    long badn = 0, undecided = 0, full = 0, with_flakes = 0;
    for (long j = 0; j < M; ++j) {
        const double d = 3.0 + 72.0 * U(rng), az = U(rng) * SG_TWO_PI, el = (U(rng) - 0.7) * 0.4;
        const T px = (T)(d * std::cos(el) * std::cos(az)), py = (T)(d * std::cos(el) * std::sin(az)), pz = (T)(d * std::sin(el));
        // one beam per lane: three value columns
        double l_a1[LMAX], l_a2[LMAX], l_rho[LMAX];
        SgBeamOut lo{};
        T d_l; double th_l;
        const int L0 = sg_beam_scan<T, LMAX, 1>(px, py, pz, tab, div, l_a1, l_a2, l_rho, 0, lo, d_l, th_l, false);
        // the wave scan, one-word lists: ranges, words, keys, counter, bin starts; overflow slot of its own
        double w_rho[LMAX];
        uint32_t w_rec[LMAX + 1];                          // (read through a double pointer: keep it 8-byte aligned and a word longer)
        alignas(8) uint32_t rec_store[LMAX + 2];
        (void)w_rec;
        int w_cnt[64], w_key[LMAX], w_st[2];
        std::vector<double> ov((size_t)SG_OV_STRIDE, 0.0);
        SgBeamOut wo{};
        T d_w; double th_w;
        const int L1 = sg_wave_scan<T, LMAX, 1, DEFER, true>(true, px, py, pz, tab, div, reinterpret_cast<double *>(rec_store), nullptr, w_rho, w_cnt, w_key, w_st, 0,
                                                             wo, d_w, th_w, false, ov.data(), SG_OV_CAP);
        if (DEFER && (wo.n_hits & SG_HITS_UNDECIDED)) { ++undecided; if (!wo.overflow) { ++badn; printf("beam %ld: undecided without the overflow flag\n", j); } continue; }
        bool ok = L0 == L1 && lo.n_hits == wo.n_hits && lo.overflow == wo.overflow && memcmp(&d_l, &d_w, sizeof(T)) == 0 && memcmp(&th_l, &th_w, 8) == 0;
        double th_r, th_lft;
        sg_beam_limits(th_w, div, th_r, th_lft);
        for (int i = 0; ok && i < L1; ++i) {
            double a1, a2;
            sg_hit_angles(rec_store[i], tab.entries, th_r, th_lft, a1, a2);
            ok = memcmp(&a1, &l_a1[i], 8) == 0 && memcmp(&a2, &l_a2[i], 8) == 0 && memcmp(&w_rho[i], &l_rho[i], 8) == 0;
        }
        // flakes beyond the list, up to the slot's capacity, lie in the overflow slot as the scan met them: the same SET as the per-lane scan's
        // count says (their order is the consumer's business: it sorts)
        if (ok && wo.overflow && wo.n_hits <= SG_OV_CAP) {
            for (int h = LMAX; ok && h < wo.n_hits; ++h) ok = ov[2 + 3 * h + 2] > 0.0;
        }
        if (L1 > 0) ++with_flakes;
        if (wo.overflow) ++full;
        if (!ok) {
            if (badn < 10) printf("MISMATCH beam %ld: lists %d / %d, flakes met %d / %d\n", j, L0, L1, lo.n_hits, wo.n_hits);
            ++badn;
        }
    }
    printf("wave<%s, %s, %s> r=%.3f div=%.3f: %ld beams, %ld mismatches; %ld with flakes, %ld beyond the list, %ld undecided\n", sizeof(T) == 4 ? "float32" : "float64",
           DEFER ? "deferred" : "in place", coarse_index ? "coarse index" : "plain search", flake_r, div, M, badn, with_flakes, full, undecided);
    return badn;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 50000;
    const double bd = 0.1718873385392;
    long bad = 0;
    bad += run<float, true>(n, 21, 0.004, 18000, bd, true);        // the pass over all rows as it runs: deferred tests, coarse index
    bad += run<float, false>(n, 22, 0.02, 18000, bd, false);       // many flakes per beam, lists overflow; the exact-math instantiation's in-place tests
    bad += run<double, true>(n, 23, 0.01, 18000, bd, true);
    bad += run<float, true>(n / 4, 24, 0.01, 18000, 1.7, true);    // 30 mrad beams: wedges over many bins (the per-lane loop beyond the second bin)
    bad += run<double, false>(n / 4, 25, 0.01, 40000, 0.02, false);
    return bad != 0;
}
