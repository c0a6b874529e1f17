// Host probe (no GPU): how many (beam, record) pairs the candidate scan tests per beam as a function of the number of azimuth bins a table is
// filed under (csrc/sg_table_host.h, sg_beam.h), on a bench table and the rows of a bench sweep.
//   python: bench.make_tables(...)[0] -> table.bin (K x 3 float64), bench.make_frame(...)[:, :3] -> rows.bin (N x 3 float32)
//   hipcc --cuda-host-only -x hip -O2 -std=c++17 -DPROBE_NBINS=4096 -I lidar_snow_sim_amd/csrc -I include scripts/probe/bin_width_probe.cpp -o probe && ./probe table.bin rows.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
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
#undef __device__
#define __device__
#include "sg_common.h"
#undef SG_NBINS
#define SG_NBINS PROBE_NBINS
#include "sg_beam.h"
#include "sg_table_host.h"

static std::vector<char> slurp(const char *p) { FILE *f = fopen(p, "rb"); std::vector<char> b; if (!f) return b; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); b.resize((size_t)n); if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) b.clear(); fclose(f); return b; }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::vector<char> tb = slurp(argv[1]), rb = slurp(argv[2]);
    const int64_t K = (int64_t)(tb.size() / 24), N = (int64_t)(rb.size() / 12);
    const double *xyr = (const double *)tb.data();
    const float *rows = (const float *)rb.data();
    std::vector<SgEntry> entries;
    std::vector<uint32_t> start;
    uint32_t max_bin = 0;
    int64_t bad = -1;
    if (sg_file_table_host(xyr, K, entries, start, max_bin, &bad)) { printf("filing failed\n"); return 1; }
    const double div = 0.1718873385392;
    double pairs = 0, hits = 0, bins = 0, beams = 0;
    for (int64_t i = 0; i < N; ++i) {
        const float px = rows[3 * i], py = rows[3 * i + 1], pz = rows[3 * i + 2];
        float d_t;
        const SgBeamGeo g = sg_beam_geometry<float>(px, py, pz, div, false, d_t);
        const int nb = SG_NBINS;
        const double inv = SG_NBINS / SG_TWO_PI;
        const int b_lo = sg_bin_of(g.theta_r - SG_BEAM_MARGIN, inv, nb), b_hi = sg_bin_of(g.theta_l + SG_BEAM_MARGIN, inv, nb);
        int span = b_hi - b_lo; if (span < 0) span += nb;
        int b = b_lo;
        for (int s = 0; s <= span; ++s) {
            for (uint32_t e = start[b]; e < start[b + 1]; ++e) {
                const SgEntry &f = entries[e];
                if (!(f.rho < g.d)) break;
                pairs += 1;
                if (s > 0 && !(f.flags & 1u)) continue;
                double a1, a2;
                if (sg_flake_hits(g, f, a1, a2)) hits += 1;
            }
            if (++b == nb) b = 0;
        }
        bins += span + 1; beams += 1;
    }
    printf("bins %5d: records %zu (%.2f per flake), per beam: %.2f bins touched, %.2f pairs tested, %.3f flakes met; %.1f %% of the pairs intersect\n", (int)SG_NBINS,
           entries.size() - 1, (double)(entries.size() - 1) / (double)K, bins / beams, pairs / beams, hits / beams, 100.0 * hits / pairs);
    return 0;
}
