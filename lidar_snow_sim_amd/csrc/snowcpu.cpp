// snowcpu.cpp -- libsnowcpu.so, the CPU twin of the augment path (include/snowgpu_cpu.h; SURVEY 8 b / 8 d: "the build's C++ CPU restatement").
//
// The per-beam arithmetic is NOT restated here: this file includes the kernels' own device code -- sg_beam.h (geometry, candidate scan,
// compute_occlusion_dict, amplitudes, pruned received power, decision), sg_table_host.h (table filing, the code snowgpu_upload_table runs),
// sg_row.h (output rows) -- and compiles it for the host (hipcc --cuda-host-only; the handful of device intrinsics it uses are overloaded
// below, as in tests/host_harness/).  What is written here is the driver the kernels' launch sequence is on the GPU: stable channel sort
// (simulation.py:447), one beam after the other through the global-list form of the per-beam chain (lists of run-time capacity: what
// k_beams_huge runs), np.round / noise-floor filter / statistics (simulation.py:516-530), spread over host threads by (frame, 256 sorted rows).
//
// A measurement baseline and a parity check (bench.py: cpu_twin; tests/test_gpu_parity.py) -- never a fallback: nothing in the package loads it.
// Build: lidar_snow_sim_amd/build.py (hipcc --cuda-host-only -x hip -O2 -ffp-contract=off -fPIC -shared).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <sched.h>
// host overloads of the device intrinsics the included code uses (a "wave" of one lane)
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
#include "sg_row.h"
#include "sg_table_host.h"
#include "../../include/snowgpu_cpu.h"

namespace {

struct CpuTable {
    std::vector<SgEntry> entries;
    std::vector<uint32_t> start;
    SgTable tab{};
};

int usable_cpus()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) return n; }
    const unsigned hc = std::thread::hardware_concurrency();
    return hc ? (int)hc : 1;
}

template <typename F>
void parallel_for(int64_t n_items, int threads, F &&body)
{
    std::atomic<int64_t> next{0};
    auto work = [&](int t) { for (int64_t i; (i = next.fetch_add(1)) < n_items;) body(i, t); };
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n_items));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
}

template <typename T>
int augment_batch(int n_frames, const int64_t *off, const T *rows, int n_tables, const double *const *xyr, const int64_t *tk, const int32_t *table_ids,
                  const SgLasers &las, double div, const double *thr_poly, int threads, T *out_rows, int32_t *out_src, int64_t *out_counts,
                  int64_t *out_stats, int32_t *status)
{
    const int n_las = las.n;
    std::atomic<int> err{0}, err_row{-1};
    auto fail = [&](int code, int64_t where) { int z = 0; if (err.compare_exchange_strong(z, code)) err_row.store((int)where); };
    // ---- tables: filed once, as snowgpu_upload_table files them (sg_table_host.h)
    std::vector<CpuTable> tabs((size_t)n_tables);
    uint32_t max_flakes = 0;
    parallel_for(n_tables, threads, [&](int64_t t, int) {
        CpuTable &c = tabs[(size_t)t];
        uint32_t max_bin = 0;
        int64_t bad = -1;
        if (sg_file_table_host(xyr[t], tk[t], c.entries, c.start, max_bin, &bad)) { fail(3 /* SNOWGPU_E_TABLE */, t); return; }
        c.tab.entries = c.entries.data(); c.tab.bin_start = c.start.data(); c.tab.bin_q = nullptr;
        c.tab.n_bins = SG_NBINS; c.tab.n_entries = c.start[SG_NBINS]; c.tab.inv_bin_w = SG_NBINS / SG_TWO_PI;
        c.tab.n_flakes = (uint32_t)tk[t]; c.tab.max_bin = max_bin;
    });
    if (err.load()) { status[0] = err.load(); status[1] = err_row.load(); return status[0]; }
    for (int t = 0; t < n_tables; ++t) max_flakes = std::max<uint32_t>(max_flakes, (uint32_t)tk[t]);
    const int cap = (int)std::min<uint32_t>(std::max<uint32_t>(max_flakes, 64u), 8192u);      // the global-list tier's capacity (snowgpu_api.cpp: h_cap)
    std::vector<double> R(SG_RBINS);
    for (int k = 0; k < SG_RBINS; ++k) R[k] = sg_range_bin(k);
    const int64_t n_total = off[n_frames];
    // ---- stable channel sort per frame (simulation.py:447; the reference's argsort is unstable: DESIGN.md "canonical order")
    std::vector<int32_t> perm((size_t)n_total);
    parallel_for(n_frames, threads, [&](int64_t f, int) {
        const int64_t b = off[f], n = off[f + 1] - b;
        int32_t *p = perm.data() + b;
        for (int64_t i = 0; i < n; ++i) p[i] = (int32_t)i;
        const T *fr = rows + b * 5;
        std::stable_sort(p, p + n, [fr](int32_t x, int32_t y) { return fr[(size_t)x * 5 + 4] < fr[(size_t)y * 5 + 4]; });
    });
    // ---- per beam: items of 256 sorted positions of one frame
    std::vector<uint32_t> rec((size_t)n_total);
    std::vector<std::atomic<long long>> diff2((size_t)n_frames);
    for (auto &d : diff2) d.store(0);
    std::vector<int64_t> item_f, item_g;
    for (int f = 0; f < n_frames; ++f)
        for (int64_t g = off[f]; g < off[f + 1]; g += 256) { item_f.push_back(f); item_g.push_back(g); }
    std::vector<std::vector<double>> scratch((size_t)std::max(threads, 1));
    parallel_for((int64_t)item_f.size(), threads, [&](int64_t it, int t) {
        std::vector<double> &sc = scratch[(size_t)t];
        if (sc.empty()) sc.resize((size_t)4 * (size_t)(cap + 2));
        double *s_a1 = sc.data(), *s_a2 = s_a1 + cap + 2, *s_rho = s_a2 + cap + 2, *s_ratio = s_rho + cap + 2;
        const int f = (int)item_f[(size_t)it];
        const int64_t b = off[f], g1 = std::min<int64_t>(item_g[(size_t)it] + 256, off[f + 1]);
        long long d2 = 0;
        for (int64_t g = item_g[(size_t)it]; g < g1; ++g) {
            const T *row = rows + (b + perm[(size_t)g]) * 5;
            const T px = row[0], py = row[1], pz = row[2], pch = row[4];
            const int ch = (int)pch;
            if (!(((T)ch == pch) && ch >= 0 && ch < n_las)) { rec[(size_t)g] = SG_REC_COPY; continue; }     // simulation.py:80, :482 (Q5)
            const int32_t tid_tab = table_ids[(int64_t)f * n_las + ch];
            if (tid_tab < 0 || tid_tab >= n_tables) { fail(1 /* SNOWGPU_E_INVALID */, g); rec[(size_t)g] = 0; continue; }
            SgBeamOut o;
            o.overflow = 0; o.range_error = 0; o.diff2 = 0.0; o.has_power = 0; o.n_flakes = 0; o.n_hits = 0; o.label = 0; o.new_i = 0; o.k_best = 0;
            sg_beam<T, 0, 0>(px, py, pz, ch, tabs[(size_t)tid_tab].tab, &las, div, s_a1, s_a2, s_rho, s_ratio, 0, o, 0, nullptr, nullptr, nullptr, false, 1, cap);
            uint32_t rc = 0;
            if (o.overflow) fail(6 /* SNOWGPU_E_OVERFLOW */, g);
            else if (o.range_error) fail(4 /* SNOWGPU_E_RANGE */, g);
            if (o.has_power) {
                double best = 0.0;
                int k_best = 0;
                sg_lane_power<0, false, 8, 0>(o.n_flakes, R.data(), s_a1, s_a2, s_rho, s_ratio, 0, best, k_best, 1, cap);
                T d_t;
                if constexpr (sizeof(T) == 4) d_t = sqrtf((px * px + py * py) + pz * pz);
                else d_t = sqrt((px * px + py * py) + pz * pz);
                sg_beam_decide((double)d_t, ch, &las, best, k_best, o);
                rc = sg_pack_record(o);
                d2 += (long long)o.diff2;
            }
            rec[(size_t)g] = rc;
        }
        if (d2) diff2[(size_t)f].fetch_add(d2);
    });
    if (err.load()) { status[0] = err.load(); status[1] = err_row.load(); return status[0]; }
    // ---- output rows, np.round, noise-floor filter, statistics (simulation.py:516-530; k_compact_* on the GPU)
    parallel_for(n_frames, threads, [&](int64_t f, int) {
        const int64_t b = off[f], n = off[f + 1] - b;
        const double p0 = thr_poly[f * 3], p1 = thr_poly[f * 3 + 1], p2 = thr_poly[f * 3 + 2];
        int64_t kept = 0, att = 0;
        for (int64_t g = b; g < b + n; ++g) {
            const SgRow<T> o = sg_rebuild_row<T>(rows + (b + perm[(size_t)g]) * 5, rec[(size_t)g]);
            const T dd2 = o.dd * o.dd;
            const double thr = (p0 * (double)dd2 + p1 * (double)o.dd) + p2;        // :465-469: d^2 in the row dtype
            const bool keep = (o.lab == (T)2) || ((double)o.i > thr);             // :518-520
            if (!keep) continue;
            att += o.lab == (T)1;
            T *q = out_rows + (b + kept) * 5;
            q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = o.i; q[4] = o.lab;
            if (out_src) out_src[b + kept] = perm[(size_t)g];
            ++kept;
        }
        out_counts[f] = kept;
        out_stats[f * 3 + 0] = att;                                               // :525
        out_stats[f * 3 + 1] = n - kept;                                          // :522
        const double diff_sum = (double)diff2[(size_t)f].load() / 2.0;
        out_stats[f * 3 + 2] = att > 0 ? (int64_t)(diff_sum / (double)att) : 0;   // :527-530 int()
    });
    status[0] = 0; status[1] = -1;
    return 0;
}

}  // namespace

extern "C" const char *snowgpu_cpu_version(void) { return "snowcpu 0.1.0 host (the kernels' device code compiled for the host; baseline, not a fallback)"; }

extern "C" int snowgpu_cpu_augment_batch(int n_frames, const int64_t *frame_offsets, const void *rows, int dtype, int n_tables,
                                         const double *const *tables_xyr, const int64_t *tables_k, const int32_t *table_ids, int n_lasers,
                                         const double *focal_slope, const double *focal_offset, const int32_t *min_intensity,
                                         const int32_t *max_intensity, double beam_divergence_deg, const double *thr_poly, double noise_floor,
                                         int threads, void *out_rows, int32_t *out_src, int64_t *out_counts, int64_t *out_stats, int32_t *status)
{
    int32_t local[2] = {0, -1};
    if (!status) status = local;
    status[0] = 1; status[1] = -1;
    if (n_frames <= 0 || !frame_offsets || !rows || (dtype != 0 && dtype != 1) || n_tables <= 0 || !tables_xyr || !tables_k || !table_ids ||
        n_lasers <= 0 || n_lasers > SG_MAX_LASERS || !focal_slope || !focal_offset || !min_intensity || !max_intensity || !thr_poly ||
        !out_rows || !out_counts || !out_stats || !(beam_divergence_deg > 0 && beam_divergence_deg < 45.0))
        return 1;                                               /* SNOWGPU_E_INVALID */
    (void)noise_floor;                                          // (part of thr_poly already: simulation.py:462 multiplies the fitted line by it)
    SgLasers las{};
    las.n = n_lasers;
    for (int c = 0; c < n_lasers; ++c) { las.focal_slope[c] = focal_slope[c]; las.focal_offset[c] = focal_offset[c]; las.min_i[c] = min_intensity[c]; las.max_i[c] = max_intensity[c]; }
    if (threads <= 0) threads = usable_cpus();
    if (dtype == 0)
        return augment_batch<float>(n_frames, frame_offsets, (const float *)rows, n_tables, tables_xyr, tables_k, table_ids, las, beam_divergence_deg,
                                    thr_poly, threads, (float *)out_rows, out_src, out_counts, out_stats, status);
    return augment_batch<double>(n_frames, frame_offsets, (const double *)rows, n_tables, tables_xyr, tables_k, table_ids, las, beam_divergence_deg,
                                 thr_poly, threads, (double *)out_rows, out_src, out_counts, out_stats, status);
}
