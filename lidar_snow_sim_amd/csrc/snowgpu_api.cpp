// snowgpu_api.cpp -- the C ABI of libsnowgpu.so (include/snowgpu.h): context, table filing, scratch
// management and the launch sequence of one augment batch.  Host-side C++; every kernel lives in
// snowgpu_kernels.hip / snowgpu_prepass.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/snowgpu.h"
#include "sg_common.h"
#include "sg_prepass.h"

namespace {

struct DeviceTable {
    SgEntry *entries = nullptr;
    uint32_t *bin_start = nullptr;
    SgTable desc{};
};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    int ensure(size_t n)
    {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) return (int)e;
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct snowgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux2 = nullptr;           // side stream of the noise-threshold prepass (only the compaction needs its result)
    hipEvent_t ev_fork0 = nullptr, ev_join0 = nullptr;
    hipStream_t aux = nullptr;            // side stream: launch-order bookkeeping that only needs the sort, next to the prepass
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    std::vector<DeviceTable> tables;
    SgTable *d_tables = nullptr;      // device mirror of the descriptors
    size_t d_tables_cap = 0;
    bool tables_dirty = true;
    uint32_t max_flakes = 0;          // largest uploaded table (drives the LMAX choice)
    SgLasers h_las{};
    SgLasers *d_las = nullptr;
    double *d_rgrid = nullptr;
    int32_t *d_status = nullptr;      // 4 ints
    // scratch shared by every batch
    DevBuf<int32_t> tile_hist, tile_base, ovf_list, ovf_list2, perm, ctile_cnt, ctile_base, table_ids, out_src;
    DevBuf<unsigned long long> seg_tbl_cnt, seg_tbl_base;
    DevBuf<int32_t> seg_blk, seg_cnt, seg_frame, seg_n, seg_of_blk, pq_list, ptile_cnt, ptile_base;
    DevBuf<double> pq_dict;
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;
    DevBuf<int64_t> seg_start;
    bool linear_order = false;   // experiments: SNOWGPU_LINEAR_ORDER=1 keeps the first pass in sorted-row order
    DevBuf<uint16_t> rank;
    DevBuf<uint8_t> keep, rows_in, rows_tmp, rows_out;
    DevBuf<int64_t> frame_off, out_counts, out_stats;
    DevBuf<double> thr_poly, plane, dbg_rj, dbg_ratio, user_thr, out_thr;
    DevBuf<int32_t> user_perm;
    DevBuf<int32_t> dbg_count;
    DevBuf<unsigned long long> diff2;
    DevBuf<SgTable> frame_tables;
    SgPrepassScratch prepass{};
    // measurement hooks (snowgpu_profile_begin / _end)
    std::vector<hipEvent_t> ev_start, ev_stop;
    int ev_used = 0;
    bool prof = false;
    hipStream_t prof_stream = nullptr;
    int exact_math = 0;
    unsigned long long *phase_cycles = nullptr;   // device [8], experiments only
};

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return SNOWGPU_E_HIP;                                                                 \
        }                                                                                         \
    } while (0)

#define ENSURE(ctx, buf, n)                                                                       \
    do {                                                                                          \
        if ((buf).ensure(n)) { (ctx)->err = "hipMalloc failed for " #buf; return SNOWGPU_E_HIP; } \
    } while (0)

static int fail(snowgpu_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

// simulation.py:106-116: R = np.round(np.linspace(0, 120 + c*tau_h, 1230), 2).
// linspace: k * step (+ 0.0), last element = stop; round(., 2): rint(v * 100) / 100.
static void range_grid(double *out)
{
    const double stop = 120 + 299792458.0 * 1e-8;
    const int num = SG_RBINS;
    const double step = stop / (num - 1);
    for (int k = 0; k < num; ++k) {
        double v = (double)k * step + 0.0;
        if (k == num - 1) v = stop;
        out[k] = std::rint(v * 100.0) / 100.0;
    }
}

extern "C" const char *snowgpu_version(void) { return "snowgpu 0.1.0 gfx950 (MI355X) hip"; }

extern "C" int snowgpu_range_grid(double *out)
{
    if (!out) return SNOWGPU_E_INVALID;
    range_grid(out);
    return SNOWGPU_OK;
}

extern "C" const char *snowgpu_last_error(const snowgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int snowgpu_create(int device, snowgpu_ctx **out)
{
    if (!out) return SNOWGPU_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SNOWGPU_E_NO_DEVICE;
    snowgpu_ctx *ctx = new snowgpu_ctx();
    ctx->device = device;
    { const char *lo = std::getenv("SNOWGPU_LINEAR_ORDER"); ctx->linear_order = lo && lo[0] == '1'; }
    *out = ctx;   // hand the context back even on failure so that last_error is readable
    HIPCHK(ctx, hipSetDevice(device));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux2, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork0, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join0, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork2, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_las, sizeof(SgLasers)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_rgrid, sizeof(double) * SG_RBINS));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_status, sizeof(int32_t) * 8));
    double grid[SG_RBINS];
    range_grid(grid);
    HIPCHK(ctx, hipMemcpy(ctx->d_rgrid, grid, sizeof(grid), hipMemcpyHostToDevice));
    ctx->h_las.n = 0;
    return SNOWGPU_OK;
}

extern "C" void snowgpu_destroy(snowgpu_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->tables) {
        if (t.entries) (void)hipFree(t.entries);
        if (t.bin_start) (void)hipFree(t.bin_start);
    }
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->d_las) (void)hipFree(ctx->d_las);
    if (ctx->d_rgrid) (void)hipFree(ctx->d_rgrid);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    ctx->tile_hist.release(); ctx->tile_base.release(); ctx->ovf_list.release(); ctx->ovf_list2.release(); ctx->perm.release();
    ctx->seg_tbl_cnt.release(); ctx->seg_tbl_base.release(); ctx->seg_blk.release(); ctx->seg_cnt.release(); ctx->seg_frame.release(); ctx->seg_n.release(); ctx->seg_start.release(); ctx->seg_of_blk.release(); ctx->pq_list.release(); ctx->ptile_cnt.release(); ctx->ptile_base.release(); ctx->pq_dict.release();
    ctx->ctile_cnt.release(); ctx->ctile_base.release(); ctx->table_ids.release(); ctx->out_src.release();
    ctx->rank.release(); ctx->keep.release(); ctx->rows_in.release(); ctx->rows_tmp.release(); ctx->rows_out.release();
    ctx->frame_off.release(); ctx->out_counts.release(); ctx->out_stats.release();
    ctx->thr_poly.release(); ctx->plane.release(); ctx->dbg_rj.release(); ctx->dbg_ratio.release();
    ctx->dbg_count.release(); ctx->diff2.release(); ctx->frame_tables.release(); ctx->user_thr.release(); ctx->out_thr.release(); ctx->user_perm.release();
    sg_prepass_release(&ctx->prepass);
    for (auto e : ctx->ev_start) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_stop) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_fork2) (void)hipEventDestroy(ctx->ev_fork2);
    if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
    if (ctx->ev_fork0) (void)hipEventDestroy(ctx->ev_fork0);
    if (ctx->ev_join0) (void)hipEventDestroy(ctx->ev_join0);
    if (ctx->aux2) (void)hipStreamDestroy(ctx->aux2);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int snowgpu_set_lasers(snowgpu_ctx *ctx, int n, const double *focal_slope, const double *focal_offset,
                                  const int32_t *min_intensity, const int32_t *max_intensity)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n <= 0 || n > SG_MAX_LASERS || !focal_slope || !focal_offset || !min_intensity || !max_intensity)
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_lasers: need 1..256 lasers and four arrays");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n; ++i) {
        ctx->h_las.focal_slope[i] = focal_slope[i];
        ctx->h_las.focal_offset[i] = focal_offset[i];
        ctx->h_las.min_i[i] = min_intensity[i];
        ctx->h_las.max_i[i] = max_intensity[i];
    }
    ctx->h_las.n = n;
    HIPCHK(ctx, hipMemcpy(ctx->d_las, &ctx->h_las, sizeof(SgLasers), hipMemcpyHostToDevice));
    return SNOWGPU_OK;
}

// ---- table filing --------------------------------------------------------------------------------
// Per flake (beam-independent, hoisted out of get_occlusions' beam loop): rho, phi, the two tangent
// angles.  Same operation order as the reference (geometry.py:138-190, :32-80; simulation.py:351-352).
static bool forward_of(double ray, double centre)
{
    double d = ray - centre;
    return (std::fabs(d) < SG_PI / 2) || (std::fabs(d - SG_TWO_PI) < SG_PI / 2) || (std::fabs(d + SG_TWO_PI) < SG_PI / 2);
}

static bool derive_flake(double x, double y, double r, SgEntry *f)
{
    if (!(std::isfinite(x) && std::isfinite(y) && std::isfinite(r)) || !(r > 0)) return false;
    f->x = x; f->y = y; f->r = r;
    f->rho = std::sqrt(x * x + y * y);
    if (!(f->rho > r)) return false;                   // disk contains the origin: sqrt of a negative below
    f->phi = std::atan2(y, x);
    if (f->phi < 0) f->phi = f->phi + SG_TWO_PI;
    double a[2], b[2];
    const double disc = r * std::sqrt(x * x + y * y - r * r);
    if (std::fabs(x) - r == 0) {
        a[0] = 1.0; b[0] = 0.0;
        a[1] = (y * y - x * x) / (2 * x * y); b[1] = -1.0;
    } else {
        a[0] = (-x * y + disc) / (r * r - x * x);
        a[1] = (-x * y - disc) / (r * r - x * x);
        b[0] = b[1] = -1.0;
    }
    double ang[2];
    for (int i = 0; i < 2; ++i) {
        double ray1 = std::atan(-a[i] / b[i]);
        double ray2 = ray1 + SG_PI;
        if (ray1 < 0) ray1 = ray1 + SG_TWO_PI;
        ray1 = std::fabs(ray1);
        if (b[i] == 0) { ray1 = SG_PI / 2; ray2 = 3 * SG_PI / 2; }
        const bool ok1 = forward_of(ray1, f->phi), ok2 = forward_of(ray2, f->phi);
        if (ok1 == ok2) return false;                  // the reference would raise / mis-align (geometry.py:72)
        ang[i] = ok1 ? ray1 : ray2;
    }
    const double lo = std::min(ang[0], ang[1]), hi = std::max(ang[0], ang[1]);
    if (hi - lo > SG_PI) { f->t0 = hi; f->t1 = lo; } else { f->t0 = lo; f->t1 = hi; }
    return true;
}

static int bin_of(double theta, double inv_w, int nb)
{
    theta = std::fmod(theta, SG_TWO_PI);
    if (theta < 0) theta += SG_TWO_PI;
    int b = (int)std::floor(theta * inv_w);
    if (b < 0) b = 0;
    if (b >= nb) b = nb - 1;
    return b;
}

extern "C" int snowgpu_upload_table(snowgpu_ctx *ctx, int table_id, const double *xyr, int64_t k)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (table_id < 0 || table_id > (1 << 20) || k < 0 || (k > 0 && !xyr))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_upload_table: bad table id or size");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int nb = SG_NBINS;
    const double inv_w = nb / SG_TWO_PI;
    std::vector<SgEntry> fl((size_t)k);
    std::vector<int> b0((size_t)k), span((size_t)k);
    std::vector<uint32_t> count((size_t)nb + 1, 0);
    for (int64_t i = 0; i < k; ++i) {
        if (!derive_flake(xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2], &fl[(size_t)i])) {
            char buf[160];
            snprintf(buf, sizeof buf, "snowgpu_upload_table: row %lld (%g, %g, %g) is not a disk clear of the origin",
                     (long long)i, xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]);
            return fail(ctx, SNOWGPU_E_TABLE, buf);
        }
        SgEntry &f = fl[(size_t)i];
        f.src = (uint32_t)i;
        const double alpha = std::asin(std::min(1.0, f.r / f.rho));
        const double lo = f.phi - alpha - SG_BIN_MARGIN, hi = f.phi + alpha + SG_BIN_MARGIN;
        int s;
        if (hi - lo >= SG_TWO_PI - 2.0 / inv_w) { b0[(size_t)i] = 0; s = nb; }
        else {
            const int bl = bin_of(lo, inv_w, nb), bh = bin_of(hi, inv_w, nb);
            b0[(size_t)i] = bl;
            s = bh - bl;
            if (s < 0) s += nb;
            s += 1;
        }
        span[(size_t)i] = s;
        for (int t = 0; t < s; ++t) count[(size_t)((b0[(size_t)i] + t) % nb)]++;
    }
    std::vector<uint32_t> start((size_t)nb + 1, 0);
    for (int b = 0; b < nb; ++b) start[(size_t)b + 1] = start[(size_t)b] + count[(size_t)b];
    const size_t n_entries = start[(size_t)nb];
    if (n_entries > (size_t)64 * (size_t)std::max<int64_t>(k, 1) + 4096)
        return fail(ctx, SNOWGPU_E_TABLE, "snowgpu_upload_table: flakes so close to the sensor that they cover most azimuths");
    std::vector<SgEntry> entries(n_entries + 1);   // + one spare record: the scan prefetches entry e + 1
    std::memset(&entries[n_entries], 0, sizeof(SgEntry));
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < k; ++i) {
        for (int t = 0; t < span[(size_t)i]; ++t) {
            const int b = (b0[(size_t)i] + t) % nb;
            SgEntry e = fl[(size_t)i];
            e.flags = (t == 0) ? 1u : 0u;
            entries[fill[(size_t)b]++] = e;
        }
    }
    uint32_t max_bin = 0;
    for (int b = 0; b < nb; ++b) {
        std::sort(entries.begin() + (ptrdiff_t)start[(size_t)b], entries.begin() + (ptrdiff_t)start[(size_t)b + 1],
                  [](const SgEntry &p, const SgEntry &q) { return p.rho < q.rho || (p.rho == q.rho && p.src < q.src); });
        max_bin = std::max(max_bin, start[(size_t)b + 1] - start[(size_t)b]);
    }
    if ((size_t)table_id >= ctx->tables.size()) ctx->tables.resize((size_t)table_id + 1);
    DeviceTable &dt = ctx->tables[(size_t)table_id];
    if (dt.entries) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(dt.entries); dt.entries = nullptr; }
    if (dt.bin_start) { (void)hipFree(dt.bin_start); dt.bin_start = nullptr; }
    HIPCHK(ctx, hipMalloc((void **)&dt.entries, (n_entries + 1) * sizeof(SgEntry)));
    HIPCHK(ctx, hipMalloc((void **)&dt.bin_start, ((size_t)nb + 1) * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemcpy(dt.entries, entries.data(), (n_entries + 1) * sizeof(SgEntry), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dt.bin_start, start.data(), ((size_t)nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    dt.desc.entries = dt.entries;
    dt.desc.bin_start = dt.bin_start;
    dt.desc.n_bins = (uint32_t)nb;
    dt.desc.n_entries = (uint32_t)n_entries;
    dt.desc.inv_bin_w = inv_w;
    dt.desc.n_flakes = (uint32_t)k;
    dt.desc.max_bin = max_bin;
    ctx->tables_dirty = true;
    ctx->max_flakes = std::max(ctx->max_flakes, (uint32_t)k);
    return SNOWGPU_OK;
}

extern "C" int snowgpu_table_count(const snowgpu_ctx *ctx)
{
    if (!ctx) return 0;
    int n = 0;
    for (auto &t : ctx->tables) n += t.entries != nullptr;
    return n;
}

static int sync_tables(snowgpu_ctx *ctx)
{
    if (!ctx->tables_dirty) return SNOWGPU_OK;
    const size_t n = std::max<size_t>(ctx->tables.size(), 1);
    if (n > ctx->d_tables_cap) {
        if (ctx->d_tables) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->d_tables); }
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_tables, n * sizeof(SgTable)));
        ctx->d_tables_cap = n;
    }
    std::vector<SgTable> h(n);
    for (size_t i = 0; i < ctx->tables.size(); ++i) h[i] = ctx->tables[i].desc;
    HIPCHK(ctx, hipMemcpy(ctx->d_tables, h.data(), n * sizeof(SgTable), hipMemcpyHostToDevice));
    ctx->tables_dirty = false;
    return SNOWGPU_OK;
}

// Expected flakes per beam for a target at the table's edge ~ K * delta / (2 pi); real sweeps sit well
// below that (flakes in range scale with (d / R0)^2).  The first pass runs with the smallest list that most
// beams fit in -- its LDS footprint decides how many waves hide each other's latency -- and hands the rest
// to the next capacity.
static void choose_tiers(const snowgpu_ctx *ctx, double beam_div_deg, int tiers[4], int *n_tiers)
{
    const double expect = (double)ctx->max_flakes * (beam_div_deg * (SG_PI / 180.0)) / SG_TWO_PI;
    int n = 0;
    if (expect <= 12.0) tiers[n++] = 4;
    if (expect <= 24.0) tiers[n++] = 8;
    if (expect <= 40.0) tiers[n++] = 16;
    tiers[n++] = SG_LCAP;
    *n_tiers = n;
}

// ---- the batch launch sequence (everything on device pointers) --------------------------------------
struct BatchDev {
    int n_frames;
    int64_t n_total;
    int64_t max_frame;   // rows of the largest frame (host knowledge; n_total is a safe bound)
    int64_t uniform_rows = 0;   // > 0 when the host knows that all frames have this many rows
    const int64_t *frame_off;
    const void *rows;
    int dtype;
    const int32_t *table_ids;
    double beam_div_deg;
    const double *thr_poly;   // may be null -> prepass with plane
    const double *plane;
    double noise_floor;
    const int32_t *perm;      // may be null -> device sort
    void *out_rows;
    int32_t *out_src;
    int64_t *out_counts;
    int64_t *out_stats;
    double *out_thr_poly;     // may be null
    int32_t *status;
    hipStream_t stream;
    // debug tap
    int32_t *dbg_count = nullptr;
    double *dbg_rj = nullptr, *dbg_ratio = nullptr;
    int dbg_cap = 0;
    int32_t *perm_out = nullptr;   // where the permutation actually used lives (device)
};

static int run_batch(snowgpu_ctx *ctx, BatchDev &b)
{
    if (ctx->h_las.n <= 0) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_lasers has not been called");
    if (b.beam_div_deg <= 0 || b.beam_div_deg >= 45.0)
        return fail(ctx, SNOWGPU_E_INVALID, "beam divergence must be in (0, 45) degrees");
    int rc = sync_tables(ctx);
    if (rc) return rc;
    const size_t esz = b.dtype == 0 ? 4 : 8;
    const int64_t max_tiles = std::max<int64_t>(1, (b.max_frame + SG_TILE - 1) / SG_TILE);
    const size_t n = (size_t)b.n_total;
    hipStream_t st = b.stream;
    HIPCHK(ctx, hipMemsetAsync(b.status, 0, sizeof(int32_t) * 8, st));
    HIPCHK(ctx, hipMemsetAsync(b.status + 1, 0xff, sizeof(int32_t), st));   // status[1] = first offending row, -1 = none
    if (n == 0) {
        HIPCHK(ctx, hipMemsetAsync(b.out_counts, 0, sizeof(int64_t) * (size_t)b.n_frames, st));
        HIPCHK(ctx, hipMemsetAsync(b.out_stats, 0, sizeof(int64_t) * 3 * (size_t)b.n_frames, st));
        return SNOWGPU_OK;
    }
    // 0. noise-threshold prepass (simulation.py:449-467) unless the caller brought the polynomial.  Only the compaction
    // (the noise-floor decision) needs its result, so it runs on its own stream next to the sort and the per-beam
    // kernels: bandwidth-bound reductions beside latency-bound scans.
    const double *thr = b.thr_poly;
    bool pre_forked = false;
    if (!thr) {
        if (!b.plane) return fail(ctx, SNOWGPU_E_INVALID, "either thr_poly or plane must be given");
        ENSURE(ctx, ctx->thr_poly, (size_t)b.n_frames * 3);
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork0, st));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->aux2, ctx->ev_fork0, 0));
        int e = sg_prepass_run(&ctx->prepass, b.rows, b.dtype, b.frame_off, b.n_frames, b.n_total, b.max_frame, b.plane,
                               b.noise_floor, ctx->thr_poly.p, b.status, ctx->aux2);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("prepass: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
        thr = ctx->thr_poly.p;
        if (b.out_thr_poly)
            HIPCHK(ctx, hipMemcpyAsync(b.out_thr_poly, thr, sizeof(double) * 3 * (size_t)b.n_frames, hipMemcpyDeviceToDevice, ctx->aux2));
        HIPCHK(ctx, hipEventRecord(ctx->ev_join0, ctx->aux2));
        pre_forked = true;
    } else if (b.out_thr_poly) {
        HIPCHK(ctx, hipMemcpyAsync(b.out_thr_poly, thr, sizeof(double) * 3 * (size_t)b.n_frames, hipMemcpyDeviceToDevice, st));
    }
    // 1. channel sort (simulation.py:447)
    const int32_t *perm = b.perm;
    if (!perm) {
        ENSURE(ctx, ctx->tile_hist, (size_t)b.n_frames * (size_t)max_tiles * 256);
        ENSURE(ctx, ctx->tile_base, (size_t)b.n_frames * (size_t)max_tiles * 256);
        ENSURE(ctx, ctx->rank, n);
        ENSURE(ctx, ctx->perm, n);
        int e = sg_launch_sort(b.rows, b.dtype, b.frame_off, b.n_frames, b.n_total, ctx->tile_hist.p, ctx->tile_base.p,
                               ctx->rank.p, ctx->perm.p, b.status, max_tiles, st);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("sort launch: ") + hipGetErrorString((hipError_t)e));
        perm = ctx->perm.p;
    }
    b.perm_out = const_cast<int32_t *>(perm);
    // 1b. on the side stream, next to the prepass: table descriptors per (frame, channel) and the launch order of the
    // first pass -- by flake table (segments of the device sort) unless the caller brought the
    // permutation (no channel histogram then) or table ids are too sparse for the segment builder.
    const int64_t n_ft = (int64_t)b.n_frames * ctx->h_las.n;
    ENSURE(ctx, ctx->frame_tables, (size_t)n_ft);
    int tiers[4], n_tiers = 0;
    choose_tiers(ctx, b.beam_div_deg, tiers, &n_tiers);
    const int first_block = sg_beams_block(tiers[0]);
    const bool use_seg = !b.perm && !ctx->linear_order && ctx->tables.size() <= 65536 && b.n_frames <= (1 << 22)
                         && b.n_total < ((int64_t)1 << 31);
    if (use_seg) {
        const size_t P = (size_t)b.n_frames * 256;
        ENSURE(ctx, ctx->seg_tbl_cnt, ctx->tables.size() + 1); ENSURE(ctx, ctx->seg_tbl_base, ctx->tables.size() + 1);
        ENSURE(ctx, ctx->seg_blk, P); ENSURE(ctx, ctx->seg_cnt, P);
        ENSURE(ctx, ctx->seg_frame, P); ENSURE(ctx, ctx->seg_start, P); ENSURE(ctx, ctx->seg_n, 2);
        ENSURE(ctx, ctx->seg_of_blk, (size_t)((b.n_total + first_block - 1) / first_block) + P);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, st));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    {
        int e = sg_launch_resolve_tables(ctx->d_tables, (int)ctx->tables.size(), b.table_ids, n_ft, ctx->frame_tables.p, ctx->aux);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("table resolve launch: ") + hipGetErrorString((hipError_t)e));
        if (use_seg) {
            e = sg_launch_segments(b.frame_off, b.n_frames, ctx->tile_base.p, max_tiles, b.table_ids, ctx->h_las.n, (int)ctx->tables.size(), first_block,
                                   ctx->seg_tbl_cnt.p, ctx->seg_tbl_base.p, ctx->seg_blk.p, ctx->seg_start.p, ctx->seg_cnt.p, ctx->seg_frame.p,
                                   ctx->seg_n.p, ctx->seg_of_blk.p, ctx->aux);
            if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("segment launch: ") + hipGetErrorString((hipError_t)e));
        }
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
    // 3. beams
    ENSURE(ctx, ctx->rows_tmp, n * 5 * esz);
    ENSURE(ctx, ctx->keep, n);
    // flag bytes of the first pass (2: overflowed, 16 + n_flakes: queued for k_power): nothing stale may be left in them
    HIPCHK(ctx, hipMemsetAsync(ctx->keep.p, 0, n, st));
    ENSURE(ctx, ctx->diff2, (size_t)b.n_frames);
    const int32_t ovf_cap = (int32_t)std::min<size_t>(n, (size_t)1 << 24);
    ENSURE(ctx, ctx->ovf_list, (size_t)ovf_cap);
    ENSURE(ctx, ctx->ovf_list2, (size_t)ovf_cap);
    HIPCHK(ctx, hipMemsetAsync(ctx->diff2.p, 0, sizeof(unsigned long long) * (size_t)b.n_frames, st));
    SgBeamArgs a{};
    a.rows = b.rows; a.frame_off = b.frame_off; a.n_frames = b.n_frames; a.n_total = b.n_total; a.perm = perm;
    a.uniform_rows = (b.uniform_rows > 0 && b.n_total < ((int64_t)1 << 31)) ? b.uniform_rows : 0;
    a.inv_uniform_rows = a.uniform_rows > 0 ? 1.0f / (float)a.uniform_rows : 0.0f;
    a.tables = ctx->d_tables; a.n_tables = (int32_t)ctx->tables.size(); a.table_ids = b.table_ids; a.las = ctx->d_las;
    a.frame_tables = ctx->frame_tables.p;
    a.rgrid = ctx->d_rgrid; a.beam_div_deg = b.beam_div_deg; a.tmp_rows = ctx->rows_tmp.p;
    a.keep = ctx->keep.p; a.status = b.status; a.diff2 = ctx->diff2.p;
    a.dbg_count = b.dbg_count; a.dbg_rj = b.dbg_rj; a.dbg_ratio = b.dbg_ratio; a.dbg_cap = b.dbg_cap;
    a.exact_math = ctx->exact_math;
    a.phase_cycles = ctx->phase_cycles;
    ENSURE(ctx, ctx->ctile_cnt, 2 * ((size_t)b.n_frames * (size_t)max_tiles + 1));     // also the tiles of the overflow list builder
    ENSURE(ctx, ctx->ctile_base, 2 * ((size_t)b.n_frames * (size_t)max_tiles + 1));
    // pass t reads the overflow list of pass t-1 (counter status[1 + t]) and fills its own (status[2 + t]);
    // the counts live on the device, grids are sized for the worst case and idle blocks leave at once.
    int32_t *lists[2] = {ctx->ovf_list.p, ctx->ovf_list2.p};
    if (use_seg) {
        a.seg_blk = ctx->seg_blk.p; a.seg_start = ctx->seg_start.p; a.seg_cnt = ctx->seg_cnt.p; a.seg_frame = ctx->seg_frame.p;
        a.seg_n = ctx->seg_n.p; a.seg_of_blk = ctx->seg_of_blk.p;
        a.grid_blocks = (b.n_total + first_block - 1) / first_block + (int64_t)b.n_frames * 256;   // every non-empty pair wastes less than one block
    }
    HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    {   // queue of the first pass: one slot per row would always do; slots carry (range, ratio) x (capacity + 1)
        const size_t stride = 2 * ((size_t)tiers[0] + 1);
        if (n * stride * sizeof(double) > ((size_t)32 << 30))
            return fail(ctx, SNOWGPU_E_INVALID, "batch too large for the received-power queue of this table density: split it");
        if (b.n_frames >= (1 << 25)) return fail(ctx, SNOWGPU_E_INVALID, "too many frames in one batch");
        ENSURE(ctx, ctx->pq_list, n);
        ENSURE(ctx, ctx->ptile_cnt, 2 * ((n + SG_TILE - 1) / SG_TILE + 1));
        ENSURE(ctx, ctx->ptile_base, 2 * ((n + SG_TILE - 1) / SG_TILE + 1));
        ENSURE(ctx, ctx->pq_dict, n * stride);
        a.pq_list = ctx->pq_list.p; a.pq_dict = ctx->pq_dict.p; a.pq_count = b.status + 6; a.pq_cap = (int32_t)n; a.pq_stride = (int32_t)stride;
    }
    for (int t = 0; t < n_tiers; ++t) {
        if (t > 0) a.seg_blk = nullptr;
        a.work_list = t == 0 ? nullptr : lists[(t - 1) & 1];
        a.work_count = t == 0 ? nullptr : b.status + 1 + t;
        a.work_cap = ovf_cap;
        a.ovf_list = lists[t & 1];
        a.ovf_count = b.status + 2 + t;
        a.ovf_cap = ovf_cap;
        // measurement hooks: one event pair around ALL capacity tiers of the per-beam kernel
        const bool timed = ctx->prof && ctx->ev_used < (int)ctx->ev_start.size();
        if (timed && t == 0) { HIPCHK(ctx, hipEventRecord(ctx->ev_start[(size_t)ctx->ev_used], st)); ctx->prof_stream = st; }
        int e = sg_launch_beams(&a, b.dtype, tiers[t], st);
        if (!e && t == 0) {
            // the first pass queued the beams that met a flake: their received-power phase runs on the side stream,
            // next to the (latency-bound, mostly empty) later capacity tiers
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork2, st));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork2, 0));
            // two runs of the list, each in sorted-row order: beams with one flake (most of them), then the rest -- the
            // loops of k_power run S + 1 times, and a wave is as slow as its longest lane
            e = sg_launch_list(ctx->keep.p, b.n_total, ctx->ptile_cnt.p, ctx->ptile_base.p, ctx->pq_list.p, b.status + 6, (int32_t)n, 16, 17, 18, 255,
                               ctx->aux);
            if (!e) e = sg_launch_power(&a, b.dtype, tiers[0], ctx->aux);
            HIPCHK(ctx, hipEventRecord(ctx->ev_join2, ctx->aux));
        }
        if (!e && t == 0 && n_tiers > 1)   // the first pass flags its overflowed beams; build the ordered list from the flags
            e = sg_launch_list(ctx->keep.p, b.n_total, ctx->ctile_cnt.p, ctx->ctile_base.p, lists[0], b.status + 2, ovf_cap, 2, 2, 1, 0, st);
        if (t == n_tiers - 1) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join2, 0));
        if (timed && t == n_tiers - 1) { HIPCHK(ctx, hipEventRecord(ctx->ev_stop[(size_t)ctx->ev_used], st)); ctx->ev_used++; }
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("beam launch: ") + hipGetErrorString((hipError_t)e));
    }
    int e = 0;
    // 4. round + noise-floor filter + compaction + stats (simulation.py:516-530)
    if (pre_forked) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join0, 0));
    e = sg_launch_compact(ctx->rows_tmp.p, b.dtype, thr, ctx->keep.p, perm, b.frame_off, b.n_frames, b.n_total,
                          ctx->ctile_cnt.p, ctx->ctile_base.p, b.out_rows, b.out_src, b.out_counts, b.out_stats,
                          ctx->diff2.p, max_tiles, st);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("compaction launch: ") + hipGetErrorString((hipError_t)e));
    return SNOWGPU_OK;
}

static int status_to_error(snowgpu_ctx *ctx, const int32_t st[8])
{
    char buf[200];
    switch (st[0]) {
    case 0: return SNOWGPU_OK;
    case SNOWGPU_E_RANGE:
        snprintf(buf, sizeof buf, "index out of bounds for the %d-bin range grid: a simulated point lies at >= ~120 m (sorted row %d)", SG_RBINS, st[1]);
        return fail(ctx, SNOWGPU_E_RANGE, buf);
    case SNOWGPU_E_CHANNELS:
        return fail(ctx, SNOWGPU_E_CHANNELS, "channel column holds values other than integers in [0, 255]; pass an explicit permutation");
    case SNOWGPU_E_OVERFLOW:
        snprintf(buf, sizeof buf, "more than %d flakes intersect one beam (sorted row %d)", SG_LCAP, st[1]);
        return fail(ctx, SNOWGPU_E_OVERFLOW, buf);
    case SNOWGPU_E_GROUND:
        return fail(ctx, SNOWGPU_E_GROUND, "fewer than 3 ground points in a frame");
    case SNOWGPU_E_INVALID:
        return fail(ctx, SNOWGPU_E_INVALID, "a table id in table_ids was never uploaded");
    default:
        snprintf(buf, sizeof buf, "device status %d", st[0]);
        return fail(ctx, SNOWGPU_E_INVALID, buf);
    }
}

extern "C" int snowgpu_augment_batch_device(snowgpu_ctx *ctx, int n_frames, int64_t n_total, int64_t max_frame_rows,
                                            const int64_t *d_frame_offsets,
                                            const void *d_rows, int dtype, const int32_t *d_table_ids,
                                            double beam_divergence_deg, const double *d_thr_poly, const double *d_plane,
                                            double noise_floor, const int32_t *d_perm, void *d_out_rows, int32_t *d_out_src,
                                            int64_t *d_out_counts, int64_t *d_out_stats, double *d_out_thr_poly,
                                            int32_t *d_status, void *stream)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || n_total < 0 || !d_frame_offsets || (n_total > 0 && !d_rows) || !d_table_ids || !d_out_rows ||
        !d_out_src || !d_out_counts || !d_out_stats || !d_status || (dtype != 0 && dtype != 1))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_augment_batch_device: null pointer or bad dtype");
    if (n_total >= ((int64_t)1 << 31)) return fail(ctx, SNOWGPU_E_INVALID, "batch too large: split it below 2^31 rows");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    BatchDev b{};
    b.n_frames = n_frames; b.n_total = n_total; b.max_frame = (max_frame_rows > 0 && max_frame_rows <= n_total) ? max_frame_rows : n_total;
    b.frame_off = d_frame_offsets;
    b.uniform_rows = (max_frame_rows > 0 && max_frame_rows * (int64_t)n_frames == n_total) ? max_frame_rows : 0; b.rows = d_rows;
    b.dtype = dtype; b.table_ids = d_table_ids; b.beam_div_deg = beam_divergence_deg; b.thr_poly = d_thr_poly;
    b.plane = d_plane; b.noise_floor = noise_floor; b.perm = d_perm; b.out_rows = d_out_rows; b.out_src = d_out_src;
    b.out_counts = d_out_counts; b.out_stats = d_out_stats; b.out_thr_poly = d_out_thr_poly; b.status = d_status;
    b.stream = stream ? (hipStream_t)stream : ctx->stream;
    return run_batch(ctx, b);
}

static int host_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                      const int32_t *table_ids, double beam_div_deg, const double *thr_poly, const double *plane,
                      double noise_floor, const int32_t *perm, void *out_rows, int32_t *out_src, int64_t *out_counts,
                      int64_t *out_stats, double *out_thr_poly, int dbg_cap, int32_t *dbg_count, double *dbg_rj,
                      double *dbg_ratio, int32_t *perm_out)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || !frame_offsets || !table_ids || (dtype != 0 && dtype != 1))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_augment_batch: null pointer or bad dtype");
    if (frame_offsets[0] != 0) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets[0] must be 0");
    int64_t max_frame = 0;
    for (int f = 0; f < n_frames; ++f) {
        if (frame_offsets[f + 1] < frame_offsets[f]) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets must be non-decreasing");
        max_frame = std::max(max_frame, frame_offsets[f + 1] - frame_offsets[f]);
    }
    const int64_t n_total = frame_offsets[n_frames];
    if (n_total >= ((int64_t)1 << 31)) return fail(ctx, SNOWGPU_E_INVALID, "batch too large: split it below 2^31 rows");
    if (n_total > 0 && (!rows || !out_rows || !out_src)) return fail(ctx, SNOWGPU_E_INVALID, "null row buffers");
    if (!out_counts || !out_stats) return fail(ctx, SNOWGPU_E_INVALID, "null count/stat buffers");
    if (ctx->h_las.n <= 0) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_lasers has not been called");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t esz = dtype == 0 ? 4 : 8, n = (size_t)n_total;
    const size_t row_bytes = n * 5 * esz;
    hipStream_t st = ctx->stream;
    ENSURE(ctx, ctx->rows_in, std::max<size_t>(row_bytes, 8));
    ENSURE(ctx, ctx->rows_out, std::max<size_t>(row_bytes, 8));
    ENSURE(ctx, ctx->out_src, std::max<size_t>(n, 1));
    ENSURE(ctx, ctx->frame_off, (size_t)n_frames + 1);
    ENSURE(ctx, ctx->out_counts, (size_t)n_frames);
    ENSURE(ctx, ctx->out_stats, (size_t)n_frames * 3);
    ENSURE(ctx, ctx->table_ids, (size_t)n_frames * (size_t)ctx->h_las.n);
    ENSURE(ctx, ctx->thr_poly, (size_t)n_frames * 3);
    ENSURE(ctx, ctx->plane, (size_t)n_frames * 4);
    if (row_bytes) HIPCHK(ctx, hipMemcpyAsync(ctx->rows_in.p, rows, row_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->frame_off.p, frame_offsets, sizeof(int64_t) * ((size_t)n_frames + 1), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->table_ids.p, table_ids, sizeof(int32_t) * (size_t)n_frames * (size_t)ctx->h_las.n, hipMemcpyHostToDevice, st));
    // the user polynomial goes to its own buffer so that the prepass scratch (ctx->thr_poly) stays free
    DevBuf<double> &user_thr = ctx->user_thr;
    const double *d_thr = nullptr;
    if (thr_poly) {
        if (user_thr.ensure((size_t)n_frames * 3)) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed for thr_poly");
        HIPCHK(ctx, hipMemcpyAsync(user_thr.p, thr_poly, sizeof(double) * 3 * (size_t)n_frames, hipMemcpyHostToDevice, st));
        d_thr = user_thr.p;
    } else if (plane) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->plane.p, plane, sizeof(double) * 4 * (size_t)n_frames, hipMemcpyHostToDevice, st));
    }
    DevBuf<int32_t> &user_perm = ctx->user_perm;
    if (perm) {
        if (user_perm.ensure(std::max<size_t>(n, 1))) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed for perm");
        if (n) HIPCHK(ctx, hipMemcpyAsync(user_perm.p, perm, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    }
    DevBuf<double> &d_out_thr = ctx->out_thr;
    if (out_thr_poly && d_out_thr.ensure((size_t)n_frames * 3)) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed");
    BatchDev b{};
    b.n_frames = n_frames; b.n_total = n_total; b.max_frame = max_frame; b.frame_off = ctx->frame_off.p; b.rows = ctx->rows_in.p;
    {
        bool uni = max_frame > 0;
        for (int f = 0; f < n_frames && uni; ++f) uni = (frame_offsets[f + 1] - frame_offsets[f]) == max_frame;
        b.uniform_rows = uni ? max_frame : 0;
    }
    b.dtype = dtype; b.table_ids = ctx->table_ids.p; b.beam_div_deg = beam_div_deg; b.thr_poly = d_thr;
    b.plane = (!thr_poly && plane) ? ctx->plane.p : nullptr; b.noise_floor = noise_floor; b.perm = perm ? user_perm.p : nullptr;
    b.out_rows = ctx->rows_out.p; b.out_src = ctx->out_src.p; b.out_counts = ctx->out_counts.p; b.out_stats = ctx->out_stats.p;
    b.out_thr_poly = out_thr_poly ? d_out_thr.p : nullptr; b.status = ctx->d_status; b.stream = st;
    if (dbg_count) {
        ENSURE(ctx, ctx->dbg_count, std::max<size_t>(n, 1));
        ENSURE(ctx, ctx->dbg_rj, std::max<size_t>(n * (size_t)dbg_cap, 1));
        ENSURE(ctx, ctx->dbg_ratio, std::max<size_t>(n * (size_t)dbg_cap, 1));
        HIPCHK(ctx, hipMemsetAsync(ctx->dbg_count.p, 0, sizeof(int32_t) * std::max<size_t>(n, 1), st));
        b.dbg_count = ctx->dbg_count.p; b.dbg_rj = ctx->dbg_rj.p; b.dbg_ratio = ctx->dbg_ratio.p; b.dbg_cap = dbg_cap;
    }
    int rc = run_batch(ctx, b);
    int32_t status[8] = {0, -1, 0, 0, 0, 0, 0, 0};
    if (rc == SNOWGPU_OK) {
        HIPCHK(ctx, hipMemcpyAsync(status, ctx->d_status, sizeof status, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(out_counts, ctx->out_counts.p, sizeof(int64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(out_stats, ctx->out_stats.p, sizeof(int64_t) * 3 * (size_t)n_frames, hipMemcpyDeviceToHost, st));
        if (row_bytes) {
            HIPCHK(ctx, hipMemcpyAsync(out_rows, ctx->rows_out.p, row_bytes, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipMemcpyAsync(out_src, ctx->out_src.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
        }
        if (out_thr_poly) HIPCHK(ctx, hipMemcpyAsync(out_thr_poly, d_out_thr.p, sizeof(double) * 3 * (size_t)n_frames, hipMemcpyDeviceToHost, st));
        if (dbg_count && n) {
            HIPCHK(ctx, hipMemcpyAsync(dbg_count, ctx->dbg_count.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipMemcpyAsync(dbg_rj, ctx->dbg_rj.p, sizeof(double) * n * (size_t)dbg_cap, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipMemcpyAsync(dbg_ratio, ctx->dbg_ratio.p, sizeof(double) * n * (size_t)dbg_cap, hipMemcpyDeviceToHost, st));
        }
        if (perm_out && n && b.perm_out) HIPCHK(ctx, hipMemcpyAsync(perm_out, b.perm_out, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    }
    hipError_t se = hipStreamSynchronize(st);
    if (rc != SNOWGPU_OK) return rc;
    if (se != hipSuccess) return fail(ctx, SNOWGPU_E_HIP, std::string("stream synchronize: ") + hipGetErrorString(se));
    return status_to_error(ctx, status);
}

extern "C" int snowgpu_augment_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                                     const int32_t *table_ids, double beam_divergence_deg, const double *thr_poly,
                                     const double *plane, double noise_floor, const int32_t *perm, void *out_rows,
                                     int32_t *out_src, int64_t *out_counts, int64_t *out_stats, double *out_thr_poly)
{
    return host_batch(ctx, n_frames, frame_offsets, rows, dtype, table_ids, beam_divergence_deg, thr_poly, plane, noise_floor,
                      perm, out_rows, out_src, out_counts, out_stats, out_thr_poly, 0, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int snowgpu_debug_occlusions(snowgpu_ctx *ctx, int64_t n_rows, const void *rows, int dtype, const int32_t *table_ids,
                                        double beam_divergence_deg, int cap, int32_t *count, double *rj, double *ratio,
                                        int32_t *sorted_src)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_rows < 0 || cap <= 0 || !count || !rj || !ratio || !sorted_src) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_debug_occlusions: bad arguments");
    const int64_t off[2] = {0, n_rows};
    const double thr[3] = {0.0, 0.0, -1.0};   // keep everything
    const size_t esz = dtype == 0 ? 4 : 8;
    std::vector<unsigned char> out_rows((size_t)n_rows * 5 * esz + 8);
    std::vector<int32_t> out_src((size_t)n_rows + 1);
    int64_t cnt = 0, stats[3];
    return host_batch(ctx, 1, off, rows, dtype, table_ids, beam_divergence_deg, thr, nullptr, 0.7, nullptr, out_rows.data(),
                      out_src.data(), &cnt, stats, nullptr, cap, count, rj, ratio, sorted_src);
}

extern "C" int sg_sample_table(double occupancy, double scale_mm, double R0, uint64_t seed, int64_t n_cand, double *d_xyr,
                               int64_t cap, int64_t *n_rows, void *stream);   // snowgpu_sampler.hip

extern "C" int snowgpu_sample_table(snowgpu_ctx *ctx, int table_id, double occupancy_ratio, double diameter_scale_mm, double r_0,
                                    uint64_t seed, double *xyr_out, int64_t cap, int64_t *n_out)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (!(occupancy_ratio > 0) || !(occupancy_ratio < 0.05) || !(diameter_scale_mm > 0) || !(r_0 > 0.05) || !(r_0 <= 500.0) || !n_out)
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_sample_table: need 0 < occupancy < 0.05, scale > 0, 0.05 < R0 <= 500 m");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double target = occupancy_ratio * SG_PI * r_0 * r_0;
    const double s_m = diameter_scale_mm / 1000.0;
    const double mean_area = SG_PI * s_m * s_m / 3.0;         // E[pi r^2], r^2 = d^2/4 - h^2, h ~ U(-d/2, d/2), d ~ Exp(s)
    int64_t n_cand = (int64_t)(1.3 * target / mean_area) + 4096;
    std::vector<double> host;
    int64_t rows = 0;
    for (int attempt = 0;; ++attempt) {
        if (n_cand > ((int64_t)1 << 27)) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_sample_table: table would exceed 2^27 candidates");
        DevBuf<double> d_xyr;
        if (d_xyr.ensure((size_t)n_cand * 3)) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed for the sampled table");
        int rc = sg_sample_table(occupancy_ratio, diameter_scale_mm, r_0, seed, n_cand, d_xyr.p, n_cand, &rows, ctx->stream);
        if (rc == -2 && attempt < 4) { d_xyr.release(); n_cand *= 2; continue; }     // not enough darts: throw more
        if (rc != 0) {
            d_xyr.release();
            if (rc > 0) return fail(ctx, SNOWGPU_E_HIP, std::string("sampler: ") + hipGetErrorString((hipError_t)rc));
            return fail(ctx, SNOWGPU_E_TABLE, rc == -3 ? "sampler: a dart overlaps more than 4 earlier darts (occupancy too high for this sampler)"
                                                       : "sampler: acceptance did not settle / target area not reached");
        }
        host.resize((size_t)rows * 3);
        hipError_t e = rows ? hipMemcpy(host.data(), d_xyr.p, sizeof(double) * 3 * (size_t)rows, hipMemcpyDeviceToHost) : hipSuccess;
        d_xyr.release();
        if (e != hipSuccess) return fail(ctx, SNOWGPU_E_HIP, std::string("sampler copy: ") + hipGetErrorString(e));
        break;
    }
    *n_out = rows;
    if (xyr_out) {
        if (cap < rows) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_sample_table: output buffer too small (see *n_out)");
        std::memcpy(xyr_out, host.data(), sizeof(double) * 3 * (size_t)rows);
    }
    if (table_id >= 0) return snowgpu_upload_table(ctx, table_id, host.data(), rows);
    return SNOWGPU_OK;
}

extern "C" int snowgpu_set_exact_math(snowgpu_ctx *ctx, int on)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    ctx->exact_math = on ? 1 : 0;
    return SNOWGPU_OK;
}


// Experiments only (not in snowgpu.h): per-phase cycle counters of the per-beam kernel.
extern "C" int snowgpu_debug_phase_cycles(snowgpu_ctx *ctx, int enable, unsigned long long *out8)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (enable) {
        if (!ctx->phase_cycles) HIPCHK(ctx, hipMalloc((void **)&ctx->phase_cycles, 64 * sizeof(unsigned long long)));
        HIPCHK(ctx, hipMemset(ctx->phase_cycles, 0, 64 * sizeof(unsigned long long)));
    } else if (ctx->phase_cycles) {
        HIPCHK(ctx, hipDeviceSynchronize());
        if (out8) HIPCHK(ctx, hipMemcpy(out8, ctx->phase_cycles, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(ctx->phase_cycles);
        ctx->phase_cycles = nullptr;
    }
    return SNOWGPU_OK;
}

extern "C" int snowgpu_host_alloc(snowgpu_ctx *ctx, size_t bytes, void **ptr)
{
    if (!ctx || !ptr) return SNOWGPU_E_INVALID;
    *ptr = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostMalloc(ptr, std::max<size_t>(bytes, 8), hipHostMallocPortable));
    return SNOWGPU_OK;
}

extern "C" int snowgpu_host_free(snowgpu_ctx *ctx, void *ptr)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (!ptr) return SNOWGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostFree(ptr));
    return SNOWGPU_OK;
}

extern "C" int snowgpu_profile_begin(snowgpu_ctx *ctx, int max_launches)
{
    if (!ctx || max_launches <= 0) return SNOWGPU_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    while ((int)ctx->ev_start.size() < max_launches) {
        hipEvent_t a, b;
        HIPCHK(ctx, hipEventCreate(&a));
        HIPCHK(ctx, hipEventCreate(&b));
        ctx->ev_start.push_back(a);
        ctx->ev_stop.push_back(b);
    }
    ctx->ev_used = 0;
    ctx->prof = true;
    return SNOWGPU_OK;
}

extern "C" int snowgpu_profile_end(snowgpu_ctx *ctx, double *beam_kernel_ms, int *n_launches)
{
    if (!ctx || !beam_kernel_ms || !n_launches) return SNOWGPU_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->prof = false;
    double sum = 0.0;
    if (ctx->ev_used > 0) HIPCHK(ctx, hipEventSynchronize(ctx->ev_stop[(size_t)ctx->ev_used - 1]));
    for (int i = 0; i < ctx->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_start[(size_t)i], ctx->ev_stop[(size_t)i]));
        sum += ms;
    }
    *beam_kernel_ms = sum;
    *n_launches = ctx->ev_used;
    ctx->ev_used = 0;
    return SNOWGPU_OK;
}

// augment() followed by ground_water_augmentation() on its output (pointcloud_viewer.py:2807-2821), with the
// intermediate cloud staying on the device.
extern "C" int snowgpu_augment_wet_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                                         const int32_t *table_ids, double beam_divergence_deg, const double *thr_poly,
                                         const double *plane, double noise_floor, const int32_t *perm, const double *wet_plane,
                                         double water_height, double pavement_depth, double wet_noise_floor, double power_factor,
                                         int flat_earth, double delta, int replace, double *out_rows, int32_t *out_src,
                                         int64_t *out_counts, int64_t *out_stats, int32_t *out_flags)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || !frame_offsets || !wet_plane || !out_counts || !out_stats || !out_flags)
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_augment_wet_batch: null pointer");
    const int64_t n_total = frame_offsets[n_frames];
    const size_t n = (size_t)std::max<int64_t>(n_total, 0), esz = dtype == 0 ? 4 : 8;
    int64_t max_frame = 0;
    for (int f = 0; f < n_frames; ++f) max_frame = std::max(max_frame, frame_offsets[f + 1] - frame_offsets[f]);
    // stage 1: snowfall, results left on the device (ctx->rows_out / out_src / out_counts)
    std::vector<unsigned char> tmp_rows(n * 5 * esz + 8);
    std::vector<int32_t> snow_src(n + 1);
    std::vector<int64_t> snow_counts((size_t)n_frames);
    int rc = host_batch(ctx, n_frames, frame_offsets, rows, dtype, table_ids, beam_divergence_deg, thr_poly, plane, noise_floor, perm,
                        tmp_rows.data(), snow_src.data(), snow_counts.data(), out_stats, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    // stage 2: wet ground on the compacted rows still resident in ctx->rows_out (frame f: rows [off[f], off[f] + count[f]))
    hipStream_t st = ctx->stream;
    DevBuf<double> wet_out;
    DevBuf<int32_t> wet_src, wet_flags;
    DevBuf<int64_t> wet_counts;
    DevBuf<double> d_plane;
    if (wet_out.ensure(std::max<size_t>(n * 5, 1)) || wet_src.ensure(std::max<size_t>(n, 1)) || wet_flags.ensure((size_t)n_frames) ||
        wet_counts.ensure((size_t)n_frames) || d_plane.ensure((size_t)n_frames * 4))
        return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed (wet stage)");
    HIPCHK(ctx, hipMemcpyAsync(d_plane.p, wet_plane, sizeof(double) * 4 * (size_t)n_frames, hipMemcpyHostToDevice, st));
    SgWetParams wp{};
    wp.water_height = water_height; wp.pavement_depth = pavement_depth; wp.noise_floor = wet_noise_floor;
    wp.power_factor = power_factor; wp.flat_earth = flat_earth; wp.delta = delta; wp.replace = replace;
    int e = sg_wet_run(&ctx->prepass, ctx->rows_out.p, dtype, ctx->frame_off.p, ctx->out_counts.p, n_frames, n_total, max_frame,
                       d_plane.p, &wp, wet_out.p, wet_src.p, wet_counts.p, wet_flags.p, ctx->d_status, st);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("wet ground: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
    HIPCHK(ctx, hipMemcpyAsync(out_counts, wet_counts.p, sizeof(int64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(out_flags, wet_flags.p, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToHost, st));
    std::vector<int32_t> wsrc(n + 1);
    if (n) {
        HIPCHK(ctx, hipMemcpyAsync(out_rows, wet_out.p, n * 5 * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(wsrc.data(), wet_src.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    }
    hipError_t se = hipStreamSynchronize(st);
    wet_out.release(); wet_src.release(); wet_flags.release(); wet_counts.release(); d_plane.release();
    if (se != hipSuccess) return fail(ctx, SNOWGPU_E_HIP, std::string("stream synchronize: ") + hipGetErrorString(se));
    // compose the source indices: wet row -> snow row -> input row
    if (out_src)
        for (int f = 0; f < n_frames; ++f) {
            const int64_t b = frame_offsets[f];
            for (int64_t i = 0; i < out_counts[f]; ++i) out_src[b + i] = snow_src[(size_t)(b + wsrc[(size_t)(b + i)])];
        }
    return SNOWGPU_OK;
}

extern "C" int snowgpu_wet_ground_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                                        const double *plane, double water_height, double pavement_depth, double noise_floor,
                                        double power_factor, int flat_earth, double delta, int replace, double *out_rows,
                                        int32_t *out_src, int64_t *out_counts, int32_t *out_flags)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || !frame_offsets || !plane || !out_counts || !out_flags || (dtype != 0 && dtype != 1))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_wet_ground_batch: null pointer or bad dtype");
    int64_t max_frame = 0;
    for (int f = 0; f < n_frames; ++f) {
        if (frame_offsets[f + 1] < frame_offsets[f]) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets must be non-decreasing");
        max_frame = std::max(max_frame, frame_offsets[f + 1] - frame_offsets[f]);
    }
    const int64_t n_total = frame_offsets[n_frames];
    if (n_total > 0 && (!rows || !out_rows || !out_src)) return fail(ctx, SNOWGPU_E_INVALID, "null row buffers");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t esz = dtype == 0 ? 4 : 8, n = (size_t)n_total;
    hipStream_t st = ctx->stream;
    ENSURE(ctx, ctx->rows_in, std::max<size_t>(n * 5 * esz, 8));
    ENSURE(ctx, ctx->rows_out, std::max<size_t>(n * 5 * 8, 8));
    ENSURE(ctx, ctx->out_src, std::max<size_t>(n, 1));
    ENSURE(ctx, ctx->frame_off, (size_t)n_frames + 1);
    ENSURE(ctx, ctx->out_counts, (size_t)n_frames);
    ENSURE(ctx, ctx->plane, (size_t)n_frames * 4);
    ENSURE(ctx, ctx->dbg_count, (size_t)n_frames);   // reused as the per-frame "returned unchanged" flags
    if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->rows_in.p, rows, n * 5 * esz, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->frame_off.p, frame_offsets, sizeof(int64_t) * ((size_t)n_frames + 1), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->plane.p, plane, sizeof(double) * 4 * (size_t)n_frames, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t) * 8, st));
    SgWetParams wp{};
    wp.water_height = water_height; wp.pavement_depth = pavement_depth; wp.noise_floor = noise_floor;
    wp.power_factor = power_factor; wp.flat_earth = flat_earth; wp.delta = delta; wp.replace = replace;
    int e = sg_wet_run(&ctx->prepass, ctx->rows_in.p, dtype, ctx->frame_off.p, nullptr, n_frames, n_total, max_frame, ctx->plane.p, &wp,
                       (double *)ctx->rows_out.p, ctx->out_src.p, ctx->out_counts.p, ctx->dbg_count.p, ctx->d_status, st);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("wet ground: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
    HIPCHK(ctx, hipMemcpyAsync(out_counts, ctx->out_counts.p, sizeof(int64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(out_flags, ctx->dbg_count.p, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToHost, st));
    if (n) {
        HIPCHK(ctx, hipMemcpyAsync(out_rows, ctx->rows_out.p, n * 5 * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(out_src, ctx->out_src.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    return SNOWGPU_OK;
}
