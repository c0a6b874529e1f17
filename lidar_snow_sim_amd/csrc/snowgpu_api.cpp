// snowgpu_api.cpp -- the C ABI of libsnowgpu.so (include/snowgpu.h): context, table filing, scratch
// management and the launch sequence of one augment batch.  Host-side C++; every kernel lives in
// snowgpu_kernels.hip / snowgpu_prepass.hip.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/snowgpu.h"
#include "sg_common.h"
#include "sg_table_host.h"
#include "sg_prepass.h"
#include "sg_plane.h"

namespace {

struct DeviceTable {
    SgEntry *entries = nullptr;
    uint32_t *bin_start = nullptr;
    uint32_t *bin_q = nullptr;
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

// NUMA placement of the host threads that copy rows (packed result transfer).  On a two-socket host a core reaches the other socket's
// memory at a fraction of the speed: with free-roaming threads the same call gave 1.4 - 2.2 G points/s from run to run, with the threads
// on the wrong node 1.3, on the right one 2.3.  The right one is where the caller's row buffers live (asked of the kernel per call:
// get_mempolicy on their first pages); if that cannot be told -- a container may forbid the call -- the node the device hangs on.
static bool cpus_of_node(int node, cpu_set_t *out)
{
    char path[128];
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *fh = std::fopen(path, "r");
    if (!fh) return false;
    char list[4096] = {0};
    const bool ok = std::fgets(list, (int)sizeof list, fh) != nullptr;
    std::fclose(fh);
    if (!ok) return false;
    cpu_set_t allowed, node_set;
    CPU_ZERO(&allowed); CPU_ZERO(&node_set);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    for (char *p = list; *p;) {                       // "0-63,128-191"
        char *end = nullptr;
        long a = std::strtol(p, &end, 10), b = a;
        if (end == p) break;
        if (*end == '-') { p = end + 1; b = std::strtol(p, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, &node_set);
        if (*end != ',') break;
        p = end + 1;
    }
    if (CPU_COUNT(&node_set) == 0) return false;
    *out = node_set;
    return true;
}

static int node_of_device(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
    char path[256];
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *fh = std::fopen(path, "r");
    if (!fh) return -1;
    int node = -1;
    const int got = std::fscanf(fh, "%d", &node);
    std::fclose(fh);
    return got == 1 ? node : -1;
}

static int node_of_address(const void *p)
{
    if (!p) return -1;
    int node = -1;
    // get_mempolicy(&node, NULL, 0, addr, MPOL_F_NODE | MPOL_F_ADDR): the node of the page that holds addr
    const long rc = syscall(SYS_get_mempolicy, &node, nullptr, 0UL, const_cast<void *>(p), 1UL /* MPOL_F_NODE */ | 2UL /* MPOL_F_ADDR */);
    return rc == 0 ? node : -1;
}

// Host threads that put output rows together in the packed result transfer (snowgpu_set_result_transfer): plain copies, no arithmetic.
struct AsmPool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::deque<std::function<void()>> q;
    size_t pending = 0;
    bool stop = false;
    cpu_set_t want{};                 // where the threads should run (set_node), applied by each thread before its next job
    std::atomic<int> want_gen{0};
    int node = -2;
    void set_node(int nd)
    {
        if (nd == node) return;
        cpu_set_t c;
        if (nd < 0 || !cpus_of_node(nd, &c)) return;
        { std::lock_guard<std::mutex> lk(mu); want = c; node = nd; }
        want_gen.fetch_add(1);
    }
    void start(int n)
    {
        for (int i = 0; i < n; ++i)
            threads.emplace_back([this]() {
                int seen = 0;
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this]() { return stop || !q.empty(); });
                        if (q.empty()) return;
                        job = std::move(q.front());
                        q.pop_front();
                        if (seen != want_gen.load()) {      // (best effort: a forbidden call leaves the thread where it is)
                            seen = want_gen.load();
                            (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &want);
                        }
                    }
                    job();
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    void push(std::function<void()> job)
    {
        { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(job)); ++pending; }
        cv.notify_one();
    }
    void wait_idle()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [this]() { return pending == 0; });
    }
    ~AsmPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
};

}  // namespace

struct snowgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // side streams of one batch; each forks from the caller's stream and joins back before the compaction
    hipStream_t aux = nullptr;            // table resolve + segment order (next to the prepass), later k_power of the first pass
    hipStream_t aux2 = nullptr;           // noise-threshold prepass (only the compaction needs its result)
    hipStream_t aux3 = nullptr;           // later capacity tiers beyond the first of them
    int32_t *tier_hint_h = nullptr, *tier_hint_d = nullptr;   // beams per later tier of a recent batch, written by the device into page-locked host memory
    hipEvent_t ev_fork0 = nullptr, ev_join0 = nullptr;   // prepass
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;     // resolve / segments
    hipEvent_t ev_fp = nullptr, ev_join2 = nullptr;       // the pass over all rows (and its plan) done -> k_power_few / k_power
    hipEvent_t ev_few = nullptr;                          // k_power_few done -> (large batches) the tiers and the prepass
    hipEvent_t ev_lists = nullptr, ev_join3 = nullptr;   // tier lists built -> later tiers
    std::string err;
    std::vector<DeviceTable> tables;
    SgTable *d_tables = nullptr;      // device mirror of the descriptors
    size_t d_tables_cap = 0;
    bool tables_dirty = true;
    uint32_t max_flakes = 0;          // largest uploaded table (drives the capacity-tier choice)
    SgLasers h_las{};
    SgLasers *d_las = nullptr;
    double *d_rgrid = nullptr;
    int32_t *d_status = nullptr;      // 8 ints
    SgFov fov{};                      // camera-FOV crop applied by the compaction (snowgpu_set_fov)
    int fov_pre = 0;                  // also crop the INPUT rows before anything else (host entries; precompute.py:96-99)
    DevBuf<uint8_t> rows_crop;
    DevBuf<int32_t> crop_src, crop_out_src;
    DevBuf<int64_t> crop_counts, crop_off, crop_stats;
    // scratch shared by every batch
    DevBuf<int32_t> tile_hist, tile_base, perm, ctile_cnt, ctile_base, table_ids, out_src;
    DevBuf<uint8_t> srows;            // channel-sorted copy of the frames whose rows did not come channel-sorted (firing order)
    DevBuf<int32_t> tile_unsorted, frame_unsorted;
    DevBuf<unsigned long long> seg_tbl_cnt, seg_tbl_base;
    DevBuf<int32_t> seg_blk, seg_cnt, seg_frame, seg_n, seg_of_blk;
    DevBuf<int64_t> seg_start;
    DevBuf<uint32_t> rec, rec_q;      // result records: one per sorted position / per queue slot
    DevBuf<uint8_t> rng;              // range of every simulated beam, per sorted position, in the row dtype
    DevBuf<double> dq;                // dict queue of the first pass (SoA planes)
    DevBuf<int32_t> dq_g;
    DevBuf<uint16_t> dq_sc;
    DevBuf<unsigned long long> qn;    // per region: front | back << 32
    DevBuf<int2_t> pw_items;          // work items of k_power
    DevBuf<int32_t> back_list, bbase; // k_power_all: the multi-flake beams of all regions closed up; where each region's run goes
    DevBuf<double> ov;                // overflow slots of the pass over all rows (SG_OV_STRIDE doubles per sorted position)
    DevBuf<uint16_t> ov_sc;
    DevBuf<int32_t> tier_list, tier_sparse, tbase, redo_list;
    DevBuf<double> tq[SG_MAX_CLASSES];        // dict hand-over buffers of the list-mode tiers
    DevBuf<uint16_t> tq_sc[SG_MAX_CLASSES];
    DevBuf<double> h_lists;           // global-list tier: per-lane lists
    bool linear_order = false;   // experiments: SNOWGPU_LINEAR_ORDER=1 keeps the first pass in sorted-row order
    int64_t tier_cap_override = 0;    // tests: SNOWGPU_TIER_CAP=<entries> shrinks the hand-over buffers (in-place fallback runs)
    int first_tier_override = 0;      // tests: SNOWGPU_FIRST_TIER=4|8|16|63
    int few = 2;                      // SNOWGPU_FEW=0..3: beams with up to this many flakes go through k_power_few (0: all through k_power)
    int kp_all = 0;                   // SNOWGPU_KP_ALL=1: large batches run ONE work queue / persistent kernel (k_power_all) for what k_power_few leaves instead of k_power<4> / <8> / <16> side by side (A/B; same bytes)
    int kp_all_waves = 6;             // SNOWGPU_KP_ALL_WAVES: persistent one-wave blocks of k_power_all per CU (8 fit; the prepass runs beside it)
    int kp_all_ticket = 0;            // SNOWGPU_KP_ALL_TICKET=1: its waves draw items from an atomic cursor instead of striding
    int heavy_tail = -1;              // SNOWGPU_HEAVY_TAIL=0 / 1: never / always the long-tail order of the received-power phase (default: by the last batches' tier counts)
    int per_lane_scan = 0;            // experiments / validation: SNOWGPU_PER_LANE_SCAN=-1 wave scan in the tiers too
    bool tier_rows_auto = true;       // row kernels for the tiers of small batches (SNOWGPU_TIER_ROWS=0 switches that off too)
    bool tier_rows = false;           // SNOWGPU_TIER_ROWS=1: the later tiers as row kernels (snowgpu_rows.hip: G lanes per beam) -- measured slower, kept for A/B
    hipStream_t lane_stream[3] = {nullptr, nullptr, nullptr};     // snowgpu_lane_stream: one per priority level, made on demand
    int stats_early = -1;             // SNOWGPU_STATS_EARLY=0 / 1: the prepass' per-tile statistics inside the sort's first pass / as a kernel of their own on the prepass stream (default: the latter for batches of more than 16 frames)
    int prepass_with_few = -1;        // SNOWGPU_PREPASS_WITH_FEW=0 / 1: never / always start the prepass beside k_power_few (default: long-tail batches only)
    bool serial = false;              // experiments: SNOWGPU_SERIAL=1 keeps every kernel on the caller's stream (pure kernel times)
    DevBuf<int32_t> chunk_blk;
    DevBuf<uint16_t> rank;
    DevBuf<uint8_t> keep, rows_in, rows_out;
    DevBuf<int64_t> frame_off, out_counts, out_stats;
    DevBuf<double> thr_poly, plane, dbg_rj, dbg_ratio, user_thr, out_thr;
    DevBuf<int32_t> user_perm;
    DevBuf<int32_t> dbg_count;
    DevBuf<SgTable> frame_tables;
    SgPrepassScratch prepass{};
    // ground plane estimated on the device when a batch brings neither a plane nor a polynomial (planes.py:12-50)
    SgPlaneScratch plane_scr{};
    SgPlaneParams plane_par{SG_PLANE_REFERENCE, 1024, 5, 0, -1.55};
    DevBuf<double> plane_est, wet_plane_est;
    DevBuf<int32_t> plane_info;
    int64_t resident_rows = -1;       // rows snowgpu_prepass_stats left in rows_in (and their dtype): a following snowgpu_augment_batch with
    int resident_dtype = -1;          // rows == NULL computes on them instead of uploading the same rows again ...
    std::vector<int64_t> resident_off;   // ... if it names the same frames (frame offsets compared entry by entry)
    DevBuf<int32_t> stats_hist;       // snowgpu_prepass_stats: n_frames x 50 x 2555
    DevBuf<double> stats_rec;
    // fused snow + wet (snowgpu_augment_wet_batch*): the snowfall result stays here
    DevBuf<uint8_t> snow_rows;
    DevBuf<int32_t> snow_src, wet_flags;
    DevBuf<int64_t> snow_counts, wet_counts;
    DevBuf<double> wet_rows, wet_plane;
    // measurement hooks (snowgpu_profile_begin / _end)
    std::vector<hipEvent_t> ev_start, ev_stop;
    int ev_used = 0;
    bool prof = false;
    hipStream_t prof_stream = nullptr;
    int exact_math = 0;
    // Host-pointer batches run as a pipeline of chunks (whole frames, about pipe_rows rows each); see host_batch_pipelined.
    snowgpu_ctx *root = nullptr;          // set in a lane: the context whose tables, lasers and settings it computes with
    std::vector<snowgpu_ctx *> lanes;     // further compute lanes of the host pipeline (own stream, events and scratch), made on first use
    int pipe_lanes = 2;                   // SNOWGPU_PIPE_LANES: chunks computing side by side (lane 0 is the context itself).  Downloads are the
                                          // runtime's copy, i.e. the DMA engine: a copy kernel of ours was measured (scripts/probe/chain_probe.hip) --
                                          // while ANY kernel writes host memory every kernel boundary on the device waits for its outstanding
                                          // writes (3 us per dependent launch become 17 - 41 us) -- and dropped
    // The small arrays of a host-pointer batch cross the link as ONE block each way, through page-locked mailboxes: frame
    // offsets | table ids | planes or polynomials going up, status | counts | statistics | polynomials coming back (a
    // single sweep otherwise spends a quarter of its time on seven tiny dependent copies).
    char *mail_up_h = nullptr, *mail_dn_h = nullptr;
    size_t mail_up_cap = 0, mail_dn_cap = 0;
    DevBuf<uint8_t> mail_up_d, mail_dn_d;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    std::vector<hipEvent_t> pipe_ev;      // [2 c] chunk c has been uploaded, [2 c + 1] computed
    DevBuf<int64_t> pipe_off;         // chunk-local frame offsets of every chunk, concatenated
    DevBuf<int32_t> pipe_status;      // 8 status words per chunk
    int64_t pipe_rows = (int64_t)3 << 19;   // snowgpu_set_pipeline; 0: no pipeline (one upload, one download)
    std::vector<double> wet_lines;    // snowgpu_set_wet_lines: consumed by the next snowgpu_wet_ground_batch
    DevBuf<double> d_wet_lines;
    int wet_estimation = 0;           // snowgpu_set_wet_estimation: 0 'linear', 1 'poly' (seeded RANSAC on the device)
    uint64_t wet_seed = 0;
    DevBuf<double> wet_fit;           // n_frames x 8: the curves the last wet-ground call fitted (snowgpu_wet_last_fit)
    int wet_fit_frames = 0;
    int32_t h_status[8] = {0, -1, 0, 0, 0, 0, 0, 0};   // status words of the last host-pointer batch (tier counts summed over chunks)
    // Packed result transfer of the pipelined host entry (snowgpu_set_result_transfer): per kept row 4 + 4 (8 for float64 rows) bytes come
    // down the link, the moved coordinates of scattered rows apart; host threads copy x, y, z from the caller's input rows.
    int result_mode = 0;                  // 0: whole rows (+ source indices) over the link; 1: packed
    int asm_threads = 0;                  // host threads of the packed mode (0: the CPUs this process may use, minus two, at most 8)
    DevBuf<uint32_t> pk_meta;
    DevBuf<uint8_t> pk_int, pk_mv;
    DevBuf<int64_t> pk_mvcnt;
    DevBuf<int32_t> pk_tile_mv, pk_tile_mv_base;      // per lane
    char *st_pk = nullptr;                // page-locked staging: meta | intensities | moved coordinates | counts
    size_t st_pk_cap = 0;
    std::vector<hipEvent_t> pk_ev;        // [2 c] counts of chunk c on the host, [2 c + 1] its packed data
    AsmPool *pool = nullptr;
    double pk_times[4] = {0, 0, 0, 0};    // last packed call: ms until all enqueued, all downloads landed, all rows assembled; host bytes copied
    // The caller fits the noise threshold (snowgpu_set_threshold_callback): page-locked staging for the device half of the prepass --
    // histograms | records | status words per group | the polynomials the callback writes -- and one event per group
    // compact input of the call in flight (snowgpu_augment_batch_compact): the channel bytes; `rows` then are (x, y, z, intensity) float32
    const uint8_t *in_channels = nullptr;
    DevBuf<uint8_t> rows_c4, rows_ch;     // their device staging (16 + 1 bytes per row), expanded into rows_in by k_expand_rows
    snowgpu_threshold_fn thr_fn = nullptr;
    void *thr_user = nullptr;
    char *thr_stage = nullptr;
    size_t thr_stage_cap = 0;
    std::vector<hipEvent_t> thr_ev;
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

// streams and events of one launch sequence (a user-visible context, or a lane of its host pipeline)
static int init_streams(snowgpu_ctx *ctx)
{
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    {   // The prepass stream (aux2) sits in the HIGH-priority pool.  Rounds 1-3 needed that for scheduling (its scratch-array passes
        // were starved by the long-lived LDS-heavy blocks beside them: 5.04 vs 4.93 ms per step); for the lean prepass it no longer
        // matters on the device entry (4.62 vs 4.68 ms the other way round) -- but the runtime hands out hardware queues per priority
        // pool, and with aux2 in the normal pool the host pipeline's streams share queues: 1.47 instead of 1.85 G points/s through the
        // host entry (measured).
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, 0));
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->aux2, hipStreamNonBlocking, greatest));
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->aux3, hipStreamNonBlocking, 0));
    }
    for (hipEvent_t *ep : {&ctx->ev_fork0, &ctx->ev_join0, &ctx->ev_fork, &ctx->ev_join, &ctx->ev_join2, &ctx->ev_lists, &ctx->ev_join3, &ctx->ev_few, &ctx->ev_fp})
        HIPCHK(ctx, hipEventCreateWithFlags(ep, hipEventDisableTiming));
    if (hipHostMalloc((void **)&ctx->tier_hint_h, 64, hipHostMallocMapped) == hipSuccess) {
        std::memset(ctx->tier_hint_h, 0, 64);
        if (hipHostGetDevicePointer((void **)&ctx->tier_hint_d, ctx->tier_hint_h, 0) != hipSuccess) ctx->tier_hint_d = nullptr;
    } else { (void)hipGetLastError(); ctx->tier_hint_h = nullptr; }
    return SNOWGPU_OK;
}

// Upload / download streams and chunk events of the host pipeline: made on the first pipelined batch, so that a context that
// only ever sees device-resident batches keeps the normal-priority queue pool to its own streams (see host_batch_pipelined).
static int ensure_pipeline(snowgpu_ctx *ctx, int n_chunks, int n_lanes)
{
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (!ctx->s_h2d) HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->s_h2d, hipStreamNonBlocking, greatest));
    if (!ctx->s_d2h) HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->s_d2h, hipStreamNonBlocking, least));
    while ((int)ctx->pipe_ev.size() < 2 * n_chunks) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pipe_ev.push_back(e);
    }
    while ((int)ctx->lanes.size() < n_lanes - 1) {       // lane 0 is the context itself; the others: ONE stream each (low-priority pool)
        snowgpu_ctx *ln = new snowgpu_ctx();
        ln->device = ctx->device;
        ln->root = ctx;
        ctx->lanes.push_back(ln);
        if (hipStreamCreateWithPriority(&ln->stream, hipStreamNonBlocking, least) != hipSuccess) return fail(ctx, SNOWGPU_E_HIP, "lane stream");
        for (hipEvent_t *ep : {&ln->ev_fork0, &ln->ev_join0, &ln->ev_fork, &ln->ev_join, &ln->ev_join2, &ln->ev_lists, &ln->ev_join3, &ln->ev_few, &ln->ev_fp})
            HIPCHK(ctx, hipEventCreateWithFlags(ep, hipEventDisableTiming));
    }
    return SNOWGPU_OK;
}

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
    { const char *v = std::getenv("SNOWGPU_TIER_CAP"); ctx->tier_cap_override = v ? std::atoll(v) : 0; }
    { const char *v = std::getenv("SNOWGPU_FIRST_TIER"); ctx->first_tier_override = v ? std::atoi(v) : 0; }
    { const char *v = std::getenv("SNOWGPU_FEW"); ctx->few = v ? std::max(0, std::min(3, std::atoi(v))) : 2; }
    { const char *v = std::getenv("SNOWGPU_HEAVY_TAIL"); ctx->heavy_tail = v ? (v[0] == '1' ? 1 : 0) : -1; }
    { const char *v = std::getenv("SNOWGPU_SERIAL"); ctx->serial = v && v[0] == '1'; }
    { const char *v = std::getenv("SNOWGPU_STATS_EARLY"); ctx->stats_early = v ? (v[0] == '1' ? 1 : 0) : -1; }
    { const char *v = std::getenv("SNOWGPU_PREPASS_WITH_FEW"); ctx->prepass_with_few = v ? (v[0] == '1' ? 1 : 0) : -1; }
    { const char *v = std::getenv("SNOWGPU_KP_ALL"); if (v) ctx->kp_all = v[0] != '0'; }
    { const char *v = std::getenv("SNOWGPU_KP_ALL_WAVES"); if (v) ctx->kp_all_waves = std::min(std::max(std::atoi(v), 1), 8); }
    { const char *v = std::getenv("SNOWGPU_KP_ALL_TICKET"); if (v) ctx->kp_all_ticket = v[0] == '1'; }
    { const char *v = std::getenv("SNOWGPU_PER_LANE_SCAN"); ctx->per_lane_scan = v ? std::atoi(v) : 0; }
    { const char *v = std::getenv("SNOWGPU_TIER_ROWS"); ctx->tier_rows = v && v[0] == '1'; ctx->tier_rows_auto = !v; }
    // In a process that has loaded PyTorch's HIP runtime layer the runtime moves device-to-host copies with a full-grid blit
    // kernel, which stalls whatever computes beside it: one lane and larger chunks lose least there (1.8 instead of 1.3 G
    // points/s in-process).  SNOWGPU_PIPE_LANES / snowgpu_set_pipeline override either way.
    if (void *h = dlopen("libc10_hip.so", RTLD_NOLOAD | RTLD_LAZY)) {
        dlclose(h);
        ctx->pipe_lanes = 1;
        ctx->pipe_rows = (int64_t)3 << 20;
    }
    { const char *v = std::getenv("SNOWGPU_PIPE_LANES"); if (v) ctx->pipe_lanes = std::min(std::max(std::atoi(v), 1), 4); }
    int rc = init_streams(ctx);
    if (rc) return rc;
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
    for (snowgpu_ctx *ln : ctx->lanes) snowgpu_destroy(ln);      // a lane owns a stream, events and scratch only
    ctx->lanes.clear();
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t w : {ctx->s_h2d, ctx->s_d2h, ctx->lane_stream[0], ctx->lane_stream[1], ctx->lane_stream[2]}) if (w) { (void)hipStreamSynchronize(w); (void)hipStreamDestroy(w); }
    if (ctx->tier_hint_h) (void)hipHostFree(ctx->tier_hint_h);
    if (ctx->thr_stage) (void)hipHostFree(ctx->thr_stage);
    for (hipEvent_t e : ctx->thr_ev) (void)hipEventDestroy(e);
    if (ctx->mail_up_h) (void)hipHostFree(ctx->mail_up_h);
    if (ctx->mail_dn_h) (void)hipHostFree(ctx->mail_dn_h);
    ctx->mail_up_d.release(); ctx->mail_dn_d.release(); ctx->d_wet_lines.release(); ctx->wet_fit.release();
    delete ctx->pool; ctx->pool = nullptr;
    if (ctx->st_pk) (void)hipHostFree(ctx->st_pk);
    for (hipEvent_t e : ctx->pk_ev) (void)hipEventDestroy(e);
    ctx->pk_meta.release(); ctx->pk_int.release(); ctx->pk_mv.release(); ctx->pk_mvcnt.release(); ctx->pk_tile_mv.release(); ctx->pk_tile_mv_base.release();
    for (hipEvent_t e : ctx->pipe_ev) (void)hipEventDestroy(e);
    ctx->pipe_off.release(); ctx->pipe_status.release();
    for (auto &t : ctx->tables) {
        if (t.entries) (void)hipFree(t.entries);
        if (t.bin_start) (void)hipFree(t.bin_start);
        if (t.bin_q) (void)hipFree(t.bin_q);
    }
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->d_las) (void)hipFree(ctx->d_las);
    if (ctx->d_rgrid) (void)hipFree(ctx->d_rgrid);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    ctx->tile_hist.release(); ctx->tile_base.release(); ctx->perm.release(); ctx->srows.release(); ctx->tile_unsorted.release(); ctx->frame_unsorted.release();
    ctx->seg_tbl_cnt.release(); ctx->seg_tbl_base.release(); ctx->seg_blk.release(); ctx->seg_cnt.release(); ctx->seg_frame.release();
    ctx->seg_n.release(); ctx->seg_start.release(); ctx->seg_of_blk.release(); ctx->chunk_blk.release();
    ctx->rec.release(); ctx->rec_q.release(); ctx->rng.release(); ctx->dq.release(); ctx->dq_g.release(); ctx->dq_sc.release(); ctx->qn.release(); ctx->pw_items.release(); ctx->ov.release(); ctx->ov_sc.release();
    ctx->redo_list.release(); ctx->back_list.release(); ctx->bbase.release(); ctx->rows_c4.release(); ctx->rows_ch.release();
    ctx->tier_list.release(); ctx->tier_sparse.release(); ctx->tbase.release(); ctx->h_lists.release();
    for (int k = 0; k < SG_MAX_CLASSES; ++k) { ctx->tq[k].release(); ctx->tq_sc[k].release(); }
    ctx->ctile_cnt.release(); ctx->ctile_base.release(); ctx->table_ids.release(); ctx->out_src.release();
    ctx->rank.release(); ctx->keep.release(); ctx->rows_in.release(); ctx->rows_out.release();
    ctx->frame_off.release(); ctx->out_counts.release(); ctx->out_stats.release();
    ctx->thr_poly.release(); ctx->plane.release(); ctx->dbg_rj.release(); ctx->dbg_ratio.release();
    ctx->dbg_count.release(); ctx->frame_tables.release(); ctx->user_thr.release(); ctx->out_thr.release(); ctx->user_perm.release();
    ctx->snow_rows.release(); ctx->snow_src.release(); ctx->wet_flags.release(); ctx->snow_counts.release();
    ctx->wet_counts.release(); ctx->wet_rows.release(); ctx->wet_plane.release();
    ctx->rows_crop.release(); ctx->crop_src.release(); ctx->crop_out_src.release(); ctx->crop_counts.release(); ctx->crop_off.release(); ctx->crop_stats.release();
    sg_prepass_release(&ctx->prepass);
    sg_plane_release(&ctx->plane_scr);
    ctx->plane_est.release(); ctx->wet_plane_est.release(); ctx->plane_info.release(); ctx->stats_hist.release(); ctx->stats_rec.release();
    for (auto e : ctx->ev_start) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_stop) (void)hipEventDestroy(e);
    for (hipEvent_t e : {ctx->ev_fork0, ctx->ev_join0, ctx->ev_fork, ctx->ev_join, ctx->ev_join2, ctx->ev_lists, ctx->ev_join3, ctx->ev_few, ctx->ev_fp})
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {ctx->aux3, ctx->aux2, ctx->aux, ctx->stream})
        if (st) (void)hipStreamDestroy(st);
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
extern "C" int sg_table_index(const SgEntry *entries, const uint32_t *start, uint32_t *q, void *stream);   // snowgpu_tables.hip

// hand a filed table (device arrays, owned by the context from here on) to the table list under table_id
static int register_table(snowgpu_ctx *ctx, int table_id, SgEntry *entries, uint32_t *bin_start, uint32_t n_entries, uint32_t k,
                          uint32_t max_bin)
{
    if ((size_t)table_id >= ctx->tables.size()) ctx->tables.resize((size_t)table_id + 1);
    DeviceTable &dt = ctx->tables[(size_t)table_id];
    if (dt.entries || dt.bin_start) (void)hipStreamSynchronize(ctx->stream);     // no batch may still read the old table
    if (dt.entries) (void)hipFree(dt.entries);
    if (dt.bin_start) (void)hipFree(dt.bin_start);
    if (dt.bin_q) (void)hipFree(dt.bin_q);
    dt.entries = entries; dt.bin_start = bin_start; dt.bin_q = nullptr;
    dt.desc = SgTable{};
    ctx->tables_dirty = true;
    uint32_t *q = nullptr;
    int e = (int)hipMalloc((void **)&q, (size_t)SG_NBINS * SG_QSTEPS * sizeof(uint32_t));
    if (!e) e = sg_table_index(entries, bin_start, q, ctx->stream);
    if (!e) e = (int)hipStreamSynchronize(ctx->stream);
    if (e) {
        if (q) (void)hipFree(q);
        (void)hipFree(entries); (void)hipFree(bin_start);
        dt.entries = nullptr; dt.bin_start = nullptr;
        return fail(ctx, SNOWGPU_E_HIP, std::string("table index: ") + hipGetErrorString((hipError_t)e));
    }
    dt.bin_q = q;
    dt.desc.entries = entries;
    dt.desc.bin_start = bin_start;
    dt.desc.bin_q = q;
    dt.desc.n_bins = (uint32_t)SG_NBINS;
    dt.desc.n_entries = n_entries;
    dt.desc.inv_bin_w = SG_NBINS / SG_TWO_PI;
    dt.desc.n_flakes = k;
    dt.desc.max_bin = max_bin;
    ctx->tables_dirty = true;
    ctx->max_flakes = std::max(ctx->max_flakes, k);
    return SNOWGPU_OK;
}

extern "C" int sg_file_table_stage_a(const double *d_xyr, int64_t k, SgEntry *fl, int32_t *b0, int32_t *span, uint32_t *count,
                                     uint32_t *start, uint32_t *fill, int32_t *misc, void *stream);   // snowgpu_tables.hip
extern "C" int sg_file_table_stage_b(int64_t k, const SgEntry *fl, const int32_t *b0, const int32_t *span, const uint32_t *start,
                                     uint32_t *fill, SgEntry *tmp, SgEntry *entries, void *stream);
extern "C" int sg_table_dump(const SgEntry *entries, uint32_t n_entries, double *d_out, void *stream);

// File a table whose rows are in DEVICE memory (a table sampled there): derive, bin, sort on the device; only the record
// count comes back to size the allocation.
static int file_table_device(snowgpu_ctx *ctx, int table_id, const double *d_xyr, int64_t k)
{
    hipStream_t st = ctx->stream;
    const size_t nb = SG_NBINS;
    DevBuf<SgEntry> fl, tmp;
    DevBuf<int32_t> b0, span, misc;
    DevBuf<uint32_t> count, start, fill;
    auto cleanup = [&]() { fl.release(); tmp.release(); b0.release(); span.release(); misc.release(); count.release(); start.release(); fill.release(); };
    if (fl.ensure((size_t)std::max<int64_t>(k, 1)) || b0.ensure((size_t)std::max<int64_t>(k, 1)) || span.ensure((size_t)std::max<int64_t>(k, 1)) ||
        misc.ensure(2) || count.ensure(nb + 1) || start.ensure(nb + 1) || fill.ensure(nb + 1)) {
        cleanup();
        return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed while filing a table");
    }
    int e = sg_file_table_stage_a(d_xyr, k, fl.p, b0.p, span.p, count.p, start.p, fill.p, misc.p, st);
    uint32_t n_entries = 0;
    int32_t h_misc[2] = {0, 0};
    if (!e) e = (int)hipMemcpyAsync(&n_entries, start.p + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (!e) e = (int)hipMemcpyAsync(h_misc, misc.p, sizeof h_misc, hipMemcpyDeviceToHost, st);
    if (!e) e = (int)hipStreamSynchronize(st);
    if (e) { cleanup(); return fail(ctx, SNOWGPU_E_HIP, std::string("table filing: ") + hipGetErrorString((hipError_t)e)); }
    if (h_misc[0] > 0) {
        cleanup();
        char buf[128];
        snprintf(buf, sizeof buf, "table row %d is not a disk clear of the origin", 0x7fffffff - h_misc[0]);
        return fail(ctx, SNOWGPU_E_TABLE, buf);
    }
    if ((size_t)n_entries > (size_t)64 * (size_t)std::max<int64_t>(k, 1) + 4096) {
        cleanup();
        return fail(ctx, SNOWGPU_E_TABLE, "flakes so close to the sensor that they cover most azimuths");
    }
    SgEntry *d_entries = nullptr;
    uint32_t *d_start = nullptr;
    if (tmp.ensure((size_t)n_entries + 1) || hipMalloc((void **)&d_entries, ((size_t)n_entries + 1) * sizeof(SgEntry)) != hipSuccess ||
        hipMalloc((void **)&d_start, (nb + 1) * sizeof(uint32_t)) != hipSuccess) {
        if (d_entries) (void)hipFree(d_entries);
        cleanup();
        return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed while filing a table");
    }
    e = sg_file_table_stage_b(k, fl.p, b0.p, span.p, start.p, fill.p, tmp.p, d_entries, st);
    if (!e) e = (int)hipMemcpyAsync(d_start, start.p, (nb + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
    if (!e) e = (int)hipStreamSynchronize(st);
    cleanup();
    if (e) { (void)hipFree(d_entries); (void)hipFree(d_start); return fail(ctx, SNOWGPU_E_HIP, std::string("table filing: ") + hipGetErrorString((hipError_t)e)); }
    return register_table(ctx, table_id, d_entries, d_start, n_entries, (uint32_t)k, (uint32_t)h_misc[1]);
}

// snowgpu_upload_table for rows that already live in DEVICE memory (K x 3 float64): filed by kernels, no host copy.
extern "C" int snowgpu_file_table_device(snowgpu_ctx *ctx, int table_id, const double *d_xyr, int64_t n_flakes)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (table_id < 0 || table_id > (1 << 20) || n_flakes < 0 || (n_flakes > 0 && !d_xyr))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_file_table_device: bad table id or size");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return file_table_device(ctx, table_id, d_xyr, n_flakes);
}

// Debug / parity tap: the per-flake quantities of a filed table by table row -- range (simulation.py:332), azimuth
// (:351-352) and the two tangent angles ordered (right, left) (geometry.py:138-190, :32-80).  out: K x 4 doubles (host).
extern "C" int snowgpu_debug_table(snowgpu_ctx *ctx, int table_id, double *out, int64_t cap_rows)
{
    if (!ctx || !out) return SNOWGPU_E_INVALID;
    if (table_id < 0 || (size_t)table_id >= ctx->tables.size() || !ctx->tables[(size_t)table_id].entries)
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_debug_table: unknown table id");
    const DeviceTable &dt = ctx->tables[(size_t)table_id];
    const size_t k = dt.desc.n_flakes;
    if ((int64_t)k > cap_rows) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_debug_table: buffer too small");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<double> d;
    if (d.ensure(std::max<size_t>(k * 4, 1))) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed");
    int e = sg_table_dump(dt.entries, dt.desc.n_entries, d.p, ctx->stream);
    if (!e && k) e = (int)hipMemcpyAsync(out, d.p, sizeof(double) * 4 * k, hipMemcpyDeviceToHost, ctx->stream);
    if (!e) e = (int)hipStreamSynchronize(ctx->stream);
    d.release();
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("table dump: ") + hipGetErrorString((hipError_t)e));
    return SNOWGPU_OK;
}

extern "C" int snowgpu_upload_table(snowgpu_ctx *ctx, int table_id, const double *xyr, int64_t k)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (table_id < 0 || table_id > (1 << 20) || k < 0 || (k > 0 && !xyr))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_upload_table: bad table id or size");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int nb = SG_NBINS;
    std::vector<SgEntry> entries;
    std::vector<uint32_t> start;
    uint32_t max_bin = 0;
    int64_t bad_row = -1;
    const int frc = sg_file_table_host(xyr, k, entries, start, max_bin, &bad_row);
    if (frc == 1) {
        char buf[160];
        snprintf(buf, sizeof buf, "snowgpu_upload_table: row %lld (%g, %g, %g) is not a disk clear of the origin",
                 (long long)bad_row, xyr[3 * bad_row], xyr[3 * bad_row + 1], xyr[3 * bad_row + 2]);
        return fail(ctx, SNOWGPU_E_TABLE, buf);
    }
    if (frc == 2) return fail(ctx, SNOWGPU_E_TABLE, "snowgpu_upload_table: flakes so close to the sensor that they cover most azimuths");
    const size_t n_entries = start[(size_t)nb];
    SgEntry *d_entries = nullptr;
    uint32_t *d_start = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_entries, (n_entries + 1) * sizeof(SgEntry)));
    if (hipMalloc((void **)&d_start, ((size_t)nb + 1) * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(d_entries); return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed for bin offsets"); }
    HIPCHK(ctx, hipMemcpy(d_entries, entries.data(), (n_entries + 1) * sizeof(SgEntry), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(d_start, start.data(), ((size_t)nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    return register_table(ctx, table_id, d_entries, d_start, (uint32_t)n_entries, (uint32_t)k, max_bin);
}

extern "C" int snowgpu_free_table(snowgpu_ctx *ctx, int table_id)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (table_id < 0 || (size_t)table_id >= ctx->tables.size()) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_free_table: unknown table id");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DeviceTable &dt = ctx->tables[(size_t)table_id];
    if (dt.entries || dt.bin_start) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (dt.entries) { (void)hipFree(dt.entries); dt.entries = nullptr; }
    if (dt.bin_start) { (void)hipFree(dt.bin_start); dt.bin_start = nullptr; }
    if (dt.bin_q) { (void)hipFree(dt.bin_q); dt.bin_q = nullptr; }
    dt.desc = SgTable{};
    ctx->tables_dirty = true;
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
    // (measured on the 40 k-flake tables of C1, expect = 19: a 4-entry first pass is 5 % faster than an 8-entry one, a 16-entry one
    // half as fast -- the lists' LDS footprint decides the occupancy of the pass over ALL rows, the tiers only see the long ones)
    int first = expect <= 24.0 ? 4 : (expect <= 48.0 ? 8 : (expect <= 96.0 ? 16 : SG_LCAP));
    if (ctx->first_tier_override == 4 || ctx->first_tier_override == 8 || ctx->first_tier_override == 16 ||
        ctx->first_tier_override == SG_LCAP)
        first = ctx->first_tier_override;
    int n = 0;
    for (int c : {4, 8, 16, SG_LCAP})
        if (c >= first) tiers[n++] = c;
    *n_tiers = n;
}

// dict hand-over buffer of a list-mode tier: entries it holds for a batch of n rows (the rest of the class, if any,
// runs the received-power phase in place)
static int64_t tier_queue_cap(const snowgpu_ctx *ctx, int lmax, int64_t n)
{
    if (ctx->tier_cap_override > 0) return std::min<int64_t>(ctx->tier_cap_override, std::max<int64_t>(n, 1));
    // A buffer for every row while that costs at most 1 GiB per tier: it saves the launch of the in-place fallback pass -- a
    // chip-sized grid that finds nothing to do but sits in the chain of dependent launches a small batch is bound by.  This
    // includes the 1.5 M-row chunks of the host pipeline: 0.33 GB (8 entries) + 0.63 GB (16 entries) per context and compute
    // lane, i.e. about 1 GB per lane of the 288 GB (DESIGN.md section 3 lists it).  Beyond 1 GiB: the fractions below.
    const int64_t slot_bytes = (int64_t)sizeof(double) * (3 * (int64_t)lmax + 2) + 2;
    if (std::max<int64_t>(n, 1) * slot_bytes <= ((int64_t)1 << 30)) return std::max<int64_t>(n, 1);
    const int64_t div = lmax <= 8 ? 4 : (lmax <= 16 ? 16 : 64);
    return std::min<int64_t>(std::max<int64_t>(n, 1), std::max<int64_t>(n / div, 4096));
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
    bool no_fov = false;           // debug tap: never crop
    bool want_perm = false;        // the caller reads perm_out back: the sort writes the permutation of channel-sorted frames too
    SgPackOut *pack = nullptr;     // packed result transfer: the compaction writes these instead of out_rows / out_src (tile scratch filled in here)
    bool serial = false;           // every kernel on `stream`: no fork / join events (chunks of the host pipeline)
    bool defer_thr = false;        // the caller fits the noise threshold itself while the per-beam kernels run (snowgpu_set_threshold_callback):
                                   // run_batch stops ahead of the compaction, launches no prepass; run_compaction finishes with b.thr_poly
};

static int launch_compaction(snowgpu_ctx *ctx, BatchDev &b, const int32_t *perm, const double *thr, size_t regions, int64_t max_tiles);

static int run_batch(snowgpu_ctx *ctx, BatchDev &b)
{
    snowgpu_ctx *R = ctx->root ? ctx->root : ctx;      // a lane computes on the tables / lasers / settings of its root
    if (R->h_las.n <= 0) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_lasers has not been called");
    if (b.beam_div_deg <= 0 || b.beam_div_deg >= 45.0)
        return fail(ctx, SNOWGPU_E_INVALID, "beam divergence must be in (0, 45) degrees");
    int rc = sync_tables(R);
    if (rc) { if (R != ctx) ctx->err = R->err; return rc; }
    const int64_t max_tiles = std::max<int64_t>(1, (b.max_frame + SG_TILE - 1) / SG_TILE);
    const size_t n = (size_t)b.n_total;
    const size_t esz = b.dtype == 0 ? 4 : 8;
    hipStream_t st = b.stream;
    const bool serial = R->serial || b.serial;
    hipStream_t s_aux = serial ? st : ctx->aux, s_aux2 = serial ? st : ctx->aux2, s_aux3 = serial ? st : ctx->aux3;
    HIPCHK(ctx, hipMemsetAsync(b.status, 0, sizeof(int32_t) * 8, st));      // (status[1] = first offending row, -1 = none: set by the first kernel)
    if (n == 0) {
        HIPCHK(ctx, hipMemsetAsync(b.status + 1, 0xff, sizeof(int32_t), st));
        HIPCHK(ctx, hipMemsetAsync(b.out_counts, 0, sizeof(int64_t) * (size_t)b.n_frames, st));
        HIPCHK(ctx, hipMemsetAsync(b.out_stats, 0, sizeof(int64_t) * 3 * (size_t)b.n_frames, st));
        if (b.pack) HIPCHK(ctx, hipMemsetAsync(b.pack->mv_counts, 0, sizeof(int64_t) * (size_t)b.n_frames, st));
        return SNOWGPU_OK;
    }
    // 0. noise-threshold prepass (simulation.py:449-467) unless the caller brought the polynomial.  Only the compaction
    // (the noise-floor decision) needs its result, so it runs on its own stream next to the received-power kernels:
    // bandwidth-bound reductions beside latency-bound persistent waves.
    const double *thr = b.thr_poly;
    bool pre_forked = false;
    // The prepass' per-tile statistics ride on the channel sort's first pass over the rows when the plane is known by then: a
    // caller's plane, or the reference-today plane (a constant).  Estimated planes (least squares, RANSAC) come later, on the
    // prepass stream, and the statistics keep their own pass.
    const bool fuse_stats = !b.thr_poly && !b.defer_thr && !b.perm && (b.plane != nullptr || R->plane_par.method == SG_PLANE_REFERENCE);
    const double *early_plane = b.plane;
    double *lean_part = nullptr;
    if (fuse_stats) {
        if (!early_plane) {
            ENSURE(ctx, ctx->plane_est, (size_t)b.n_frames * 4);
            ENSURE(ctx, ctx->plane_info, (size_t)b.n_frames * 4);
            int pe = sg_plane_run(&ctx->plane_scr, &R->plane_par, b.rows, b.dtype, b.frame_off, nullptr, b.n_frames, b.n_total, b.max_frame,
                                  ctx->plane_est.p, ctx->plane_info.p, st);
            if (pe) return fail(ctx, SNOWGPU_E_HIP, std::string("plane estimate: ") + (pe > 0 ? hipGetErrorString((hipError_t)pe) : "allocation"));
            early_plane = ctx->plane_est.p;
        }
        lean_part = sg_prepass_reserve_tiles(&ctx->prepass, b.n_frames, b.max_frame);
        if (!lean_part) return fail(ctx, SNOWGPU_E_HIP, "prepass: allocation");
    }
    bool hist_early = false;
    // Large batches take the statistics out of the sort again: as a kernel of their own on the prepass stream, behind the histogram fill, they
    // run beside the sort and the scan (the sort's first pass 0.33 -> 0.20 ms on the step's critical path, the scan a little slower for the
    // company: C2 - 0.7 %, C2fire - 0.8 %, C3 - 0.9 % on one box).  Same sums in the same order as inside the sort (k_lean_stats deals the
    // rows to its threads as k_sort_hist does): same bits.  Only with a caller's plane -- nothing on `st` has to make it first.
    const bool stats_early = fuse_stats && !serial && early_plane == b.plane && (R->stats_early < 0 ? b.n_frames > 16 : R->stats_early == 1);
    auto launch_prepass = [&]() -> int {
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork0, st));
        HIPCHK(ctx, hipStreamWaitEvent(s_aux2, ctx->ev_fork0, 0));
        const double *pl = fuse_stats ? early_plane : b.plane;
        if (!pl) {                                  // simulation.py:449 calculate_plane(pc): on the device, by the context's method
            ENSURE(ctx, ctx->plane_est, (size_t)b.n_frames * 4);
            ENSURE(ctx, ctx->plane_info, (size_t)b.n_frames * 4);
            int pe = sg_plane_run(&ctx->plane_scr, &R->plane_par, b.rows, b.dtype, b.frame_off, nullptr, b.n_frames, b.n_total, b.max_frame,
                                  ctx->plane_est.p, ctx->plane_info.p, s_aux2);
            if (pe) return fail(ctx, SNOWGPU_E_HIP, std::string("plane estimate: ") + (pe > 0 ? hipGetErrorString((hipError_t)pe) : "allocation"));
            pl = ctx->plane_est.p;
        }
        int e = sg_prepass_run(&ctx->prepass, b.rows, b.dtype, b.frame_off, b.n_frames, b.n_total, b.max_frame, pl,
                               b.noise_floor, ctx->thr_poly.p, b.status, s_aux2, fuse_stats ? 1 : 0, ctx->srows.p, ctx->frame_unsorted.p, hist_early ? 1 : 0);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("prepass: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
        if (b.out_thr_poly)
            HIPCHK(ctx, hipMemcpyAsync(b.out_thr_poly, ctx->thr_poly.p, sizeof(double) * 3 * (size_t)b.n_frames, hipMemcpyDeviceToDevice, s_aux2));
        HIPCHK(ctx, hipEventRecord(ctx->ev_join0, s_aux2));
        pre_forked = true;
        return SNOWGPU_OK;
    };
    if (!thr && b.defer_thr) {
        // (no prepass here: the caller fits the polynomial from the device half it already has; run_compaction brings it)
    } else if (!thr) {
        ENSURE(ctx, ctx->thr_poly, (size_t)b.n_frames * 3);
        thr = ctx->thr_poly.p;
        if (!serial) {           // the prepass' histogram fill runs on ITS stream, beside the sort
            // Forked from the caller's stream FIRST: the fill is then ordered behind whatever the caller queued before this call (the wet
            // kernels of a fused call fill and read the same histogram on `st`), and a stream capture of this call records it in the graph --
            // issued on a stream that has not joined the capture it ran once, eagerly, and every replay but the first added into a stale
            // histogram.
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork0, st));
            HIPCHK(ctx, hipStreamWaitEvent(s_aux2, ctx->ev_fork0, 0));
            int he = sg_prepass_clear_hist(&ctx->prepass, b.n_frames, s_aux2);
            if (he) return fail(ctx, SNOWGPU_E_HIP, std::string("prepass: ") + (he > 0 ? hipGetErrorString((hipError_t)he) : "allocation"));
            hist_early = true;
            if (fuse_stats && stats_early) {
                int se = sg_prepass_stats_early(&ctx->prepass, b.rows, b.dtype, b.frame_off, b.n_frames, b.max_frame, early_plane, s_aux2);
                if (se) return fail(ctx, SNOWGPU_E_HIP, std::string("prepass statistics: ") + (se > 0 ? hipGetErrorString((hipError_t)se) : "allocation"));
            }
        }
    } else if (b.out_thr_poly) {
        HIPCHK(ctx, hipMemcpyAsync(b.out_thr_poly, thr, sizeof(double) * 3 * (size_t)b.n_frames, hipMemcpyDeviceToDevice, st));
    }
    // 1. channel sort (simulation.py:447).  A frame whose rows come channel-sorted (channel-major sweeps) keeps the identity and is read
    // in place; any other frame (firing order, as in an STF .bin) gets a sorted copy from the sort's second pass.  Either way sorted
    // position g is "row g" of one of the two arrays for everything downstream: no gather through the permutation.
    const int32_t *perm = b.perm;
    ENSURE(ctx, ctx->srows, n * 5 * esz);
    ENSURE(ctx, ctx->frame_unsorted, (size_t)b.n_frames);
    if (!perm) {
        ENSURE(ctx, ctx->tile_hist, (size_t)b.n_frames * (size_t)max_tiles * 256);
        ENSURE(ctx, ctx->tile_base, (size_t)b.n_frames * (size_t)max_tiles * 256);
        ENSURE(ctx, ctx->tile_unsorted, (size_t)b.n_frames * (size_t)max_tiles);
        ENSURE(ctx, ctx->rank, n);
        ENSURE(ctx, ctx->perm, n);
        ENSURE(ctx, ctx->keep, n);                // channel bytes between the two sort passes; flag / keep bytes afterwards
        // (first pass and per-frame scan here; the second pass -- the sorted copy of unsorted frames -- further down, so that the segment
        // builder, which needs the scan only, runs on its side stream beside it)
        int e = sg_launch_sort(b.rows, b.dtype, b.frame_off, b.n_frames, b.n_total, ctx->tile_hist.p, ctx->tile_base.p,
                               ctx->rank.p, ctx->keep.p, ctx->perm.p, b.status, max_tiles, (fuse_stats && !stats_early) ? early_plane : nullptr, (fuse_stats && !stats_early) ? lean_part : nullptr,
                               ctx->tile_unsorted.p, ctx->frame_unsorted.p, ctx->srows.p, b.want_perm ? 1 : 0, 1, st);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("sort launch: ") + hipGetErrorString((hipError_t)e));
        perm = ctx->perm.p;
    } else {
        int e = sg_launch_gather_rows(b.rows, b.dtype, b.frame_off, b.n_frames, b.n_total, b.max_frame, perm, ctx->srows.p, ctx->frame_unsorted.p, b.status, st);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("gather launch: ") + hipGetErrorString((hipError_t)e));
    }
    b.perm_out = const_cast<int32_t *>(perm);
    // 1b. on the side stream: table descriptors per (frame, channel) and the launch order of the pass over all rows -- by flake
    // table (segments of the device sort) unless the caller brought the permutation (no channel histogram
    // then) or table ids are too sparse for the segment builder.
    const int64_t n_ft = (int64_t)b.n_frames * R->h_las.n;
    ENSURE(ctx, ctx->frame_tables, (size_t)n_ft);
    int tiers[4], n_tiers = 0;
    choose_tiers(R, b.beam_div_deg, tiers, &n_tiers);
    const int first_block = sg_beams_block(tiers[0]);
    const bool use_seg = !b.perm && !R->linear_order && R->tables.size() <= 65536 && b.n_frames <= (1 << 22)
                         && b.n_total < ((int64_t)1 << 31);
    if (use_seg) {
        const size_t P = (size_t)b.n_frames * 256;
        ENSURE(ctx, ctx->seg_tbl_cnt, (R->tables.size() + 1) * SG_TBL_STRIDE); ENSURE(ctx, ctx->seg_tbl_base, R->tables.size() + 1);
        ENSURE(ctx, ctx->seg_blk, P); ENSURE(ctx, ctx->seg_cnt, P);
        ENSURE(ctx, ctx->seg_frame, P); ENSURE(ctx, ctx->seg_start, P); ENSURE(ctx, ctx->seg_n, 2);
        ENSURE(ctx, ctx->seg_of_blk, ((size_t)((b.n_total + first_block - 1) / first_block) + P) * SG_BLKREC);
        ENSURE(ctx, ctx->chunk_blk, 2);
    }
    // (The pass over all rows is ONE launch.  Cut into several, with k_power of one range beside the scan of the next, it gained
    // nothing: both are bound by the LDS their lists need, so sharing a CU only trades waves.)
    const int64_t total_blocks_ub = (b.n_total + first_block - 1) / first_block + (use_seg ? (int64_t)b.n_frames * 256 : 0);
    // Everything this step counts up from zero lies in ONE block, cleared by one fill (each fill is a launch on the chain between
    // the sort and the scan): per region the queue counter and the SG_MAX_CLASSES tier-list counters; per frame the intensity
    // statistics; the tier lists' lengths, the work-item counters, the row kernels' redo counters.
    static_assert(SG_MAX_CLASSES * sizeof(int32_t) == 2 * sizeof(unsigned long long), "a region's tier counters are two 64-bit words");
    const size_t q_chunk = 8 * (size_t)first_block;
    const size_t regions = std::max<size_t>((size_t)b.n_frames * 256, n / q_chunk + 2);
    const size_t zero_words = 3 * regions + 2 * (size_t)b.n_frames + 8;      // (.. and per frame the compaction's count of finished tiles)
    ENSURE(ctx, ctx->qn, zero_words);
    ENSURE(ctx, ctx->tbase, SG_MAX_CLASSES * regions);
    // A batch of up to four frames builds its segments (and clears the zero block) with ONE block on the caller's stream: no fill, no hop
    // to the side stream and back between the sort and the scan.
    bool seg_small = false;
    if (use_seg) {
        const int se = sg_launch_segments_small(b.frame_off, b.n_frames, ctx->tile_base.p, max_tiles, b.table_ids, R->h_las.n, (int)R->tables.size(), first_block,
                                                ctx->seg_blk.p, ctx->seg_start.p, ctx->seg_cnt.p, ctx->seg_frame.p, ctx->seg_n.p, ctx->seg_of_blk.p,
                                                ctx->chunk_blk.p, R->d_tables, ctx->frame_tables.p, ctx->qn.p, (int64_t)zero_words, st);
        if (se > 0) return fail(ctx, SNOWGPU_E_HIP, std::string("segment launch: ") + hipGetErrorString((hipError_t)se));
        seg_small = se == 0;
    }
    if (!seg_small) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork, st));
        HIPCHK(ctx, hipStreamWaitEvent(s_aux, ctx->ev_fork, 0));
        int e = 0;
        if (!use_seg) e = sg_launch_resolve_tables(R->d_tables, (int)R->tables.size(), b.table_ids, n_ft, ctx->frame_tables.p, s_aux);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("table resolve launch: ") + hipGetErrorString((hipError_t)e));
        if (use_seg) {                                // (its first kernel resolves the table descriptors on the way)
            e = sg_launch_segments(b.frame_off, b.n_frames, ctx->tile_base.p, max_tiles, b.table_ids, R->h_las.n, (int)R->tables.size(), first_block,
                                   ctx->seg_tbl_cnt.p, ctx->seg_tbl_base.p, ctx->seg_blk.p, ctx->seg_start.p, ctx->seg_cnt.p, ctx->seg_frame.p,
                                   ctx->seg_n.p, ctx->seg_of_blk.p, ctx->chunk_blk.p, R->d_tables, ctx->frame_tables.p, s_aux);
            if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("segment launch: ") + hipGetErrorString((hipError_t)e));
        }
        HIPCHK(ctx, hipEventRecord(ctx->ev_join, s_aux));
    }
    if (!b.perm) {
        int e = sg_launch_sort(b.rows, b.dtype, b.frame_off, b.n_frames, b.n_total, ctx->tile_hist.p, ctx->tile_base.p,
                               ctx->rank.p, ctx->keep.p, ctx->perm.p, b.status, max_tiles, nullptr, nullptr,
                               ctx->tile_unsorted.p, ctx->frame_unsorted.p, ctx->srows.p, b.want_perm ? 1 : 0, 2, st);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("sort launch: ") + hipGetErrorString((hipError_t)e));
    }
    // 3. beams
    ENSURE(ctx, ctx->rec, n);
    ENSURE(ctx, ctx->rec_q, n);
    ENSURE(ctx, ctx->rng, n * esz);
    ENSURE(ctx, ctx->keep, n);
    ENSURE(ctx, ctx->tier_list, n * (size_t)n_tiers);      // one list per later tier, each as long as the batch (address space) ...
    ENSURE(ctx, ctx->tier_sparse, n * (size_t)n_tiers);    // ... and the same as the scan leaves them: region by region
    ENSURE(ctx, ctx->ctile_cnt, (size_t)b.n_frames * (size_t)max_tiles + 1);
    ENSURE(ctx, ctx->ctile_base, (size_t)b.n_frames * (size_t)max_tiles + 1);
    SgBeamArgs a{};
    a.rows = b.rows; a.frame_off = b.frame_off; a.n_frames = b.n_frames; a.n_total = b.n_total; a.perm = perm;
    a.srows = ctx->srows.p; a.frame_unsorted = ctx->frame_unsorted.p;
    a.uniform_rows = (b.uniform_rows > 0 && b.n_total < ((int64_t)1 << 31)) ? b.uniform_rows : 0;
    a.inv_uniform_rows = a.uniform_rows > 0 ? 1.0f / (float)a.uniform_rows : 0.0f;
    a.las = R->d_las; a.frame_tables = ctx->frame_tables.p;
    a.rgrid = R->d_rgrid; a.beam_div_deg = b.beam_div_deg; a.rec = ctx->rec.p; a.rec_q = ctx->rec_q.p;
    a.status = b.status;
    a.rng = ctx->rng.p;
    a.dbg_count = b.dbg_count; a.dbg_rj = b.dbg_rj; a.dbg_ratio = b.dbg_ratio; a.dbg_cap = b.dbg_cap;
    a.exact_math = R->exact_math;
    a.per_lane_scan = R->per_lane_scan;
    // Later capacity tiers = classes of the tier lists; the last class is the global-list tier, whose lists hold a whole
    // table if need be (capped at 8192 flakes in one beam).
    const int n_cls = n_tiers;
    // Small batches (up to four sweeps): a tier holds a few hundred beams -- one or two waves' worth for one beam per lane, a chain of
    // dependent latencies 100 us long -- and the row kernels (snowgpu_rows.hip: G lanes per beam; scan, dict and received power in one
    // pass, no hand-over buffers) finish them in a third of that (0.334 -> 0.306 ms per single sweep); from 16 sweeps on they lose
    // (2-3x the instructions).  SNOWGPU_TIER_ROWS=1 / 0 forces either (tests/test_gpu_parity.py::test_tier_rows_switch).
    const bool rows_small = R->tier_rows_auto && b.n_total <= ((int64_t)1 << 19);
    const bool tier_rows = (R->tier_rows || rows_small) && R->tier_cap_override <= 0 && R->per_lane_scan >= 0;
    const int h_lanes = 256;
    const int h_cap = (int)std::min<uint32_t>(std::max<uint32_t>(R->max_flakes, 64u), 8192u);
    a.n_cls = n_cls;
    for (int k = 0; k + 1 < n_cls; ++k) a.cls_cap[k] = tiers[k + 1];
    a.cls_cap[n_cls - 1] = h_cap;
    ENSURE(ctx, ctx->h_lists, (size_t)4 * (size_t)(h_cap + 1) * (size_t)h_lanes);
    a.h_lists = ctx->h_lists.p; a.h_cap = h_cap; a.h_lanes = h_lanes;
    a.tier_list = ctx->tier_list.p; a.tier_stride = b.n_total; a.tier_sparse = ctx->tier_sparse.p;
    if (tier_rows) {
        ENSURE(ctx, ctx->redo_list, n * (size_t)n_tiers);
        a.redo_list = ctx->redo_list.p;
    }
    // Overflow slots: a beam of the pass over all rows that over-fills its LDS list, up to SG_OV_CAP flakes, leaves all of them in
    // the slot of its sorted position, and the tiers up to that capacity run no second scan (400 bytes per sorted position, touched
    // by the few per cent of beams that overflow: 13 GB of address space for a 256-sweep batch, 0.6 GB per chunk of the pipeline).
    const bool use_ov = tiers[0] < SG_OV_CAP && n_cls >= 2 && R->per_lane_scan >= 0 && !tier_rows &&
                        R->tier_cap_override <= 0 && n * SG_OV_STRIDE * sizeof(double) <= ((size_t)40 << 30);
    int64_t tq_caps[SG_MAX_CLASSES] = {0, 0, 0, 0};
    for (int k = 0; k + 1 < n_cls && !tier_rows; ++k) {
        if (use_ov && tiers[k + 1] <= SG_OV_CAP) continue;     // (this class reads the overflow slots: no hand-over buffer)
        tq_caps[k] = tier_queue_cap(R, tiers[k + 1], b.n_total);
        ENSURE(ctx, ctx->tq[k], ((size_t)tq_caps[k] + 64) * (3 * (size_t)tiers[k + 1] + 2));
        ENSURE(ctx, ctx->tq_sc[k], (size_t)tq_caps[k]);
    }
    // Regions of the first pass = slices of its dict queue: the segments, or plain chunks of 8 blocks in linear order.
    a.q_chunk = (int32_t)q_chunk;
    {
        const size_t planes = 3 * (size_t)tiers[0] + 2;        // range, azimuth, three values per flake
        if (n * planes * sizeof(double) > ((size_t)64 << 30))
            return fail(ctx, SNOWGPU_E_INVALID, "batch too large for the dict queue of this table density: split it");
        if (b.n_frames >= (1 << 22)) return fail(ctx, SNOWGPU_E_INVALID, "too many frames in one batch");
        ENSURE(ctx, ctx->dq, (n + 64) * planes);          // blocked SoA: groups of 64 slots
        ENSURE(ctx, ctx->dq_g, n);
        ENSURE(ctx, ctx->dq_sc, n);
        if (!seg_small) HIPCHK(ctx, hipMemsetAsync(ctx->qn.p, 0, sizeof(unsigned long long) * zero_words, st));     // (else k_seg_small cleared it)
        a.tn = (int32_t *)(ctx->qn.p + regions); a.tbase = ctx->tbase.p;
        a.diff2 = ctx->qn.p + 3 * regions;
        int32_t *small = (int32_t *)(ctx->qn.p + 3 * regions + (size_t)b.n_frames);      // 16 ints; behind them the compaction's per-frame tile counters
        a.tier_info = small; a.pw_count = small + 8; a.redo_cnt = small + 12;
        a.dq = ctx->dq.p; a.dq_g = ctx->dq_g.p; a.dq_sc = ctx->dq_sc.p; a.qn = ctx->qn.p; a.dq_n = b.n_total;
        const int lanes = first_block < 64 ? first_block : 64;
        a.n_regions_ub = use_seg ? (int64_t)b.n_frames * 256 : (b.n_total + a.q_chunk - 1) / a.q_chunk;
        a.blk_rows = first_block;
        a.kp_lds_quarters = 2;     // k_power's persistent blocks take half of each CU: the later tiers and the prepass run beside it (3 and 4 quarters, with the kernel at 168 VGPRs: C2 + 1 %, C2far + 8 %, C1 + 2 %)
        const size_t items_cap = n / (size_t)lanes + 2 * (size_t)a.n_regions_ub + 64;
        ENSURE(ctx, ctx->pw_items, 2 * items_cap);
        a.pw_items = ctx->pw_items.p;
        // beams with up to `few` flakes take their own kernel (k_power_few: registers only) -- unless the occlusion tap wants their dicts
        a.pw_items1 = (R->few > 0 && !b.dbg_count && lanes == 64) ? ctx->pw_items.p + items_cap : nullptr;
        a.front_max = a.pw_items1 ? std::min(R->few, std::min(3, tiers[0])) : 1;
    }
    if (use_ov) {
        ENSURE(ctx, ctx->ov, (n + 256) * SG_OV_STRIDE);
        ENSURE(ctx, ctx->ov_sc, n + 256);
        a.ov = ctx->ov.p; a.ov_sc = ctx->ov_sc.p; a.ov_cap = SG_OV_CAP;
    }
    if (use_seg) {
        a.seg_blk = ctx->seg_blk.p; a.seg_start = ctx->seg_start.p; a.seg_cnt = ctx->seg_cnt.p; a.seg_frame = ctx->seg_frame.p;
        a.seg_n = ctx->seg_n.p; a.seg_of_blk = ctx->seg_of_blk.p; a.chunk_blk = ctx->chunk_blk.p;
    }
    const bool few_first = a.pw_items1 && !serial && b.n_total > ((int64_t)1 << 19);
    // Large batches: ONE work queue and ONE persistent kernel (k_power_all) for everything k_power_few leaves -- the 16-entry class, the
    // 8-entry class (both from the overflow slots) and the multi-flake beams of the main queue, closed up over all regions -- instead of
    // k_power<4> / <8> / <16> side by side on three streams, each with a grid sized for a chip of its own.
    int cls8 = -1, cls16 = -1;
    for (int k = 0; k + 1 < n_cls; ++k) { if (tiers[k + 1] == 8) cls8 = k; if (tiers[k + 1] == 16) cls16 = k; }
    const bool kp_all = R->kp_all && few_first && use_ov && tiers[0] == 4 && !b.dbg_count && !tier_rows && b.n_total < ((int64_t)1 << 31);
    if (kp_all) {
        ENSURE(ctx, ctx->back_list, n + 64);
        ENSURE(ctx, ctx->bbase, regions);
        a.back_list = ctx->back_list.p; a.bbase = ctx->bbase.p;
    }
    // Where k_power<4> goes when the rare tiers are not rare.  The 63-entry and the global-list tier run behind k_power on its stream; with
    // 71 000 beams in them (C1: 40 k flakes per line) that chain -- 3.8 ms -- is the last thing to finish, and it only starts when k_power is
    // through.  The device leaves every batch's tier counts in page-locked memory (k_tier_gather); if the most recent ones that have landed
    // say that chain outlasts the 16-entry tier (per beam it costs ~22x as much: 53 - 129 ns against 2.4), k_power goes to the caller's
    // stream, ahead of the 8-entry tier, and the chain starts right behind k_power_few: C1 8.16 -> 7.93 ms.  Where the 16-entry tier is the
    // last to finish that order LOSES (C2far 8.49 -> 8.85 ms, C2 4.13 -> 4.23: the chain then takes CUs from the kernel everything waits for).
    // Whatever the words say, both orders give the same bytes (tests/test_gpu_fullsize.py).
    bool heavy_tail = false;
    if (few_first && n_cls >= 3 && ctx->tier_hint_d) {
        const volatile int32_t *hint = ctx->tier_hint_h;
        long tail = 0;
        for (int k = 2; k < n_cls && k < SG_MAX_CLASSES; ++k) tail += hint[k];
        heavy_tail = R->heavy_tail < 0 ? (tail >= 4096 && tail * 22 > (long)hint[1]) : R->heavy_tail == 1;
        a.tier_hint = ctx->tier_hint_d;
    }
    if (!seg_small) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    // measurement hooks: one event pair around the whole per-beam region
    const bool timed = ctx->prof && ctx->ev_used < (int)ctx->ev_start.size();
    if (timed) { HIPCHK(ctx, hipEventRecord(ctx->ev_start[(size_t)ctx->ev_used], st)); ctx->prof_stream = st; }
    int e = 0;
    {
        const int64_t lin_blocks = (b.n_total + first_block - 1) / first_block;
        if (use_seg) {
            a.chunk = 0;                                 // every non-empty (frame, channel) pair wastes less than one block
            a.grid_blocks = total_blocks_ub + b.max_frame / first_block + 2;
        } else {
            a.blk_lo = 0; a.blk_hi = lin_blocks;
            a.grid_blocks = lin_blocks;
        }
        e = sg_launch_beams(&a, b.dtype, tiers[0], 1, 1, st);
        // The plan of what the pass queued (work items of k_power_few / k_power; where each region's slice of the tier lists goes)
        // runs behind it on the same stream -- the tiers then start with one short kernel (k_tier_gather) and no hop between streams --
        // and the received-power kernels it feeds on a side stream, next to the later capacity tiers.
        if (!e) e = sg_launch_power(&a, b.dtype, tiers[0], st, 1, nullptr, 3);
        if (!e && kp_all) {
            // One work queue: the lists are closed up AHEAD of k_power_few (beside it, k_tier_gather's 4096 short blocks sat on the CUs when
            // k_power_few's persistent blocks arrived; those that found no room started when others had ended and walked their whole share
            // then: 0.78 instead of 0.54 ms), and k_power_few stays on the caller's stream -- it has the chip to itself anyway; what runs
            // beside k_power_all (the 63-entry / global-list chain, the prepass) forks behind it.
            e = sg_launch_tier_gather(&a, st);
            if (!e) e = sg_launch_power(&a, b.dtype, tiers[0], st, 0, ctx->ev_few, 1);
        } else if (!e) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_fp, st));
            HIPCHK(ctx, hipStreamWaitEvent(s_aux, ctx->ev_fp, 0));
            e = sg_launch_power(&a, b.dtype, tiers[0], s_aux, 0, few_first ? ctx->ev_few : nullptr, heavy_tail ? 1 : 3);
        }
        if (!e) {
            // Large batches: k_power_few has the chip to itself for its turn -- four waves per SIMD of it fill the register file, and
            // the tiers, the prepass and k_power do better behind it than beside it (measured: 4.37 against 4.69 ms per 256 sweeps when
            // they all start together; the other way round for a single sweep, where nothing fills anything).
            // k_tier_gather stays BEHIND this wait although it needs nothing of k_power_few: with it ahead the tiers and the prepass start
            // the moment k_power_few ends, together with k_power<4>, and take the CUs its persistent blocks would have taken -- 4.43 - 4.49
            // against 4.22 - 4.24 ms per step on one box (round 5); the 30 us it costs give k_power<4> its head start.
            // Long-tail batches (heavy_tail: the 63-entry chain outlasts everything) start the prepass beside k_power_few instead of behind
            // it: with the chain and three persistent kernels on the chip its small kernels wait for CUs (C1: k_pre_mean32 1.1 ms, k_lean_gather
            // 0.11 ms) and stand in the chain's way -- C1 7.39 -> 7.14 ms; where the tail is short the prepass is better off behind
            // k_power_few (C2 3.91 -> 3.95 the other way, C2far the same).  SNOWGPU_PREPASS_WITH_FEW=0 / 1 overrides.
            const bool pre_with_few = R->prepass_with_few < 0 ? heavy_tail : R->prepass_with_few == 1;
            if (pre_with_few && few_first && !kp_all && !b.thr_poly && !b.defer_thr && !pre_forked) { int prc = launch_prepass(); if (prc) return prc; }
            if (few_first && !kp_all) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_few, 0));
            if (!kp_all) e = sg_launch_tier_gather(&a, st);
        }
    }
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("beam launch: ") + hipGetErrorString((hipError_t)e));
    // The noise-threshold prepass streams the rows (bandwidth-bound, no LDS): it runs beside the received-power phase and
    // the later tiers (latency-bound, LDS-bound) rather than beside the sort and the scan, which it would slow down.
    if (!b.thr_poly && !b.defer_thr && !pre_forked) { int prc = launch_prepass(); if (prc) return prc; }
    // The scan put every over-full beam on the list of its tier (it counted on past a full list, so the beam knows which): the tiers
    // start as soon as it has ended, side by side -- class 0 on the caller's stream, class 1 on a side stream, and the classes from
    // the third on (the 63-entry and the global-list tier: few beams, long dependent chains) behind k_power on ITS stream, which is
    // free long before the 8- and 16-entry tiers are through (C2 4.41 -> 4.30 ms, C2far 9.20 -> 8.64 ms against "behind class 1").
    const bool side3 = n_cls >= 2;
    // (a small batch -- no k_power_few-first schedule -- is bound by the longest chain of dependent launches: there the rare tiers go behind
    // the 8-entry tier on the caller's stream, the shortest of the three chains in a single sweep's trace)
    const bool tail_main = n_cls >= 3 && !serial && !few_first;
    const bool tail_aux = n_cls >= 3 && !serial && !tail_main;
    if (side3) { HIPCHK(ctx, hipEventRecord(ctx->ev_lists, st)); HIPCHK(ctx, hipStreamWaitEvent(s_aux3, ctx->ev_lists, 0)); }
    if (tail_aux) HIPCHK(ctx, hipStreamWaitEvent(s_aux, ctx->ev_lists, 0));
    if (heavy_tail && !kp_all) {                         // (behind the event the other tiers' streams wait for: they start with it, not after it)
        e = sg_launch_power(&a, b.dtype, tiers[0], st, 0, nullptr, 2);
    }
    if (kp_all) {
        a.seg_blk = nullptr; a.tq = nullptr; a.tq_sc = nullptr; a.tq_cap = 0; a.tq_unsorted = 0;
        a.work_lo = 0; a.work_hi = (int32_t)std::min<int64_t>(b.n_total, INT32_MAX);
        e = sg_launch_power_all(&a, b.dtype, cls8, cls16, R->kp_all_waves, R->kp_all_ticket, st);
    }
    // what the classes held in a recent batch (page-locked words the device leaves behind, scaled to this batch's size): grids of the rare tiers
    int32_t cls_hint[SG_MAX_CLASSES] = {0, 0, 0, 0};
    if (a.tier_hint && ctx->tier_hint_h) {
        const volatile int32_t *hint = ctx->tier_hint_h;
        const int64_t then_k = hint[SG_MAX_CLASSES];
        if (then_k > 0)
            for (int k = 0; k < SG_MAX_CLASSES; ++k)
                cls_hint[k] = (int32_t)std::min<int64_t>(INT32_MAX / 8, ((int64_t)hint[k] * ((b.n_total >> 10) + 1)) / then_k + 64);
    }
    for (int k = 0; k < n_cls && !e; ++k) {
        if (kp_all && (k == cls8 || k == cls16)) continue;            // (taken by k_power_all)
        a.work_hint = k >= 2 ? cls_hint[k] : 0;
        hipStream_t sk = (k == 0 || (tail_main && k >= 2)) ? st : ((tail_aux && k >= 2) ? s_aux : s_aux3);
        a.seg_blk = nullptr;
        a.cls = k;
        if (k == n_cls - 1) {                            // the global-list tier
            e = sg_launch_huge(&a, b.dtype, sk);
            break;
        }
        const int lmax = tiers[k + 1];
        if (use_ov && lmax <= SG_OV_CAP) {               // its lists are in the overflow slots: received power only, no second scan
            a.tq = nullptr; a.tq_sc = nullptr; a.tq_cap = 0; a.ov_list = 1; a.tq_unsorted = 0;
            a.work_lo = 0; a.work_hi = (int32_t)std::min<int64_t>(b.n_total, INT32_MAX);
            e = sg_launch_power_list(&a, b.dtype, lmax, sk);
            a.ov_list = 0;
            continue;
        }
        if (tier_rows) {
            a.work_lo = 0; a.work_hi = (int32_t)std::min<int64_t>(b.n_total, INT32_MAX);
            e = sg_launch_rows(&a, b.dtype, lmax, sk);
            continue;
        }
        a.tq = ctx->tq[k].p; a.tq_sc = ctx->tq_sc[k].p; a.tq_cap = (int32_t)tq_caps[k];
        a.work_lo = 0; a.work_hi = (int32_t)tq_caps[k];
        a.tq_unsorted = 0;
        if (R->per_lane_scan < 0) e = sg_launch_beams(&a, b.dtype, lmax, 0, 1, sk);   // the wave scan in the tiers too: lists sorted in LDS (validation)
        else { a.tq_unsorted = 1; e = sg_launch_tier_scan(&a, b.dtype, lmax, sk); }   // one beam per lane, no LDS: k_power sorts as it loads
        if (!e) e = sg_launch_power_list(&a, b.dtype, lmax, sk);
        if (!e && tq_caps[k] < b.n_total) {              // entries beyond the hand-over buffer: received power in place
            a.work_lo = (int32_t)tq_caps[k]; a.work_hi = (int32_t)std::min<int64_t>(b.n_total, INT32_MAX);
            e = sg_launch_beams(&a, b.dtype, lmax, 0, 0, sk);
        }
    }
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("tier launch: ") + hipGetErrorString((hipError_t)e));
    if (side3) { HIPCHK(ctx, hipEventRecord(ctx->ev_join3, s_aux3)); HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join3, 0)); }
    HIPCHK(ctx, hipEventRecord(ctx->ev_join2, s_aux));
    HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join2, 0));
    if (timed) { HIPCHK(ctx, hipEventRecord(ctx->ev_stop[(size_t)ctx->ev_used], st)); ctx->ev_used++; }
    // 4. output rows from (sorted) rows + records, round, noise-floor filter, camera crop, compaction, stats
    // (simulation.py:516-540)
    if (pre_forked) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join0, 0));
    if (b.defer_thr && !b.thr_poly) return SNOWGPU_OK;           // the caller's polynomial is still being fitted: run_compaction finishes the batch
    return launch_compaction(ctx, b, perm, thr, regions, max_tiles);
}

// The batch's last step: output rows from (sorted) rows + records, np.round, noise-floor filter, camera crop, stable compaction,
// statistics (simulation.py:516-540).  Everything it reads was left by run_batch in the context's scratch.
static int launch_compaction(snowgpu_ctx *ctx, BatchDev &b, const int32_t *perm, const double *thr, size_t regions, int64_t max_tiles)
{
    snowgpu_ctx *R = ctx->root ? ctx->root : ctx;
    hipStream_t st = b.stream;
    if (b.pack) {
        ENSURE(ctx, ctx->pk_tile_mv, (size_t)b.n_frames * (size_t)max_tiles + 1);
        ENSURE(ctx, ctx->pk_tile_mv_base, (size_t)b.n_frames * (size_t)max_tiles + 1);
        b.pack->tile_mv = ctx->pk_tile_mv.p; b.pack->tile_mv_base = ctx->pk_tile_mv_base.p;
    }
    int e = sg_launch_compact(b.rows, ctx->srows.p, ctx->frame_unsorted.p, b.dtype, ctx->rec.p, ctx->rec_q.p, ctx->rng.p, thr, ctx->keep.p, perm, b.frame_off, b.n_frames, b.n_total,
                              ctx->ctile_cnt.p, ctx->ctile_base.p, b.out_rows, b.out_src, b.out_counts, b.out_stats,
                              ctx->qn.p + 3 * regions, b.no_fov ? nullptr : &R->fov, max_tiles, b.pack,
                              b.n_total <= ((int64_t)1 << 19) ? ctx->qn.p + 3 * regions + (size_t)b.n_frames + 8 : nullptr,      // (small batches: the scan inside the count kernel)
                              st);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("compaction launch: ") + hipGetErrorString((hipError_t)e));
    return SNOWGPU_OK;
}

// ... for a batch whose run_batch stopped ahead of it (defer_thr): b.thr_poly now holds the caller's polynomials (device memory).  The
// sizes run_batch derived are derived again -- from the same context state: tables, lasers and settings do not change inside a call.
static int run_compaction(snowgpu_ctx *ctx, BatchDev &b)
{
    snowgpu_ctx *R = ctx->root ? ctx->root : ctx;
    if (!b.thr_poly) return fail(ctx, SNOWGPU_E_INVALID, "run_compaction without threshold polynomials");
    if (b.n_total == 0) return SNOWGPU_OK;                        // (run_batch filled counts and statistics)
    const int64_t max_tiles = std::max<int64_t>(1, (b.max_frame + SG_TILE - 1) / SG_TILE);
    int tiers[4], n_tiers = 0;
    choose_tiers(R, b.beam_div_deg, tiers, &n_tiers);
    const size_t q_chunk = 8 * (size_t)sg_beams_block(tiers[0]);
    const size_t regions = std::max<size_t>((size_t)b.n_frames * 256, (size_t)b.n_total / q_chunk + 2);
    if (b.out_thr_poly)
        HIPCHK(ctx, hipMemcpyAsync(b.out_thr_poly, b.thr_poly, sizeof(double) * 3 * (size_t)b.n_frames, hipMemcpyDeviceToDevice, b.stream));
    return launch_compaction(ctx, b, b.perm ? b.perm : ctx->perm.p, b.thr_poly, regions, max_tiles);
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
        snprintf(buf, sizeof buf, "more than %d flakes intersect one beam (sorted row %d): beyond the global-list tier "
                 "(capacity = the largest uploaded table, at most SNOWGPU_MAX_FLAKES_GLOBAL)",
                 (int)std::min<uint32_t>(std::max<uint32_t>(ctx->max_flakes, 64u), 8192u), st[1]);
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

extern "C" int snowgpu_status_error(snowgpu_ctx *ctx, const int32_t *status8)
{
    if (!ctx || !status8) return SNOWGPU_E_INVALID;
    return status_to_error(ctx, status8);
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

// A host-pointer batch as a pipeline of chunks of whole frames.  What the traces of the first versions taught:
// the runtime keeps a pool of (by default four) hardware queues PER stream priority, and streams beyond that share a queue
// with another stream -- whose packets they then wait behind, events and copies included; its device-to-host copy is a
// full-grid blit kernel in a process that has initialised PyTorch and stalls every kernel beside it; a chunk's launch
// sequence is a third faster with its side streams than on one stream.  So:
//   * the upload of ALL chunks is one stream of DMA copies (high-priority pool) into a batch-sized buffer -- it never waits
//     for anything -- with one event per chunk;
//   * the chunks compute into a batch-sized result buffer, each as ONE chain of launches on one stream, alternating between
//     two lanes (the context itself and a sub-context with its own stream, events and scratch in the low-priority pool): a
//     chunk starts the moment its upload lands, and the launch latency of one chain hides behind the other;
//   * the downloads run on one more stream (low-priority pool: a hardware queue of its own), each after its chunk's event: the
//     runtime's copy, i.e. the DMA engine (a small-grid kernel of ours writing page-locked memory directly was measured slower:
//     every kernel boundary on the device then waits for the outstanding host writes).
// The host enqueues everything and waits once at the end.  Small per-frame arrays (table ids, planes / polynomials, counts,
// statistics) cross once for the whole batch; a chunk sees its slice of them.
static int host_batch_pipelined(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                                const int32_t *table_ids, double beam_div_deg, const double *thr_poly, const double *plane,
                                double noise_floor, const int32_t *perm, void *out_rows, int32_t *out_src, int64_t *out_counts,
                                int64_t *out_stats, double *out_thr_poly)
{
    const size_t esz = dtype == 0 ? 4 : 8, rb = 5 * esz, nf = (size_t)n_frames, nl = (size_t)ctx->h_las.n;
    const int64_t n_total = frame_offsets[n_frames];
    hipStream_t st = ctx->stream;
    // chunks of whole frames, about pipe_rows rows each
    std::vector<int> c_first;
    std::vector<int64_t> h_off;                  // chunk-local offsets: chunk c owns h_off[c_pos[c] .. c_pos[c] + frames + 1)
    std::vector<size_t> c_pos;
    // (the caller fits the threshold -- snowgpu_set_threshold_callback, below --: groups of 40 sweeps; the callback's cost per frame
    // falls with the group's size -- its selection runs on a thread pool, every call pays the pool's round trip -- and the calling thread
    // enqueues nothing while it is inside it: 256 sweeps, 24 / 32 / 40 / 48 / 56 sweeps per group: 1.36 / 1.32 / 1.42 / 1.44 / 1.40 G
    // points/s with the rows transfer, 1.39 / 1.42 / 1.57 / 1.63 / 1.66 with the packed one, scripts/probe/q8_group_probe.py)
    const int64_t pipe_rows = (ctx->thr_fn != nullptr && !thr_poly && !perm) ? std::max<int64_t>(ctx->pipe_rows, (int64_t)5 << 20) : ctx->pipe_rows;
    for (int f = 0; f < n_frames;) {
        int g = f;
        const int64_t base = frame_offsets[f];
        // (the last chunks are half size: what remains to be done after the last upload has landed -- the last chunk's kernels, its
        // download, the assembly of its rows -- is the part of the call nothing overlaps)
        const int64_t target = (n_total - base <= 2 * pipe_rows) ? std::max<int64_t>(pipe_rows / 2, 1) : pipe_rows;
        while (g < n_frames && (g == f || frame_offsets[g + 1] - base <= target)) ++g;
        c_first.push_back(f);
        c_pos.push_back(h_off.size());
        for (int k = f; k <= g; ++k) h_off.push_back(frame_offsets[k] - base);
        f = g;
    }
    c_first.push_back(n_frames);
    const int n_chunks = (int)c_first.size() - 1;
    const int L = std::max(1, std::min(ctx->pipe_lanes, n_chunks));
    {
        int prc = ensure_pipeline(ctx, n_chunks, L);
        if (prc) return prc;
    }
    static const bool trace = std::getenv("SNOWGPU_PIPE_TRACE") != nullptr;
    ENSURE(ctx, ctx->pipe_off, h_off.size());
    ENSURE(ctx, ctx->pipe_status, (size_t)n_chunks * 8);
    ENSURE(ctx, ctx->out_counts, nf);
    ENSURE(ctx, ctx->out_stats, nf * 3);
    ENSURE(ctx, ctx->table_ids, nf * nl);
    ENSURE(ctx, ctx->plane, nf * 4);
    ENSURE(ctx, ctx->rows_in, std::max<size_t>((size_t)n_total * rb, 8));
    const uint8_t *chn = (rows && dtype == 0) ? ctx->in_channels : nullptr;          // compact input (snowgpu_augment_batch_compact)
    if (chn) {
        ENSURE(ctx, ctx->rows_c4, std::max<size_t>((size_t)n_total * 16, 16));
        ENSURE(ctx, ctx->rows_ch, std::max<size_t>((size_t)n_total, 16));
    }
    // Packed result transfer: the compaction leaves, per kept row, its source row | label code and its intensity, and the moved
    // coordinates of the label-2 rows apart (SgPackOut); those cross the link in exact sizes once a chunk's counts have landed, and host
    // threads put the caller's rows together -- x, y, z (and the channel of rows without a laser) copied from the caller's INPUT rows.
    const bool packed = ctx->result_mode == 1 && rows != nullptr && n_total > 0;
    const size_t nt = (size_t)n_total;
    if (packed) {
        // the host threads read the caller's INPUT rows while they write out_rows: the two must not overlap (the rows transfer tolerates
        // rows == out_rows, this one would corrupt frames whose rows do not come channel-sorted); a word holds a 30-bit source row
        const char *r0 = (const char *)rows, *r1 = r0 + nt * (ctx->in_channels ? 4 : 5) * esz, *o0 = (const char *)out_rows, *o1 = o0 + nt * 5 * esz;
        if (r0 < o1 && o0 < r1) return fail(ctx, SNOWGPU_E_INVALID, "packed result transfer: out_rows overlaps rows (the rows are assembled from the input rows)");
        for (int f = 0; f < n_frames; ++f)
            if (frame_offsets[f + 1] - frame_offsets[f] >= ((int64_t)1 << 30))
                return fail(ctx, SNOWGPU_E_INVALID, "packed result transfer: a frame of 2^30 rows or more (30-bit source rows); use the rows transfer");
    }
    char *st_meta = nullptr, *st_int = nullptr, *st_mv = nullptr;
    int64_t *st_cnt = nullptr, *st_mvcnt = nullptr;
    if (packed) {
        ENSURE(ctx, ctx->pk_meta, nt);
        ENSURE(ctx, ctx->pk_int, nt * esz);
        ENSURE(ctx, ctx->pk_mv, nt * 3 * esz);
        ENSURE(ctx, ctx->pk_mvcnt, nf);
        const size_t o_int = (nt * 4 + 63) / 64 * 64, o_mv = o_int + (nt * esz + 63) / 64 * 64, o_cnt = o_mv + (nt * 3 * esz + 63) / 64 * 64;
        const size_t need = o_cnt + 16 * nf + 64;
        if (need > ctx->st_pk_cap) {
            if (ctx->st_pk) (void)hipHostFree(ctx->st_pk);
            ctx->st_pk = nullptr; ctx->st_pk_cap = 0;
            HIPCHK(ctx, hipHostMalloc((void **)&ctx->st_pk, need + need / 8, hipHostMallocDefault));
            ctx->st_pk_cap = need + need / 8;
        }
        st_meta = ctx->st_pk; st_int = ctx->st_pk + o_int; st_mv = ctx->st_pk + o_mv;
        st_cnt = (int64_t *)(ctx->st_pk + o_cnt); st_mvcnt = st_cnt + nf;
        while ((int)ctx->pk_ev.size() < 2 * n_chunks) {
            hipEvent_t e;
            HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->pk_ev.push_back(e);
        }
        if (!ctx->pool) {
            int n_thr = ctx->asm_threads;
            if (n_thr <= 0) {
                cpu_set_t cs;
                CPU_ZERO(&cs);
                int avail = (sched_getaffinity(0, sizeof cs, &cs) == 0) ? CPU_COUNT(&cs) : (int)std::thread::hardware_concurrency();
                if (FILE *fh = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {      // a container's CPU quota, if any
                    long long q = 0, per = 0;
                    if (std::fscanf(fh, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) avail = std::min<int>(avail, (int)((q + per - 1) / per));
                    std::fclose(fh);
                }
                n_thr = std::max(1, std::min(8, avail - 2));      // (eight copy at the pace of the link: measured 6 .. 14 threads, 2.25 - 2.31 G points/s)
            }
            ctx->pool = new AsmPool();
            ctx->pool->start(n_thr);
        }
        {   // the threads go where the rows they copy live (see node_of_address); the device's node if that cannot be told
            const int n_in = node_of_address(rows), n_out = node_of_address(out_rows);
            int nd = (n_out >= 0) ? n_out : n_in;
            if (nd < 0) nd = node_of_device(ctx->device);
            ctx->pool->set_node(nd);
        }
    } else {
        ENSURE(ctx, ctx->rows_out, std::max<size_t>((size_t)n_total * rb, 8));
        ENSURE(ctx, ctx->out_src, std::max<size_t>((size_t)n_total, 1));
    }
    // The small arrays lead the upload stream (chunk 0's event covers them).  On the compute stream they would leave it
    // "after a DMA copy" for the whole batch: 28 instead of 20 ms for 256 sweeps (measured).
    hipStream_t up = ctx->s_h2d;
    HIPCHK(ctx, hipMemcpyAsync(ctx->pipe_off.p, h_off.data(), sizeof(int64_t) * h_off.size(), hipMemcpyHostToDevice, up));
    HIPCHK(ctx, hipMemcpyAsync(ctx->table_ids.p, table_ids, sizeof(int32_t) * nf * nl, hipMemcpyHostToDevice, up));
    const double *d_thr = nullptr;
    if (thr_poly) {
        ENSURE(ctx, ctx->user_thr, nf * 3);
        HIPCHK(ctx, hipMemcpyAsync(ctx->user_thr.p, thr_poly, sizeof(double) * 3 * nf, hipMemcpyHostToDevice, up));
        d_thr = ctx->user_thr.p;
    } else if (plane) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->plane.p, plane, sizeof(double) * 4 * nf, hipMemcpyHostToDevice, up));
    }
    if (perm) {
        ENSURE(ctx, ctx->user_perm, (size_t)n_total);
        HIPCHK(ctx, hipMemcpyAsync(ctx->user_perm.p, perm, sizeof(int32_t) * (size_t)n_total, hipMemcpyHostToDevice, up));
    }
    if (out_thr_poly) ENSURE(ctx, ctx->out_thr, nf * 3);
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    std::vector<hipEvent_t> tev;                  // SNOWGPU_PIPE_TRACE: timed events -- base, then per chunk: uploaded, compute begins, computed, downloaded
    if (trace) {
        tev.resize(1 + 4 * (size_t)n_chunks);
        for (auto &e : tev) HIPCHK(ctx, hipEventCreate(&e));
        HIPCHK(ctx, hipStreamSynchronize(st));
        HIPCHK(ctx, hipEventRecord(tev[0], ctx->s_h2d));
    }
    // uploads: all of them, back to back (the scratch of an earlier batch on this context has been drained: every host entry
    // ends with a synchronisation)
    for (int c = 0; c < n_chunks; ++c) {
        const int64_t r0 = frame_offsets[c_first[(size_t)c]], cn = frame_offsets[c_first[(size_t)c + 1]] - r0;
        if (cn && rows && chn) {                       // compact input: 16 + 1 bytes per row up the link (k_expand_rows on the chunk's lane makes the rows)
            HIPCHK(ctx, hipMemcpyAsync(ctx->rows_c4.p + (size_t)r0 * 16, (const char *)rows + (size_t)r0 * 16, (size_t)cn * 16, hipMemcpyHostToDevice, ctx->s_h2d));
            HIPCHK(ctx, hipMemcpyAsync(ctx->rows_ch.p + (size_t)r0, chn + r0, (size_t)cn, hipMemcpyHostToDevice, ctx->s_h2d));
        } else if (cn && rows) HIPCHK(ctx, hipMemcpyAsync(ctx->rows_in.p + (size_t)r0 * rb, (const char *)rows + (size_t)r0 * rb, (size_t)cn * rb, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(ctx, hipEventRecord(ctx->pipe_ev[2 * (size_t)c], ctx->s_h2d));
        if (trace) HIPCHK(ctx, hipEventRecord(tev[1 + 4 * (size_t)c], ctx->s_h2d));
    }
    const double t_up = now();
    int rc = SNOWGPU_OK;
    // ---- packed result transfer: downloads sized by the counts, and the host threads that put the rows together ----------------------
    int pk_enq = 0, pk_asm = 0;                      // chunks whose compute and download are enqueued / whose rows are with the pool
    auto pk_mv_head = [](size_t chunk_rows) { return std::min(chunk_rows, std::max<size_t>(4096, chunk_rows / 8)); };
    auto assemble_frame = [=](int f, int64_t kept_dev, int64_t mv_at) {
        // out row j of frame f = the caller's input row src_j with the device's intensity and label; label-2 rows take their moved coordinates
        // (mv_at: where this frame's part of the batch's list of moved coordinates starts, in rows)
        const int64_t o = frame_offsets[f];
        const uint32_t n_rows = (uint32_t)(frame_offsets[f + 1] - o);          // (what came back from the device bounds no host loop or index unchecked)
        const int64_t kept = kept_dev > (int64_t)n_rows ? (int64_t)n_rows : kept_dev;
        const uint32_t *meta = (const uint32_t *)st_meta + o;
        if (esz == 4 && chn) {                         // compact input: x, y, z from the caller's (x, y, z, intensity) rows, the channel from its bytes
            const float *in = (const float *)rows + (size_t)o * 4, *it = (const float *)st_int + o, *mv = (const float *)st_mv + (size_t)mv_at * 3;
            const uint8_t *cb = chn + o;
            float *out = (float *)out_rows + (size_t)o * 5;
            for (int64_t j = 0; j < kept; ++j) {
                const uint32_t m = meta[j], code = m >> 30;
                uint32_t src = m & 0x3fffffffu;
                if (src >= n_rows) src = n_rows - 1;
                const float *ip = in + (size_t)src * 4;
                float *q = out + (size_t)j * 5;
                if (code == 2) { q[0] = mv[0]; q[1] = mv[1]; q[2] = mv[2]; mv += 3; }
                else { q[0] = ip[0]; q[1] = ip[1]; q[2] = ip[2]; }
                q[3] = it[j];
                q[4] = code == 3 ? (float)cb[src] : (float)code;
                if (out_src) out_src[o + j] = (int32_t)src;
            }
        } else if (esz == 4) {
            const float *in = (const float *)rows + (size_t)o * 5, *it = (const float *)st_int + o, *mv = (const float *)st_mv + (size_t)mv_at * 3;
            float *out = (float *)out_rows + (size_t)o * 5;
            for (int64_t j = 0; j < kept; ++j) {
                const uint32_t m = meta[j], code = m >> 30;
                uint32_t src = m & 0x3fffffffu;
                if (src >= n_rows) src = n_rows - 1;
                const float *ip = in + (size_t)src * 5;
                float *q = out + (size_t)j * 5;
                if (code == 2) { q[0] = mv[0]; q[1] = mv[1]; q[2] = mv[2]; mv += 3; }
                else { q[0] = ip[0]; q[1] = ip[1]; q[2] = ip[2]; }
                q[3] = it[j];
                q[4] = code == 3 ? ip[4] : (float)code;
                if (out_src) out_src[o + j] = (int32_t)src;
            }
        } else {
            const double *in = (const double *)rows + (size_t)o * 5, *it = (const double *)st_int + o, *mv = (const double *)st_mv + (size_t)mv_at * 3;
            double *out = (double *)out_rows + (size_t)o * 5;
            for (int64_t j = 0; j < kept; ++j) {
                const uint32_t m = meta[j], code = m >> 30;
                uint32_t src = m & 0x3fffffffu;
                if (src >= n_rows) src = n_rows - 1;
                const double *ip = in + (size_t)src * 5;
                double *q = out + (size_t)j * 5;
                if (code == 2) { q[0] = mv[0]; q[1] = mv[1]; q[2] = mv[2]; mv += 3; }
                else { q[0] = ip[0]; q[1] = ip[1]; q[2] = ip[2]; }
                q[3] = it[j];
                q[4] = code == 3 ? ip[4] : (double)code;
                if (out_src) out_src[o + j] = (int32_t)src;
            }
        }
    };
    // Enqueue what has become possible: the downloads of chunks whose counts have landed; the assembly of chunks whose downloads have.
    // wait = false: only what is ready now (called between the launches of later chunks); true: everything, blocking.
    auto pk_progress = [&](bool wait) -> hipError_t {
        while (pk_asm < pk_enq) {
            const int c = pk_asm;
            hipEvent_t ev = ctx->pk_ev[2 * (size_t)c];
            hipError_t q = wait ? hipEventSynchronize(ev) : hipEventQuery(ev);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); return hipSuccess; }     // ("not ready" must not be what the next launch check finds)
            if (q != hipSuccess) return q;
            // the chunk's words, intensities, the head of its moved-coordinates list and its counts are here
            const int f0 = c_first[(size_t)c], f1 = c_first[(size_t)c + 1];
            const size_t o = (size_t)frame_offsets[f0], head = pk_mv_head((size_t)(frame_offsets[f1] - frame_offsets[f0]));
            size_t n_mv = 0;
            for (int f = f0; f < f1; ++f) n_mv += (size_t)st_mvcnt[f];
            if (n_mv > head) {                        // a list longer than its head (more than one row in eight scattered): the rest now, waited for
                hipError_t e = hipMemcpyAsync(st_mv + (o + head) * 3 * esz, ctx->pk_mv.p + (o + head) * 3 * esz, (n_mv - head) * 3 * esz, hipMemcpyDeviceToHost, ctx->s_d2h);
                if (e == hipSuccess) e = hipEventRecord(ctx->pk_ev[2 * (size_t)c + 1], ctx->s_d2h);
                if (e == hipSuccess) e = hipEventSynchronize(ctx->pk_ev[2 * (size_t)c + 1]);
                if (e != hipSuccess) return e;
            }
            if (trace) (void)hipEventRecord(tev[4 + 4 * (size_t)c], ctx->s_d2h);
            int64_t mv_at = frame_offsets[f0];
            for (int f = f0; f < f1; ++f) {
                const int64_t kept = st_cnt[f];
                if (kept > 0) ctx->pool->push([=]() { assemble_frame(f, kept, mv_at); });
                mv_at += st_mvcnt[f];
            }
            ++pk_asm;
        }
        return hipSuccess;
    };
    // A failure inside the loop must not return while copies from / into the caller's buffers (and from h_off) are in flight:
    // every exit goes through the drain below.
#define PIPECHK(call)                                                                                              \
    { hipError_t e__ = (call); if (e__ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e__); return (int)SNOWGPU_E_HIP; } }
    // The caller fits the noise threshold (snowgpu_set_threshold_callback): per chunk the device half of the prepass leads the chunk's
    // kernels, its histograms come down while they run, and the chunk is FINISHED -- callback, polynomials up, compaction, downloads --
    // when its lane is needed again (L chunks later) or at the end; the host's selection of chunk c thus runs beside the kernels of
    // chunks c + 1 .. c + L - 1 and beside the link's traffic.
    const bool cb = ctx->thr_fn != nullptr && !thr_poly && !perm && n_total > 0;
    constexpr size_t HIST = (size_t)50 * 2555;
    int32_t *sg_hist = nullptr, *sg_stat = nullptr;
    double *sg_rec = nullptr, *sg_thr = nullptr;
    if (cb) {
        const size_t o_rec = nf * HIST * 4, o_thr = o_rec + nf * SG_PRE_REC * 8, o_stat = o_thr + nf * 3 * 8, need = o_stat + (size_t)n_chunks * 32 + 64;
        if (need > ctx->thr_stage_cap) {
            if (ctx->thr_stage) (void)hipHostFree(ctx->thr_stage);
            ctx->thr_stage = nullptr; ctx->thr_stage_cap = 0;
            HIPCHK(ctx, hipHostMalloc((void **)&ctx->thr_stage, need + need / 8, hipHostMallocDefault));
            ctx->thr_stage_cap = need + need / 8;
        }
        sg_hist = (int32_t *)ctx->thr_stage; sg_rec = (double *)(ctx->thr_stage + o_rec); sg_thr = (double *)(ctx->thr_stage + o_thr);
        sg_stat = (int32_t *)(ctx->thr_stage + o_stat);
        while ((int)ctx->thr_ev.size() < 2 * n_chunks) {
            hipEvent_t e;
            HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->thr_ev.push_back(e);
        }
        ENSURE(ctx, ctx->user_thr, nf * 3);
    }
    struct Chunk { BatchDev b; SgPackOut po; snowgpu_ctx *lc; };
    std::vector<Chunk> chunks((size_t)n_chunks);
    // chunk c: its kernels (all of them, or everything ahead of the compaction when the caller fits the threshold)
    auto compute = [&](int c) -> int {
        const int f0 = c_first[(size_t)c], f1 = c_first[(size_t)c + 1], cf = f1 - f0;
        const int64_t r0 = frame_offsets[f0], cn = frame_offsets[f1] - r0;
        const int64_t *lo = &h_off[c_pos[(size_t)c]];
        Chunk &k = chunks[(size_t)c];
        snowgpu_ctx *lc = (c % L) == 0 ? ctx : ctx->lanes[(size_t)(c % L) - 1];     // chunk c computes on lane c mod L
        k.lc = lc;
        hipStream_t cs = lc->stream;
        PIPECHK(hipStreamWaitEvent(cs, ctx->pipe_ev[2 * (size_t)c], 0));
        if (trace) PIPECHK(hipEventRecord(tev[2 + 4 * (size_t)c], cs));
        if (chn && cn) {
            int xe = sg_launch_expand_rows(ctx->rows_c4.p + (size_t)r0 * 16, ctx->rows_ch.p + (size_t)r0, ctx->rows_in.p + (size_t)r0 * rb, cn, cs);
            if (xe) return fail(ctx, SNOWGPU_E_HIP, std::string("expand launch: ") + hipGetErrorString((hipError_t)xe));
        }
        BatchDev &b = k.b;
        b = BatchDev{};
        b.n_frames = cf; b.n_total = cn; b.frame_off = ctx->pipe_off.p + c_pos[(size_t)c]; b.rows = ctx->rows_in.p + (size_t)r0 * rb;
        int64_t mx = 0;
        for (int q = 0; q < cf; ++q) mx = std::max(mx, lo[q + 1] - lo[q]);
        bool uni = mx > 0;
        for (int q = 0; q < cf && uni; ++q) uni = (lo[q + 1] - lo[q]) == mx;
        b.max_frame = mx; b.uniform_rows = uni ? mx : 0;
        b.dtype = dtype; b.table_ids = ctx->table_ids.p + (size_t)f0 * nl; b.beam_div_deg = beam_div_deg;
        b.thr_poly = d_thr ? d_thr + 3 * (size_t)f0 : nullptr;
        b.plane = (!d_thr && plane) ? ctx->plane.p + 4 * (size_t)f0 : nullptr;
        b.noise_floor = noise_floor; b.perm = perm ? ctx->user_perm.p + r0 : nullptr;
        k.po = SgPackOut{};
        if (packed) {
            k.po.meta = ctx->pk_meta.p + r0; k.po.inten = ctx->pk_int.p + (size_t)r0 * esz; k.po.mv = ctx->pk_mv.p + (size_t)r0 * 3 * esz;
            k.po.mv_counts = ctx->pk_mvcnt.p + f0;
            b.pack = &k.po;
        } else {
            b.out_rows = ctx->rows_out.p + (size_t)r0 * rb; b.out_src = ctx->out_src.p + r0;
        }
        b.out_counts = ctx->out_counts.p + f0; b.out_stats = ctx->out_stats.p + 3 * (size_t)f0;
        b.out_thr_poly = out_thr_poly ? ctx->out_thr.p + 3 * (size_t)f0 : nullptr;
        b.status = ctx->pipe_status.p + 8 * (size_t)c; b.stream = cs;
        // Beside a saturated link every cross-stream event costs more (the queues' completion signals live in host memory), so a
        // chunk keeps its kernels on one stream: 1.80 instead of 1.76 G points/s (2.09 / 1.96 without source indices), although
        // the same chunk alone is faster with its side streams.
        b.serial = true;
        if (cb && cn > 0) {
            // the device half of the prepass first (snowgpu_prepass_stats' kernels on the chunk), its results on their way down at once
            snowgpu_ctx *R = ctx;
            const double *pl = b.plane;
            if (!pl) {
                if (lc->plane_est.ensure((size_t)cf * 4) || lc->plane_info.ensure((size_t)cf * 4)) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed for the plane estimate");
                int pe = sg_plane_run(&lc->plane_scr, &R->plane_par, b.rows, dtype, b.frame_off, nullptr, cf, cn, mx, lc->plane_est.p, lc->plane_info.p, cs);
                if (pe) return fail(ctx, SNOWGPU_E_HIP, std::string("plane estimate: ") + (pe > 0 ? hipGetErrorString((hipError_t)pe) : "allocation"));
                pl = lc->plane_est.p;
            }
            if (lc->stats_hist.ensure((size_t)cf * HIST) || lc->stats_rec.ensure((size_t)cf * SG_PRE_REC) || (!lc->d_status && hipMalloc((void **)&lc->d_status, 32) != hipSuccess))
                return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed for the prepass statistics");
            PIPECHK(hipMemsetAsync(lc->d_status, 0, 32, cs));
            int se = sg_prepass_stats_run(&lc->prepass, b.rows, dtype, b.frame_off, cf, cn, mx, pl, lc->stats_hist.p, lc->stats_rec.p, lc->d_status, cs);
            if (se) return fail(ctx, SNOWGPU_E_HIP, std::string("prepass: ") + (se > 0 ? hipGetErrorString((hipError_t)se) : "allocation"));
            PIPECHK(hipEventRecord(ctx->thr_ev[2 * (size_t)c], cs));
            PIPECHK(hipStreamWaitEvent(ctx->s_d2h, ctx->thr_ev[2 * (size_t)c], 0));
            PIPECHK(hipMemcpyAsync(sg_hist + (size_t)f0 * HIST, lc->stats_hist.p, (size_t)cf * HIST * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
            PIPECHK(hipMemcpyAsync(sg_rec + (size_t)f0 * SG_PRE_REC, lc->stats_rec.p, (size_t)cf * SG_PRE_REC * 8, hipMemcpyDeviceToHost, ctx->s_d2h));
            PIPECHK(hipMemcpyAsync(sg_stat + 8 * (size_t)c, lc->d_status, 32, hipMemcpyDeviceToHost, ctx->s_d2h));
            PIPECHK(hipEventRecord(ctx->thr_ev[2 * (size_t)c + 1], ctx->s_d2h));
            b.defer_thr = true;
        }
        int brc = run_batch(lc, b);
        if (brc != SNOWGPU_OK && lc != ctx) ctx->err = lc->err;
        return brc;
    };
    // chunk c: (the caller's threshold fit and the compaction, then) its downloads
    auto finish = [&](int c) -> int {
        const int f0 = c_first[(size_t)c], f1 = c_first[(size_t)c + 1], cf = f1 - f0;
        const int64_t r0 = frame_offsets[f0], cn = frame_offsets[f1] - r0;
        Chunk &k = chunks[(size_t)c];
        BatchDev &b = k.b;
        hipStream_t cs = k.lc->stream;
        if (b.defer_thr) {
            PIPECHK(hipEventSynchronize(ctx->thr_ev[2 * (size_t)c + 1]));
            const int32_t *s8 = sg_stat + 8 * (size_t)c;
            if (s8[0] != 0) {                                  // (fewer than 3 ground rows in a frame: reported as the device prepass reports it)
                PIPECHK(hipMemcpyAsync(b.status, k.lc->d_status, 32, hipMemcpyDeviceToDevice, cs));
                return SNOWGPU_OK;                             // the chunk's status words carry the error to the end of the call
            }
            const int crc = ctx->thr_fn(ctx->thr_user, f0, cf, sg_hist + (size_t)f0 * HIST, sg_rec + (size_t)f0 * SG_PRE_REC, sg_thr + 3 * (size_t)f0);
            if (crc != 0) return fail(ctx, SNOWGPU_E_INVALID, "the threshold callback reported an error");
            PIPECHK(hipMemcpyAsync(ctx->user_thr.p + 3 * (size_t)f0, sg_thr + 3 * (size_t)f0, sizeof(double) * 3 * (size_t)cf, hipMemcpyHostToDevice, cs));
            b.thr_poly = ctx->user_thr.p + 3 * (size_t)f0;
            int crc2 = run_compaction(k.lc, b);
            if (crc2 != SNOWGPU_OK) { if (k.lc != ctx) ctx->err = k.lc->err; return crc2; }
        }
        if (packed) {
            // The chunk's words and intensities come down as two copies of its whole row range (the rows of a frame are compacted at the
            // frame's offset: what lies behind a frame's kept rows travels unused -- a copy per frame instead cost ~20 us each, 17 ms per
            // batch), then the head of its list of moved coordinates -- room for one row in eight: the list's length is only known on the
            // device, and a copy sized by it would have to queue behind the copies of every later chunk --, then its counts.  A chunk with
            // more scattered rows than that gets the rest of its list by one more copy (pk_progress).
            SgPackOut &po = k.po;
            PIPECHK(hipEventRecord(ctx->pipe_ev[2 * (size_t)c + 1], cs));
            if (trace) PIPECHK(hipEventRecord(tev[3 + 4 * (size_t)c], cs));
            PIPECHK(hipStreamWaitEvent(ctx->s_d2h, ctx->pipe_ev[2 * (size_t)c + 1], 0));
            if (cn) {
                PIPECHK(hipMemcpyAsync(st_meta + (size_t)r0 * 4, po.meta, (size_t)cn * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
                PIPECHK(hipMemcpyAsync(st_int + (size_t)r0 * esz, po.inten, (size_t)cn * esz, hipMemcpyDeviceToHost, ctx->s_d2h));
                PIPECHK(hipMemcpyAsync(st_mv + (size_t)r0 * 3 * esz, po.mv, pk_mv_head((size_t)cn) * 3 * esz, hipMemcpyDeviceToHost, ctx->s_d2h));
            }
            PIPECHK(hipMemcpyAsync(st_cnt + f0, b.out_counts, sizeof(int64_t) * (size_t)cf, hipMemcpyDeviceToHost, ctx->s_d2h));
            PIPECHK(hipMemcpyAsync(st_mvcnt + f0, po.mv_counts, sizeof(int64_t) * (size_t)cf, hipMemcpyDeviceToHost, ctx->s_d2h));
            PIPECHK(hipEventRecord(ctx->pk_ev[2 * (size_t)c], ctx->s_d2h));
            pk_enq = c + 1;
            if (hipError_t pe = pk_progress(false); pe != hipSuccess) return fail(ctx, SNOWGPU_E_HIP, std::string("packed download: ") + hipGetErrorString(pe));
            return SNOWGPU_OK;
        }
        PIPECHK(hipEventRecord(ctx->pipe_ev[2 * (size_t)c + 1], cs));
        if (trace) PIPECHK(hipEventRecord(tev[3 + 4 * (size_t)c], cs));
        PIPECHK(hipStreamWaitEvent(ctx->s_d2h, ctx->pipe_ev[2 * (size_t)c + 1], 0));
        if (cn) {
            PIPECHK(hipMemcpyAsync((char *)out_rows + (size_t)r0 * rb, b.out_rows, (size_t)cn * rb, hipMemcpyDeviceToHost, ctx->s_d2h));
            if (out_src) PIPECHK(hipMemcpyAsync(out_src + r0, b.out_src, sizeof(int32_t) * (size_t)cn, hipMemcpyDeviceToHost, ctx->s_d2h));
        }
        if (trace) PIPECHK(hipEventRecord(tev[4 + 4 * (size_t)c], ctx->s_d2h));
        return SNOWGPU_OK;
    };
    for (int c = 0; c < n_chunks && rc == SNOWGPU_OK; ++c) {
        if (cb && c >= L) rc = finish(c - L);                  // (frees the lane chunk c computes on)
        if (rc == SNOWGPU_OK) rc = compute(c);
        if (rc == SNOWGPU_OK && !cb) rc = finish(c);
    }
    for (int c = std::max(0, n_chunks - L); cb && c < n_chunks && rc == SNOWGPU_OK; ++c) rc = finish(c);
#undef PIPECHK
    if (packed) {
        ctx->pk_times[0] = now() - t_begin;
        if (rc == SNOWGPU_OK) { if (hipError_t pe = pk_progress(true); pe != hipSuccess) rc = fail(ctx, SNOWGPU_E_HIP, std::string("packed download (drain): ") + hipGetErrorString(pe)); }
        ctx->pk_times[1] = now() - t_begin;
        ctx->pool->wait_idle();                       // (also on errors: no thread may still touch the caller's buffers when this returns)
        ctx->pk_times[2] = now() - t_begin;
    }
    if (trace) fprintf(stderr, "pipe: %d chunks; uploads enqueued in %.3f ms, everything in %.3f ms\n", n_chunks, t_up - t_begin, now() - t_begin);
    hipError_t se = hipStreamSynchronize(ctx->s_h2d);
    for (int l = 1; l < L; ++l) {
        hipError_t e = hipStreamSynchronize(ctx->lanes[(size_t)l - 1]->stream);
        if (se == hipSuccess) se = e;
    }
    for (hipStream_t w : {st, ctx->s_d2h}) {
        hipError_t e = hipStreamSynchronize(w);
        if (se == hipSuccess) se = e;
    }
    if (trace) {
        fprintf(stderr, "pipe: drained at %.3f ms\n", now() - t_begin);
        for (int c = 0; c < n_chunks; ++c) {
            float t[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4; ++k) (void)hipEventElapsedTime(&t[k], tev[0], tev[1 + 4 * (size_t)c + k]);
            fprintf(stderr, "pipe chunk %2d: uploaded %7.3f  compute %7.3f .. %7.3f  downloaded %7.3f ms\n", c, t[0], t[1], t[2], t[3]);
        }
        for (auto &e : tev) (void)hipEventDestroy(e);
    }
    if (rc != SNOWGPU_OK) return rc;
    if (se != hipSuccess) return fail(ctx, SNOWGPU_E_HIP, std::string("stream synchronize: ") + hipGetErrorString(se));
    std::vector<int32_t> h_st((size_t)n_chunks * 8, 0);
    HIPCHK(ctx, hipMemcpy(h_st.data(), ctx->pipe_status.p, sizeof(int32_t) * h_st.size(), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out_counts, ctx->out_counts.p, sizeof(int64_t) * nf, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out_stats, ctx->out_stats.p, sizeof(int64_t) * 3 * nf, hipMemcpyDeviceToHost));
    if (out_thr_poly) HIPCHK(ctx, hipMemcpy(out_thr_poly, ctx->out_thr.p, sizeof(double) * 3 * nf, hipMemcpyDeviceToHost));
    int32_t agg[8] = {0, -1, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < n_chunks; ++c) {
        const int32_t *s8 = &h_st[(size_t)c * 8];
        if (agg[0] == 0 && s8[0] != 0) {           // the offending row is chunk-local on the device: report it as a row of the batch
            agg[0] = s8[0];
            agg[1] = s8[1] >= 0 ? (int32_t)std::min<int64_t>(s8[1] + frame_offsets[c_first[(size_t)c]], INT32_MAX) : -1;
        }
        for (int k = 2; k < 6; ++k) agg[k] += s8[k];
    }
    std::memcpy(ctx->h_status, agg, sizeof agg);
    return status_to_error(ctx, agg);
}

static int ensure_mail(snowgpu_ctx *ctx, size_t up, size_t dn)
{
    if (up > ctx->mail_up_cap) {
        if (ctx->mail_up_h) (void)hipHostFree(ctx->mail_up_h);
        ctx->mail_up_h = nullptr; ctx->mail_up_cap = 0;
        const size_t want = up + up / 2 + 4096;
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->mail_up_h, want, hipHostMallocDefault));
        ctx->mail_up_cap = want;
    }
    if (dn > ctx->mail_dn_cap) {
        if (ctx->mail_dn_h) (void)hipHostFree(ctx->mail_dn_h);
        ctx->mail_dn_h = nullptr; ctx->mail_dn_cap = 0;
        const size_t want = dn + dn / 2 + 4096;
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->mail_dn_h, want, hipHostMallocDefault));
        ctx->mail_dn_cap = want;
    }
    ENSURE(ctx, ctx->mail_up_d, ctx->mail_up_cap);
    ENSURE(ctx, ctx->mail_dn_d, ctx->mail_dn_cap);
    return SNOWGPU_OK;
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
    if (n_total > 0 && !out_rows) return fail(ctx, SNOWGPU_E_INVALID, "null row buffers");
    if (n_total > 0 && !rows && (ctx->resident_rows != n_total || ctx->resident_dtype != dtype || (int)ctx->resident_off.size() != n_frames + 1 ||
                                 !std::equal(ctx->resident_off.begin(), ctx->resident_off.end(), frame_offsets)))
        return fail(ctx, SNOWGPU_E_INVALID, "rows == NULL needs the rows of the last snowgpu_prepass_stats call (same frame offsets and dtype)");
    if (rows) ctx->resident_rows = -1;                 // (a fresh upload replaces whatever was resident)
    if (!out_counts || !out_stats) return fail(ctx, SNOWGPU_E_INVALID, "null count/stat buffers");
    if (ctx->h_las.n <= 0) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_lasers has not been called");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool wants_precrop = ctx->fov.enabled && ctx->fov_pre && !dbg_count && n_total > 0;
    if (wants_precrop && !rows) return fail(ctx, SNOWGPU_E_INVALID, "rows == NULL cannot be combined with the pre-augment crop");
    if (!dbg_count && !perm_out && !wants_precrop && ctx->pipe_rows > 0 && n_frames > 1 && n_total > ctx->pipe_rows + ctx->pipe_rows / 2)
        return host_batch_pipelined(ctx, n_frames, frame_offsets, rows, dtype, table_ids, beam_div_deg, thr_poly, plane, noise_floor, perm,
                                    out_rows, out_src, out_counts, out_stats, out_thr_poly);
    const size_t esz = dtype == 0 ? 4 : 8, n = (size_t)n_total;
    const size_t row_bytes = n * 5 * esz;
    hipStream_t st = ctx->stream;
    ENSURE(ctx, ctx->rows_in, std::max<size_t>(row_bytes, 8));
    ENSURE(ctx, ctx->rows_out, std::max<size_t>(row_bytes, 8));
    ENSURE(ctx, ctx->out_src, std::max<size_t>(n, 1));
    ENSURE(ctx, ctx->thr_poly, (size_t)n_frames * 3);
    // the small arrays: one block up (offsets | polynomials or planes | table ids), one block down (status | counts | stats | polynomials)
    const size_t nfz = (size_t)n_frames, nlz = (size_t)ctx->h_las.n;
    const size_t up_off = 0, up_par = up_off + 8 * (nfz + 1), up_ids = up_par + 8 * 4 * nfz, up_bytes = up_ids + 4 * nfz * nlz;
    const size_t dn_st = 0, dn_cnt = 32, dn_stats = dn_cnt + 8 * nfz, dn_thr = dn_stats + 24 * nfz, dn_bytes = dn_thr + 24 * nfz;
    {
        int mrc = ensure_mail(ctx, up_bytes, dn_bytes);
        if (mrc) return mrc;
    }
    std::memcpy(ctx->mail_up_h + up_off, frame_offsets, 8 * (nfz + 1));
    if (thr_poly) std::memcpy(ctx->mail_up_h + up_par, thr_poly, 24 * nfz);
    else if (plane) std::memcpy(ctx->mail_up_h + up_par, plane, 32 * nfz);          // neither: the plane is estimated on the device
    std::memcpy(ctx->mail_up_h + up_ids, table_ids, 4 * nfz * nlz);
    HIPCHK(ctx, hipMemcpyAsync(ctx->mail_up_d.p, ctx->mail_up_h, up_bytes, hipMemcpyHostToDevice, st));
    if (row_bytes && rows && ctx->in_channels) {       // compact input (snowgpu_augment_batch_compact)
        ENSURE(ctx, ctx->rows_c4, n * 16);
        ENSURE(ctx, ctx->rows_ch, n);
        HIPCHK(ctx, hipMemcpyAsync(ctx->rows_c4.p, rows, n * 16, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->rows_ch.p, ctx->in_channels, n, hipMemcpyHostToDevice, st));
        int xe = sg_launch_expand_rows(ctx->rows_c4.p, ctx->rows_ch.p, ctx->rows_in.p, n_total, st);
        if (xe) return fail(ctx, SNOWGPU_E_HIP, std::string("expand launch: ") + hipGetErrorString((hipError_t)xe));
    } else if (row_bytes && rows) HIPCHK(ctx, hipMemcpyAsync(ctx->rows_in.p, rows, row_bytes, hipMemcpyHostToDevice, st));
    const int64_t *d_frame_off = (const int64_t *)(ctx->mail_up_d.p + up_off);
    const int32_t *d_table_ids = (const int32_t *)(ctx->mail_up_d.p + up_ids);
    const double *d_thr = thr_poly ? (const double *)(ctx->mail_up_d.p + up_par) : nullptr;
    const double *d_plane = (thr_poly || !plane) ? nullptr : (const double *)(ctx->mail_up_d.p + up_par);
    int32_t *d_status = (int32_t *)(ctx->mail_dn_d.p + dn_st);
    int64_t *d_counts = (int64_t *)(ctx->mail_dn_d.p + dn_cnt), *d_stats = (int64_t *)(ctx->mail_dn_d.p + dn_stats);
    double *d_thr_out = (double *)(ctx->mail_dn_d.p + dn_thr);
    DevBuf<int32_t> &user_perm = ctx->user_perm;
    if (perm) {
        if (user_perm.ensure(std::max<size_t>(n, 1))) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed for perm");
        if (n) HIPCHK(ctx, hipMemcpyAsync(user_perm.p, perm, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    }
    // Pre-augment camera crop (precompute.py:96-99): the frames are compacted on the device before anything else sees
    // them; only the per-frame counts visit the host (the frame offsets of the cropped batch are made there).
    const bool precrop = ctx->fov.enabled && ctx->fov_pre && !dbg_count && n > 0;
    std::vector<int64_t> crop_off;
    const void *d_rows_used = ctx->rows_in.p;
    const int64_t *d_off_used = d_frame_off;
    int64_t n_used = n_total, max_frame_used = max_frame;
    if (precrop) {
        if (perm) return fail(ctx, SNOWGPU_E_INVALID, "a caller-supplied permutation cannot be combined with the pre-augment crop");
        const int64_t max_tiles = std::max<int64_t>(1, (max_frame + SG_TILE - 1) / SG_TILE);
        ENSURE(ctx, ctx->keep, n);
        ENSURE(ctx, ctx->ctile_cnt, (size_t)n_frames * (size_t)max_tiles + 1);
        ENSURE(ctx, ctx->ctile_base, (size_t)n_frames * (size_t)max_tiles + 1);
        ENSURE(ctx, ctx->crop_counts, (size_t)n_frames);
        ENSURE(ctx, ctx->crop_stats, (size_t)n_frames * 3);
        ENSURE(ctx, ctx->crop_off, (size_t)n_frames + 1);
        ENSURE(ctx, ctx->rows_crop, row_bytes);
        ENSURE(ctx, ctx->crop_src, n);
        ENSURE(ctx, ctx->crop_out_src, n);
        int e = sg_launch_crop_count(ctx->rows_in.p, dtype, d_frame_off, n_frames, ctx->keep.p, ctx->ctile_cnt.p, ctx->ctile_base.p,
                                     ctx->crop_counts.p, ctx->crop_stats.p, &ctx->fov, max_tiles, st);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("crop launch: ") + hipGetErrorString((hipError_t)e));
        std::vector<int64_t> cnt((size_t)n_frames);
        HIPCHK(ctx, hipMemcpyAsync(cnt.data(), ctx->crop_counts.p, sizeof(int64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        crop_off.assign((size_t)n_frames + 1, 0);
        max_frame_used = 0;
        for (int f = 0; f < n_frames; ++f) {
            crop_off[(size_t)f + 1] = crop_off[(size_t)f] + cnt[(size_t)f];
            max_frame_used = std::max(max_frame_used, cnt[(size_t)f]);
        }
        n_used = crop_off[(size_t)n_frames];
        HIPCHK(ctx, hipMemcpyAsync(ctx->crop_off.p, crop_off.data(), sizeof(int64_t) * ((size_t)n_frames + 1), hipMemcpyHostToDevice, st));
        e = sg_launch_crop_scatter(ctx->rows_in.p, dtype, ctx->keep.p, d_frame_off, ctx->crop_off.p, n_frames, ctx->ctile_base.p,
                                   ctx->rows_crop.p, ctx->crop_src.p, max_tiles, st);
        if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("crop launch: ") + hipGetErrorString((hipError_t)e));
        d_rows_used = ctx->rows_crop.p; d_off_used = ctx->crop_off.p;
    }
    BatchDev b{};
    b.n_frames = n_frames; b.n_total = n_used; b.max_frame = max_frame_used; b.frame_off = d_off_used; b.rows = d_rows_used;
    {
        bool uni = max_frame_used > 0;
        const int64_t *ho = precrop ? crop_off.data() : frame_offsets;
        for (int f = 0; f < n_frames && uni; ++f) uni = (ho[f + 1] - ho[f]) == max_frame_used;
        b.uniform_rows = uni ? max_frame_used : 0;
    }
    b.dtype = dtype; b.table_ids = d_table_ids; b.beam_div_deg = beam_div_deg; b.thr_poly = d_thr;
    b.plane = d_plane; b.noise_floor = noise_floor; b.perm = perm ? user_perm.p : nullptr;
    b.out_rows = ctx->rows_out.p; b.out_src = ctx->out_src.p; b.out_counts = d_counts; b.out_stats = d_stats;
    b.out_thr_poly = out_thr_poly ? d_thr_out : nullptr; b.status = d_status; b.stream = st;
    b.no_fov = dbg_count != nullptr;
    b.want_perm = perm_out != nullptr;
    if (dbg_count) {
        ENSURE(ctx, ctx->dbg_count, std::max<size_t>(n, 1));
        ENSURE(ctx, ctx->dbg_rj, std::max<size_t>(n * (size_t)dbg_cap, 1));
        ENSURE(ctx, ctx->dbg_ratio, std::max<size_t>(n * (size_t)dbg_cap, 1));
        HIPCHK(ctx, hipMemsetAsync(ctx->dbg_count.p, 0, sizeof(int32_t) * std::max<size_t>(n, 1), st));
        b.dbg_count = ctx->dbg_count.p; b.dbg_rj = ctx->dbg_rj.p; b.dbg_ratio = ctx->dbg_ratio.p; b.dbg_cap = dbg_cap;
    }
    int rc = SNOWGPU_OK;
    if (ctx->thr_fn && !thr_poly && !perm && !dbg_count && !precrop && n_used > 0) {
        // The caller fits the noise threshold (snowgpu_set_threshold_callback), one group = the whole (small) batch: device half of the
        // prepass, its results down, the per-beam kernels meanwhile, callback, polynomials up, compaction.
        constexpr size_t HIST = (size_t)50 * 2555;
        const size_t o_rec = nfz * HIST * 4, o_thr = o_rec + nfz * SG_PRE_REC * 8, o_stat = o_thr + nfz * 24, need = o_stat + 64;
        if (need > ctx->thr_stage_cap) {
            if (ctx->thr_stage) (void)hipHostFree(ctx->thr_stage);
            ctx->thr_stage = nullptr; ctx->thr_stage_cap = 0;
            HIPCHK(ctx, hipHostMalloc((void **)&ctx->thr_stage, need + need / 8, hipHostMallocDefault));
            ctx->thr_stage_cap = need + need / 8;
        }
        int32_t *sg_hist = (int32_t *)ctx->thr_stage, *sg_stat = (int32_t *)(ctx->thr_stage + o_stat);
        double *sg_rec = (double *)(ctx->thr_stage + o_rec), *sg_thr = (double *)(ctx->thr_stage + o_thr);
        ENSURE(ctx, ctx->stats_hist, nfz * HIST);
        ENSURE(ctx, ctx->stats_rec, nfz * SG_PRE_REC);
        ENSURE(ctx, ctx->user_thr, nfz * 3);
        while (ctx->thr_ev.size() < 2) {
            hipEvent_t ev;
            HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->thr_ev.push_back(ev);
        }
        const double *pl = d_plane;
        if (!pl) {
            ENSURE(ctx, ctx->plane_est, nfz * 4);
            ENSURE(ctx, ctx->plane_info, nfz * 4);
            int pe = sg_plane_run(&ctx->plane_scr, &ctx->plane_par, d_rows_used, dtype, d_off_used, nullptr, n_frames, n_used, max_frame_used, ctx->plane_est.p, ctx->plane_info.p, st);
            if (pe) return fail(ctx, SNOWGPU_E_HIP, std::string("plane estimate: ") + (pe > 0 ? hipGetErrorString((hipError_t)pe) : "allocation"));
            pl = ctx->plane_est.p;
        }
        HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, 32, st));
        int se2 = sg_prepass_stats_run(&ctx->prepass, d_rows_used, dtype, d_off_used, n_frames, n_used, max_frame_used, pl, ctx->stats_hist.p, ctx->stats_rec.p, ctx->d_status, st);
        if (se2) return fail(ctx, SNOWGPU_E_HIP, std::string("prepass: ") + (se2 > 0 ? hipGetErrorString((hipError_t)se2) : "allocation"));
        HIPCHK(ctx, hipMemcpyAsync(sg_hist, ctx->stats_hist.p, nfz * HIST * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(sg_rec, ctx->stats_rec.p, nfz * SG_PRE_REC * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(sg_stat, ctx->d_status, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipEventRecord(ctx->thr_ev[1], st));
        b.defer_thr = true;
        rc = run_batch(ctx, b);
        if (rc == SNOWGPU_OK) {
            HIPCHK(ctx, hipEventSynchronize(ctx->thr_ev[1]));
            if (sg_stat[0] != 0) {
                (void)hipStreamSynchronize(st);
                std::memcpy(ctx->h_status, sg_stat, 32);
                return status_to_error(ctx, sg_stat);
            }
            if (ctx->thr_fn(ctx->thr_user, 0, n_frames, sg_hist, sg_rec, sg_thr) != 0) {
                (void)hipStreamSynchronize(st);
                return fail(ctx, SNOWGPU_E_INVALID, "the threshold callback reported an error");
            }
            HIPCHK(ctx, hipMemcpyAsync(ctx->user_thr.p, sg_thr, 24 * nfz, hipMemcpyHostToDevice, st));
            b.thr_poly = ctx->user_thr.p;
            rc = run_compaction(ctx, b);
        }
    } else {
        rc = run_batch(ctx, b);
    }
    int32_t status[8] = {0, -1, 0, 0, 0, 0, 0, 0};
    if (rc == SNOWGPU_OK) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->mail_dn_h, ctx->mail_dn_d.p, out_thr_poly ? dn_bytes : dn_thr, hipMemcpyDeviceToHost, st));
        if (row_bytes && !precrop) {
            HIPCHK(ctx, hipMemcpyAsync(out_rows, ctx->rows_out.p, row_bytes, hipMemcpyDeviceToHost, st));
            if (out_src) HIPCHK(ctx, hipMemcpyAsync(out_src, ctx->out_src.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
        } else if (precrop && n_used > 0) {
            // source rows in the ORIGINAL frame: output row -> cropped row -> original row; every frame goes back to its own slot
            int e = sg_launch_compose_src(ctx->crop_off.p, d_counts, n_frames, max_frame_used, ctx->out_src.p, ctx->crop_src.p,
                                          ctx->crop_out_src.p, st);
            if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("compose launch: ") + hipGetErrorString((hipError_t)e));
            for (int f = 0; f < n_frames; ++f) {
                const size_t m = (size_t)(crop_off[(size_t)f + 1] - crop_off[(size_t)f]);
                if (!m) continue;
                HIPCHK(ctx, hipMemcpyAsync((char *)out_rows + (size_t)frame_offsets[f] * 5 * esz, (const char *)ctx->rows_out.p + (size_t)crop_off[(size_t)f] * 5 * esz,
                                           m * 5 * esz, hipMemcpyDeviceToHost, st));
                if (out_src) HIPCHK(ctx, hipMemcpyAsync(out_src + frame_offsets[f], ctx->crop_out_src.p + crop_off[(size_t)f], sizeof(int32_t) * m, hipMemcpyDeviceToHost, st));
            }
        }
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
    std::memcpy(status, ctx->mail_dn_h + dn_st, sizeof status);
    std::memcpy(out_counts, ctx->mail_dn_h + dn_cnt, 8 * nfz);
    std::memcpy(out_stats, ctx->mail_dn_h + dn_stats, 24 * nfz);
    if (out_thr_poly) std::memcpy(out_thr_poly, ctx->mail_dn_h + dn_thr, 24 * nfz);
    std::memcpy(ctx->h_status, status, sizeof status);
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

extern "C" int snowgpu_augment_batch_compact(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const float *xyzi, const uint8_t *channels,
                                             const int32_t *table_ids, double beam_divergence_deg, const double *thr_poly, const double *plane,
                                             double noise_floor, float *out_rows, int32_t *out_src, int64_t *out_counts, int64_t *out_stats,
                                             double *out_thr_poly)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (!xyzi || !channels) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_augment_batch_compact: null input");
    if (ctx->fov.enabled && ctx->fov_pre) return fail(ctx, SNOWGPU_E_INVALID, "the pre-augment crop takes (x, y, z, intensity, channel) rows: snowgpu_augment_batch");
    ctx->in_channels = channels;
    const int rc = host_batch(ctx, n_frames, frame_offsets, xyzi, 0, table_ids, beam_divergence_deg, thr_poly, plane, noise_floor, nullptr, out_rows,
                              out_src, out_counts, out_stats, out_thr_poly, 0, nullptr, nullptr, nullptr, nullptr);
    ctx->in_channels = nullptr;
    return rc;
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
    int64_t rows = 0;
    DevBuf<double> d_xyr;
    for (int attempt = 0;; ++attempt) {
        if (n_cand > ((int64_t)1 << 27)) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_sample_table: table would exceed 2^27 candidates");
        if (d_xyr.ensure((size_t)n_cand * 3)) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed for the sampled table");
        int rc = sg_sample_table(occupancy_ratio, diameter_scale_mm, r_0, seed, n_cand, d_xyr.p, n_cand, &rows, ctx->stream);
        if (rc == -2 && attempt < 4) { d_xyr.release(); n_cand *= 2; continue; }     // not enough darts: throw more
        if (rc != 0) {
            d_xyr.release();
            if (rc > 0) return fail(ctx, SNOWGPU_E_HIP, std::string("sampler: ") + hipGetErrorString((hipError_t)rc));
            return fail(ctx, SNOWGPU_E_TABLE, rc == -3 ? "sampler: a dart overlaps more than 4 earlier darts (occupancy too high for this sampler)"
                                                       : "sampler: acceptance did not settle / target area not reached");
        }
        break;
    }
    *n_out = rows;
    int rc = SNOWGPU_OK;
    if (xyr_out) {
        if (cap < rows) rc = fail(ctx, SNOWGPU_E_INVALID, "snowgpu_sample_table: output buffer too small (see *n_out)");
        else if (rows && hipMemcpy(xyr_out, d_xyr.p, sizeof(double) * 3 * (size_t)rows, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(ctx, SNOWGPU_E_HIP, "sampler copy failed");
    }
    // the table is filed where it was made: derive / bin / sort kernels on the sampler's output, no host round trip
    if (rc == SNOWGPU_OK && table_id >= 0) rc = file_table_device(ctx, table_id, d_xyr.p, rows);
    d_xyr.release();
    return rc;
}

// The status words of the last batch that went through a host-pointer entry of this context (layout: see
// snowgpu_augment_batch_device): how many beams each later capacity tier took.
extern "C" int snowgpu_last_status(snowgpu_ctx *ctx, int32_t *out8)
{
    if (!ctx || !out8) return SNOWGPU_E_INVALID;
    std::memcpy(out8, ctx->h_status, sizeof(int32_t) * 8);
    return SNOWGPU_OK;
}

// NUMA node the HIP device hangs on (sysfs, by its PCI bus id), or -1: for launchers that place one process per GPU next to it.
extern "C" int snowgpu_device_numa_node(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return -1;
    return node_of_device(device);
}

// How the results of a pipelined host-pointer batch cross the link.  mode 0 (default): the output rows and their source indices, 24 bytes
// per point.  mode 1 ("packed"): per kept row its source row | label and its intensity (8 bytes; 12 for float64 rows), the moved
// coordinates of scattered rows apart; `threads` host threads of the library (0: the CPUs this process may use minus two, at most 8) put
// the caller's out_rows / out_src together from those and from the caller's INPUT rows -- same bytes in the caller's buffers, a third of
// the download, and host cores busy copying.  For callers bound by the link who have the cores to spare.
extern "C" int snowgpu_set_result_transfer(snowgpu_ctx *ctx, int mode, int threads)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if ((mode != 0 && mode != 1) || threads < 0 || threads > 256) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_result_transfer: mode 0 or 1, 0 <= threads <= 256");
    if (mode == 1) {
        if (ctx->pool && threads != ctx->asm_threads) { delete ctx->pool; ctx->pool = nullptr; }     // (made again, with `threads`, by the next packed call)
        ctx->asm_threads = threads;
    }
    ctx->result_mode = mode;
    return SNOWGPU_OK;
}

// Timeline of the last packed call of this context, milliseconds since its start: [0] everything enqueued, [1] every download landed,
// [2] every row assembled; [3] host threads used.
extern "C" int snowgpu_debug_transfer_times(snowgpu_ctx *ctx, double *out4)
{
    if (!ctx || !out4) return SNOWGPU_E_INVALID;
    for (int i = 0; i < 3; ++i) out4[i] = ctx->pk_times[i];
    out4[3] = ctx->pool ? (double)ctx->pool->threads.size() : 0.0;
    return SNOWGPU_OK;
}

// Chunk size of the upload / compute / download pipeline of the host-pointer entry; 0 switches the pipeline off.
extern "C" int snowgpu_set_pipeline(snowgpu_ctx *ctx, int64_t chunk_rows)
{
    if (!ctx || chunk_rows < 0) return SNOWGPU_E_INVALID;
    ctx->pipe_rows = chunk_rows;
    return SNOWGPU_OK;
}

// The two fitted lines of estimate_laser_parameters (wet_ground/augmentation.py:216-219, :248-251) for the NEXT
// snowgpu_wet_ground_batch of this context, from a caller who fits them itself -- e.g. with its own NumPy, whose argpartition
// decides quirk Q8 -- instead of the device's fit: n_frames x 4 (p slope, p intercept, noise-line slope, noise-line intercept).
// Everything else (ground rows, incident angles, the < 1000-ground-rows rule, Fresnel chain, noise drop) stays on the device.
extern "C" int snowgpu_set_wet_lines(snowgpu_ctx *ctx, int n_frames, const double *lines)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || !lines) { ctx->wet_lines.clear(); return SNOWGPU_OK; }
    ctx->wet_lines.assign(lines, lines + (size_t)n_frames * 4);
    return SNOWGPU_OK;
}

// estimation_method of ground_water_augmentation (wet_ground/augmentation.py:25, :215-229, :243-253) for the wet-ground calls of this
// context: 0 = 'linear' (two regression lines), 1 = 'poly' (np.polyfit of degree 2 for the laser power, ransac_polyfit for the noise
// level -- the reference draws its RANSAC samples from NumPy's unseeded global generator; here they come from Philox keyed by `seed`).
extern "C" int snowgpu_set_wet_estimation(snowgpu_ctx *ctx, int method, uint64_t seed)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (method != 0 && method != 1) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_wet_estimation: method 0 (linear) or 1 (poly)");
    ctx->wet_estimation = method;
    ctx->wet_seed = seed;
    return SNOWGPU_OK;
}

// The curves the last wet-ground call of this context fitted, per frame: laser power c2, c1, c0 (relative_output_intensity =
// power_factor * polyval), noise level c2, c1, c0 (adaptive_noise_threshold = noise_floor * polyval), ground rows, and the RANSAC
// trial whose consensus refit was kept (-1: the fit over all points; 'linear': always -1 and c2 = 0).  out: n_frames x 8 doubles.
extern "C" int snowgpu_wet_last_fit(snowgpu_ctx *ctx, int n_frames, double *out)
{
    if (!ctx || !out || n_frames <= 0) return SNOWGPU_E_INVALID;
    if (n_frames != ctx->wet_fit_frames) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_wet_last_fit: the last wet-ground call had another number of frames");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, ctx->wet_fit.p, sizeof(double) * 8 * (size_t)n_frames, hipMemcpyDeviceToHost));
    return SNOWGPU_OK;
}

// Parity tap of the 'poly' noise fit: the device's ransac_polyfit(x, y, order=2) on m (3 .. 50) caller-supplied points with the draws of
// (seed; frame) -- out[0..2] = the quadratic's coefficients (highest power first), out[3] = the trial whose consensus refit was kept (-1: none).
extern "C" int snowgpu_debug_ransac_polyfit(snowgpu_ctx *ctx, int m, const double *x, const double *y, uint64_t seed, uint64_t frame, double *out4)
{
    if (!ctx || !x || !y || !out4) return SNOWGPU_E_INVALID;
    if (m < 3 || m > 50) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_debug_ransac_polyfit: 3 <= m <= 50");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<double> d;
    if (d.ensure(2 * 50 + 4)) return fail(ctx, SNOWGPU_E_HIP, "hipMalloc failed");
    int rc = SNOWGPU_OK;
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(d.p, x, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d.p + 50, y, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = (hipError_t)sg_debug_ransac_quad(d.p, d.p + 50, m, seed, frame, d.p + 100, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out4, d.p + 100, sizeof(double) * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = fail(ctx, SNOWGPU_E_HIP, std::string("ransac tap: ") + hipGetErrorString(e));
    d.release();
    return rc;
}

extern "C" int snowgpu_set_exact_math(snowgpu_ctx *ctx, int on)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    ctx->exact_math = on ? 1 : 0;
    return SNOWGPU_OK;
}


extern "C" int snowgpu_set_serial(snowgpu_ctx *ctx, int on)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    ctx->serial = on != 0;
    return SNOWGPU_OK;
}

extern "C" int snowgpu_lane_stream(snowgpu_ctx *ctx, int level, void **stream)
{
    if (!ctx || !stream || level < 0 || level > 2) return SNOWGPU_E_INVALID;
    *stream = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->lane_stream[level]) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const int prio = level == 0 ? greatest : level == 2 ? least : (least + greatest) / 2;      // (HIP: 1 low, 0 normal, -1 high)
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->lane_stream[level], hipStreamNonBlocking, prio));
    }
    *stream = (void *)ctx->lane_stream[level];
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

// ctx may be NULL (or already destroyed by the caller): page-locked memory is not tied to a context, and a buffer handed
// to a caller may outlive the context that allocated it.
extern "C" int snowgpu_host_free(snowgpu_ctx *ctx, void *ptr)
{
    (void)ctx;
    if (!ptr) return SNOWGPU_OK;
    return hipHostFree(ptr) == hipSuccess ? SNOWGPU_OK : SNOWGPU_E_HIP;
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

// augment() followed by ground_water_augmentation() on its output (pointcloud_viewer.py:2807-2821) as ONE launch
// sequence on the caller's stream: the snowfall rows are compacted into context scratch, the wet-ground kernels read
// them there (rows of frame f: [off[f], off[f] + snowfall count[f])), and a last kernel composes the source indices.
// No host copy, no synchronisation, no allocation after the first call of a given size.
extern "C" int snowgpu_augment_wet_batch_device(snowgpu_ctx *ctx, int n_frames, int64_t n_total, int64_t max_frame_rows,
                                                const int64_t *d_frame_offsets, const void *d_rows, int dtype,
                                                const int32_t *d_table_ids, double beam_divergence_deg, const double *d_thr_poly,
                                                const double *d_plane, double noise_floor, const int32_t *d_perm,
                                                const double *d_wet_plane, double water_height, double pavement_depth,
                                                double wet_noise_floor, double power_factor, int flat_earth, double delta, int replace,
                                                double *d_out_rows, int32_t *d_out_src, int64_t *d_out_counts, int64_t *d_out_stats,
                                                int32_t *d_out_flags, int32_t *d_status, void *stream)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || n_total < 0 || !d_frame_offsets || (n_total > 0 && !d_rows) || !d_table_ids || !d_out_rows ||
        !d_out_src || !d_out_counts || !d_out_stats || !d_out_flags || !d_status || (dtype != 0 && dtype != 1))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_augment_wet_batch_device: null pointer or bad dtype");
    if (n_total >= ((int64_t)1 << 31)) return fail(ctx, SNOWGPU_E_INVALID, "batch too large: split it below 2^31 rows");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t esz = dtype == 0 ? 4 : 8, n = (size_t)n_total;
    ENSURE(ctx, ctx->snow_rows, std::max<size_t>(n * 5 * esz, 8));
    ENSURE(ctx, ctx->snow_src, std::max<size_t>(n, 1));
    ENSURE(ctx, ctx->snow_counts, (size_t)n_frames);
    BatchDev b{};
    b.n_frames = n_frames; b.n_total = n_total; b.max_frame = (max_frame_rows > 0 && max_frame_rows <= n_total) ? max_frame_rows : n_total;
    b.frame_off = d_frame_offsets;
    b.uniform_rows = (max_frame_rows > 0 && max_frame_rows * (int64_t)n_frames == n_total) ? max_frame_rows : 0; b.rows = d_rows;
    b.dtype = dtype; b.table_ids = d_table_ids; b.beam_div_deg = beam_divergence_deg; b.thr_poly = d_thr_poly;
    b.plane = d_plane; b.noise_floor = noise_floor; b.perm = d_perm; b.out_rows = ctx->snow_rows.p; b.out_src = ctx->snow_src.p;
    b.out_counts = ctx->snow_counts.p; b.out_stats = d_out_stats; b.out_thr_poly = nullptr; b.status = d_status;
    b.stream = stream ? (hipStream_t)stream : ctx->stream;
    int rc = run_batch(ctx, b);
    if (rc) return rc;
    if (n == 0) {
        HIPCHK(ctx, hipMemsetAsync(d_out_counts, 0, sizeof(int64_t) * (size_t)n_frames, b.stream));
        HIPCHK(ctx, hipMemsetAsync(d_out_flags, 0, sizeof(int32_t) * (size_t)n_frames, b.stream));
        return SNOWGPU_OK;
    }
    SgWetParams wp{};
    wp.water_height = water_height; wp.pavement_depth = pavement_depth; wp.noise_floor = wet_noise_floor;
    wp.power_factor = power_factor; wp.flat_earth = flat_earth; wp.delta = delta; wp.replace = replace;
    wp.estimation = ctx->wet_estimation; wp.seed = ctx->wet_seed;
    ENSURE(ctx, ctx->wet_fit, (size_t)n_frames * 8);
    wp.fit_out = ctx->wet_fit.p; ctx->wet_fit_frames = n_frames;
    int e = 0;
    if (!d_wet_plane) {      // wet_ground/augmentation.py:41 calculate_plane(pointcloud) -- here the snowfall result -- on the device
        ENSURE(ctx, ctx->wet_plane_est, (size_t)n_frames * 4);
        e = sg_plane_run(&ctx->plane_scr, &ctx->plane_par, ctx->snow_rows.p, dtype, d_frame_offsets, ctx->snow_counts.p, n_frames, n_total, b.max_frame,
                         ctx->wet_plane_est.p, nullptr, b.stream);
        d_wet_plane = ctx->wet_plane_est.p;
    }
    // (the source rows of the chained result -- final row -> snowfall row -> input row -- are composed as the wet scatter writes them)
    wp.src_first = ctx->snow_src.p;
    if (!e) e = sg_wet_run(&ctx->prepass, ctx->snow_rows.p, dtype, d_frame_offsets, ctx->snow_counts.p, n_frames, n_total, b.max_frame,
                       d_wet_plane, &wp, d_out_rows, d_out_src, d_out_counts, d_out_flags, d_status, b.stream);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("wet ground: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
    return SNOWGPU_OK;
}

// The same chain for frames in HOST memory: copies in, the device entry above, copies out -- one synchronisation.
extern "C" int snowgpu_augment_wet_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                                         const int32_t *table_ids, double beam_divergence_deg, const double *thr_poly,
                                         const double *plane, double noise_floor, const int32_t *perm, const double *wet_plane,
                                         double water_height, double pavement_depth, double wet_noise_floor, double power_factor,
                                         int flat_earth, double delta, int replace, double *out_rows, int32_t *out_src,
                                         int64_t *out_counts, int64_t *out_stats, int32_t *out_flags)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || !frame_offsets || !table_ids || !out_counts || !out_stats || !out_flags || (dtype != 0 && dtype != 1))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_augment_wet_batch: null pointer or bad dtype");
    if (frame_offsets[0] != 0) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets[0] must be 0");
    if (ctx->h_las.n <= 0) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_lasers has not been called");
    int64_t max_frame = 0;
    bool uni = true;
    for (int f = 0; f < n_frames; ++f) {
        if (frame_offsets[f + 1] < frame_offsets[f]) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets must be non-decreasing");
        max_frame = std::max(max_frame, frame_offsets[f + 1] - frame_offsets[f]);
    }
    for (int f = 0; f < n_frames; ++f) uni = uni && (frame_offsets[f + 1] - frame_offsets[f]) == max_frame;
    const int64_t n_total = frame_offsets[n_frames];
    if (n_total >= ((int64_t)1 << 31)) return fail(ctx, SNOWGPU_E_INVALID, "batch too large: split it below 2^31 rows");
    if (n_total > 0 && (!rows || !out_rows || !out_src)) return fail(ctx, SNOWGPU_E_INVALID, "null row buffers");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t esz = dtype == 0 ? 4 : 8, n = (size_t)n_total, nf = (size_t)n_frames, nl = (size_t)ctx->h_las.n;
    hipStream_t st = ctx->stream;
    ENSURE(ctx, ctx->rows_in, std::max<size_t>(n * 5 * esz, 8));
    ENSURE(ctx, ctx->wet_rows, std::max<size_t>(n * 5, 1));
    ENSURE(ctx, ctx->out_src, std::max<size_t>(n, 1));
    ENSURE(ctx, ctx->frame_off, nf + 1);
    ENSURE(ctx, ctx->wet_counts, nf);
    ENSURE(ctx, ctx->wet_flags, nf);
    ENSURE(ctx, ctx->out_stats, nf * 3);
    ENSURE(ctx, ctx->table_ids, nf * nl);
    ENSURE(ctx, ctx->plane, nf * 4);
    ENSURE(ctx, ctx->wet_plane, nf * 4);
    ctx->resident_rows = -1;
    if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->rows_in.p, rows, n * 5 * esz, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->frame_off.p, frame_offsets, sizeof(int64_t) * (nf + 1), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->table_ids.p, table_ids, sizeof(int32_t) * nf * nl, hipMemcpyHostToDevice, st));
    if (wet_plane) HIPCHK(ctx, hipMemcpyAsync(ctx->wet_plane.p, wet_plane, sizeof(double) * 4 * nf, hipMemcpyHostToDevice, st));
    const double *d_thr = nullptr;
    if (thr_poly) {
        ENSURE(ctx, ctx->user_thr, nf * 3);
        HIPCHK(ctx, hipMemcpyAsync(ctx->user_thr.p, thr_poly, sizeof(double) * 3 * nf, hipMemcpyHostToDevice, st));
        d_thr = ctx->user_thr.p;
    } else if (plane) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->plane.p, plane, sizeof(double) * 4 * nf, hipMemcpyHostToDevice, st));
    }
    if (perm) {
        ENSURE(ctx, ctx->user_perm, std::max<size_t>(n, 1));
        if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->user_perm.p, perm, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    }
    int rc = snowgpu_augment_wet_batch_device(ctx, n_frames, n_total, uni ? max_frame : std::max<int64_t>(max_frame, 1) , ctx->frame_off.p,
                                              ctx->rows_in.p, dtype, ctx->table_ids.p, beam_divergence_deg, d_thr, (d_thr || !plane) ? nullptr : ctx->plane.p,
                                              noise_floor, perm ? ctx->user_perm.p : nullptr, wet_plane ? ctx->wet_plane.p : nullptr, water_height,
                                              pavement_depth, wet_noise_floor, power_factor, flat_earth, delta, replace, ctx->wet_rows.p,
                                              ctx->out_src.p, ctx->wet_counts.p, ctx->out_stats.p, ctx->wet_flags.p, ctx->d_status, st);
    int32_t status[8] = {0, -1, 0, 0, 0, 0, 0, 0};
    if (rc == SNOWGPU_OK) {
        HIPCHK(ctx, hipMemcpyAsync(status, ctx->d_status, sizeof status, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(out_counts, ctx->wet_counts.p, sizeof(int64_t) * nf, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(out_flags, ctx->wet_flags.p, sizeof(int32_t) * nf, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(out_stats, ctx->out_stats.p, sizeof(int64_t) * 3 * nf, hipMemcpyDeviceToHost, st));
        if (n) {
            HIPCHK(ctx, hipMemcpyAsync(out_rows, ctx->wet_rows.p, n * 5 * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipMemcpyAsync(out_src, ctx->out_src.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
        }
    }
    hipError_t se = hipStreamSynchronize(st);
    if (rc != SNOWGPU_OK) return rc;
    if (se != hipSuccess) return fail(ctx, SNOWGPU_E_HIP, std::string("stream synchronize: ") + hipGetErrorString(se));
    std::memcpy(ctx->h_status, status, sizeof status);
    return status_to_error(ctx, status);
}

// Camera-FOV crop of augment(only_camera_fov=True) (simulation.py:39-47, :532-540): lidar_to_rect with
// Tr_velo_to_cam (3 x 4) and R0_rect (3 x 3), rect_to_img with P2 (3 x 4), image img_h x img_w ((1024, 1920) in the
// reference).  The crop is applied by the compaction of every later batch of this context (and num_removed counts it,
// :538) until it is switched off again.  The reference's own projection code is un-vendored: textbook KITTI, float64.
// precompute.py:96-99 crops every frame to the camera's view BEFORE augment() sees it.  With this switch on (and a crop set
// by snowgpu_set_fov) the host-pointer entry snowgpu_augment_batch does the same on the device, right after the upload;
// statistics and out_src then refer to what augment() would have been given / to rows of the original frame.
extern "C" int snowgpu_set_fov_precrop(snowgpu_ctx *ctx, int on)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    ctx->fov_pre = on ? 1 : 0;
    return SNOWGPU_OK;
}

extern "C" int snowgpu_set_fov(snowgpu_ctx *ctx, int enabled, const double *v2c, const double *r0, const double *p2, int img_h, int img_w)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (!enabled) { ctx->fov.enabled = 0; return SNOWGPU_OK; }
    if (!v2c || !r0 || !p2 || img_h <= 0 || img_w <= 0) return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_fov: need V2C, R0, P2 and an image size");
    SgFov f{};
    f.enabled = 1;
    for (int i = 0; i < 4; ++i)                      // M = V2C^T . R0^T  (4 x 3)
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc = acc + v2c[4 * k + i] * r0[3 * j + k];
            f.m[3 * i + j] = acc;
        }
    for (int i = 0; i < 12; ++i) f.p[i] = p2[i];
    f.img_h = img_h; f.img_w = img_w;
    ctx->fov = f;
    return SNOWGPU_OK;
}

extern "C" int snowgpu_wet_ground_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                                        const double *plane, double water_height, double pavement_depth, double noise_floor,
                                        double power_factor, int flat_earth, double delta, int replace, double *out_rows,
                                        int32_t *out_src, int64_t *out_counts, int32_t *out_flags)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || !frame_offsets || !out_counts || !out_flags || (dtype != 0 && dtype != 1))
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
    ctx->resident_rows = -1;
    if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->rows_in.p, rows, n * 5 * esz, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->frame_off.p, frame_offsets, sizeof(int64_t) * ((size_t)n_frames + 1), hipMemcpyHostToDevice, st));
    if (plane) HIPCHK(ctx, hipMemcpyAsync(ctx->plane.p, plane, sizeof(double) * 4 * (size_t)n_frames, hipMemcpyHostToDevice, st));
    else {                   // wet_ground/augmentation.py:41 calculate_plane(pointcloud) on the device
        int pe = sg_plane_run(&ctx->plane_scr, &ctx->plane_par, ctx->rows_in.p, dtype, ctx->frame_off.p, nullptr, n_frames, n_total, max_frame,
                              ctx->plane.p, nullptr, st);
        if (pe) return fail(ctx, SNOWGPU_E_HIP, std::string("plane estimate: ") + (pe > 0 ? hipGetErrorString((hipError_t)pe) : "allocation"));
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t) * 8, st));
    SgWetParams wp{};
    wp.water_height = water_height; wp.pavement_depth = pavement_depth; wp.noise_floor = noise_floor;
    wp.power_factor = power_factor; wp.flat_earth = flat_earth; wp.delta = delta; wp.replace = replace;
    wp.estimation = ctx->wet_estimation; wp.seed = ctx->wet_seed;
    ENSURE(ctx, ctx->wet_fit, (size_t)n_frames * 8);
    wp.fit_out = ctx->wet_fit.p; ctx->wet_fit_frames = n_frames;
    if (!ctx->wet_lines.empty() && ctx->wet_estimation != 0) { ctx->wet_lines.clear(); return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_wet_lines supplies LINES: not with estimation method 'poly'"); }
    if (!ctx->wet_lines.empty()) {                      // the caller's lines (one use)
        if (ctx->wet_lines.size() != (size_t)n_frames * 4) { ctx->wet_lines.clear(); return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_wet_lines was given another number of frames"); }
        ENSURE(ctx, ctx->d_wet_lines, ctx->wet_lines.size());
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_wet_lines.p, ctx->wet_lines.data(), sizeof(double) * ctx->wet_lines.size(), hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        wp.lines = ctx->d_wet_lines.p;
        ctx->wet_lines.clear();
    }
    int e = sg_wet_run(&ctx->prepass, ctx->rows_in.p, dtype, ctx->frame_off.p, nullptr, n_frames, n_total, max_frame, ctx->plane.p, &wp,
                       (double *)ctx->rows_out.p, ctx->out_src.p, ctx->out_counts.p, ctx->dbg_count.p, ctx->d_status, st);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("wet ground: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
    HIPCHK(ctx, hipMemcpyAsync(out_counts, ctx->out_counts.p, sizeof(int64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(out_flags, ctx->dbg_count.p, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToHost, st));
    if (n) {
        HIPCHK(ctx, hipMemcpyAsync(out_rows, ctx->rows_out.p, n * 5 * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(out_src, ctx->out_src.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    }
    int32_t status[8] = {0, -1, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(status, ctx->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (status[0] == SNOWGPU_E_GROUND)     // only 'poly' reports here: np.polyfit of degree 2 over fewer than 3 points (augmentation.py:243)
        return fail(ctx, SNOWGPU_E_GROUND, "estimation method 'poly': fewer than 3 range rows of the histogram have a sparsest bin above 5");
    return status_to_error(ctx, status);
}


// ---- ground plane (tools/wet_ground/planes.py:12-50) -------------------------------------------------------------------
extern "C" int snowgpu_set_plane_method(snowgpu_ctx *ctx, int method, uint64_t seed, int max_trials, int min_rows, double standard_height)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (method < SG_PLANE_REFERENCE || method > SG_PLANE_RANSAC || max_trials < 0 || max_trials > (1 << 16) || min_rows < 0)
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_set_plane_method: method 0 (reference), 1 (least squares) or 2 (ransac); 0 <= trials <= 65536");
    ctx->plane_par.method = method;
    ctx->plane_par.seed = seed;
    ctx->plane_par.trials = max_trials > 0 ? max_trials : 1024;
    ctx->plane_par.min_rows = min_rows;
    ctx->plane_par.std_height = standard_height;
    return SNOWGPU_OK;
}

extern "C" int snowgpu_estimate_planes_device(snowgpu_ctx *ctx, int n_frames, int64_t n_total, int64_t max_frame_rows,
                                              const int64_t *d_frame_offsets, const void *d_rows, int dtype, double *d_out_planes,
                                              int32_t *d_out_info, void *stream)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || n_total < 0 || !d_frame_offsets || (n_total > 0 && !d_rows) || !d_out_planes || (dtype != 0 && dtype != 1))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_estimate_planes_device: null pointer or bad dtype");
    if (n_total >= ((int64_t)1 << 31)) return fail(ctx, SNOWGPU_E_INVALID, "batch too large: split it below 2^31 rows");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t mf = (max_frame_rows > 0 && max_frame_rows <= n_total) ? max_frame_rows : n_total;
    int e = sg_plane_run(&ctx->plane_scr, &ctx->plane_par, d_rows, dtype, d_frame_offsets, nullptr, n_frames, n_total, mf, d_out_planes, d_out_info,
                         stream ? (hipStream_t)stream : ctx->stream);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("plane estimate: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
    return SNOWGPU_OK;
}

extern "C" int snowgpu_estimate_planes(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                                       double *out_planes, int32_t *out_info)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || !frame_offsets || !out_planes || (dtype != 0 && dtype != 1))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_estimate_planes: null pointer or bad dtype");
    if (frame_offsets[0] != 0) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets[0] must be 0");
    int64_t max_frame = 0;
    for (int f = 0; f < n_frames; ++f) {
        if (frame_offsets[f + 1] < frame_offsets[f]) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets must be non-decreasing");
        max_frame = std::max(max_frame, frame_offsets[f + 1] - frame_offsets[f]);
    }
    const int64_t n_total = frame_offsets[n_frames];
    if (n_total >= ((int64_t)1 << 31)) return fail(ctx, SNOWGPU_E_INVALID, "batch too large: split it below 2^31 rows");
    if (n_total > 0 && !rows) return fail(ctx, SNOWGPU_E_INVALID, "null row buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t esz = dtype == 0 ? 4 : 8, nf = (size_t)n_frames;
    hipStream_t st = ctx->stream;
    ENSURE(ctx, ctx->frame_off, nf + 1);
    ENSURE(ctx, ctx->plane_est, nf * 4);
    ENSURE(ctx, ctx->plane_info, nf * 4);
    HIPCHK(ctx, hipMemcpyAsync(ctx->frame_off.p, frame_offsets, sizeof(int64_t) * (nf + 1), hipMemcpyHostToDevice, st));
    if (ctx->plane_par.method != SG_PLANE_REFERENCE && n_total > 0) {       // (the reference-today plane reads no row)
        ctx->resident_rows = -1;
        ENSURE(ctx, ctx->rows_in, (size_t)n_total * 5 * esz);
        HIPCHK(ctx, hipMemcpyAsync(ctx->rows_in.p, rows, (size_t)n_total * 5 * esz, hipMemcpyHostToDevice, st));
    }
    int e = sg_plane_run(&ctx->plane_scr, &ctx->plane_par, ctx->rows_in.p, dtype, ctx->frame_off.p, nullptr, n_frames, n_total, max_frame,
                         ctx->plane_est.p, ctx->plane_info.p, st);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("plane estimate: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
    HIPCHK(ctx, hipMemcpyAsync(out_planes, ctx->plane_est.p, sizeof(double) * 4 * nf, hipMemcpyDeviceToHost, st));
    if (out_info) HIPCHK(ctx, hipMemcpyAsync(out_info, ctx->plane_info.p, sizeof(int32_t) * 4 * nf, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return SNOWGPU_OK;
}


// ---- noise-threshold prepass, first half (simulation.py:449-461; wet_ground/augmentation.py:195-235) -------------------------
// For a caller that wants the reference's answer on ITS machine (quirk Q8): the 50 x 2555 histogram of (range, I / cos) over the
// ground rows and the per-frame sums, from the device; the caller takes np.argpartition(hist, 2)[:, 0] itself, fits the noise
// line and the quadratic from the sums, and hands the polynomials to snowgpu_augment_batch (thr_poly).
extern "C" int snowgpu_set_threshold_callback(snowgpu_ctx *ctx, snowgpu_threshold_fn fn, void *user)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    ctx->thr_fn = fn; ctx->thr_user = fn ? user : nullptr;
    return SNOWGPU_OK;
}

extern "C" int snowgpu_prepass_stats(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                                     const double *plane, int32_t *out_hist, double *out_rec)
{
    if (!ctx) return SNOWGPU_E_INVALID;
    if (n_frames <= 0 || !frame_offsets || !out_hist || !out_rec || (dtype != 0 && dtype != 1))
        return fail(ctx, SNOWGPU_E_INVALID, "snowgpu_prepass_stats: null pointer or bad dtype");
    if (frame_offsets[0] != 0) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets[0] must be 0");
    int64_t max_frame = 0;
    for (int f = 0; f < n_frames; ++f) {
        if (frame_offsets[f + 1] < frame_offsets[f]) return fail(ctx, SNOWGPU_E_INVALID, "frame_offsets must be non-decreasing");
        max_frame = std::max(max_frame, frame_offsets[f + 1] - frame_offsets[f]);
    }
    const int64_t n_total = frame_offsets[n_frames];
    if (n_total >= ((int64_t)1 << 31)) return fail(ctx, SNOWGPU_E_INVALID, "batch too large: split it below 2^31 rows");
    if (n_total > 0 && !rows) return fail(ctx, SNOWGPU_E_INVALID, "null row buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t esz = dtype == 0 ? 4 : 8, nf = (size_t)n_frames, hist_n = nf * 50 * 2555;
    hipStream_t st = ctx->stream;
    ENSURE(ctx, ctx->frame_off, nf + 1);
    ENSURE(ctx, ctx->plane, nf * 4);
    ENSURE(ctx, ctx->plane_info, nf * 4);
    ENSURE(ctx, ctx->rows_in, std::max<size_t>((size_t)n_total * 5 * esz, 8));
    ENSURE(ctx, ctx->stats_hist, hist_n);
    ENSURE(ctx, ctx->stats_rec, nf * SG_PRE_REC);
    HIPCHK(ctx, hipMemcpyAsync(ctx->frame_off.p, frame_offsets, sizeof(int64_t) * (nf + 1), hipMemcpyHostToDevice, st));
    if (n_total) HIPCHK(ctx, hipMemcpyAsync(ctx->rows_in.p, rows, (size_t)n_total * 5 * esz, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t) * 8, st));
    if (plane) HIPCHK(ctx, hipMemcpyAsync(ctx->plane.p, plane, sizeof(double) * 4 * nf, hipMemcpyHostToDevice, st));
    else {
        int pe = sg_plane_run(&ctx->plane_scr, &ctx->plane_par, ctx->rows_in.p, dtype, ctx->frame_off.p, nullptr, n_frames, n_total, max_frame,
                              ctx->plane.p, ctx->plane_info.p, st);
        if (pe) return fail(ctx, SNOWGPU_E_HIP, std::string("plane estimate: ") + (pe > 0 ? hipGetErrorString((hipError_t)pe) : "allocation"));
    }
    int e = sg_prepass_stats_run(&ctx->prepass, ctx->rows_in.p, dtype, ctx->frame_off.p, n_frames, n_total, max_frame, ctx->plane.p,
                                 ctx->stats_hist.p, ctx->stats_rec.p, ctx->d_status, st);
    if (e) return fail(ctx, SNOWGPU_E_HIP, std::string("prepass: ") + (e > 0 ? hipGetErrorString((hipError_t)e) : "allocation"));
    int32_t status[8] = {0, -1, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(out_hist, ctx->stats_hist.p, sizeof(int32_t) * hist_n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(out_rec, ctx->stats_rec.p, sizeof(double) * nf * SG_PRE_REC, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(status, ctx->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->resident_rows = n_total; ctx->resident_dtype = dtype;
    ctx->resident_off.assign(frame_offsets, frame_offsets + n_frames + 1);
    return status_to_error(ctx, status);
}
