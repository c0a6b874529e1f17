// snowgpu_sampler.hip -- snowflake tables sampled on the device.
//
// Counterpart of tools/snowfall/sampling.py::dart_throwing (:90-194): non-overlapping disks in a disc of radius R0,
// centres uniform in area (:145-149), sphere diameter ~ Exp(scale) mm truncated at 20 mm (:153-154), disk = slice of
// the sphere at a uniform height (:160-163), no disk over the origin (:166), no two disks overlapping (:170-174),
// stop as soon as the occupied area reaches occupancy * pi * R0^2 (:142).
//
// The reference throws darts one after another from a NumPy PCG64 stream; that stream is inherently sequential, so
// this sampler reproduces the PROCESS, not the stream (statistical parity):
//   * candidate i is drawn independently from Philox4x32-10 keyed by (seed, i) -- one thread per candidate;
//   * "dart i is rejected iff it overlaps an ACCEPTED dart j < i" is resolved exactly: overlapping pairs are found
//     through a uniform grid (cell 0.5 m, 3 x 3 neighbourhood), and the acceptance recurrence is iterated to its
//     fixed point (overlaps are rare -- occupancy ~1e-6 -- so two sweeps settle it);
//   * the stop rule is the reference's: accepted darts in index order, cut at the first one whose cumulative area
//     reaches the target (that dart is kept, as in the reference's loop).
// The host-side mirror lidar_snow_sim_amd.tools.snowfall.sampling.dart_throwing stays bit-exact with the reference
// for a given NumPy Generator; this one is for making tables where they are used.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "sg_common.h"
#include "sg_philox.h"

#define SB 256
#define SG_SAMP_CELL 0.5
#define SG_SAMP_MAXCONF 4

struct SgCand { double x, y, r; int32_t valid; int32_t state; };   // state: 0 undecided, 1 accepted, 2 rejected

struct SampArgs {
    int64_t n;                // candidates
    double R0, scale_mm;      // domain radius [m], exponential scale of the sphere diameter [mm]
    uint64_t seed;
    SgCand *cand;
    int32_t *cell_of;         // per candidate
    int32_t *cell_count;      // per cell (+1)
    int32_t *cell_start;      // per cell (+1)
    int32_t *cell_items;      // candidates grouped by cell
    int32_t *conf;            // per candidate: up to SG_SAMP_MAXCONF lower-index overlapping candidates (-1 = none)
    int32_t grid;             // cells per side
    int32_t *flags;           // [0] conflict-list overflow, [1] undecided left, [2] changed in this sweep
    double *area_scan;        // inclusive scan of accepted areas
    double target_area;
    int64_t *out_n;           // [0] rows produced, [1] cut index (-1: target not reached)
    double *out_xyr;
    int64_t out_cap;
};

__global__ __launch_bounds__(SB) void k_samp_gen(SampArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= a.n) return;
    double u_len, u_ang, u_h, u_x;
    philox_u2(a.seed, (uint64_t)i, 0, u_len, u_ang);
    philox_u2(a.seed, (uint64_t)i, 1, u_h, u_x);
    const double length = sqrt(u_len * (a.R0 * a.R0));                  // sampling.py:145
    const double angle = (u_ang * 2.0) * SG_PI;                         // :146
    const double x = length * cos(angle), y = length * sin(angle);      // :148-149
    double diam = INFINITY;                                             // :151-154 (mm), redraw above 20 mm
    for (uint32_t g = 2; diam > 20.0 && g < 64; ++g) {
        double e0, e1;
        philox_u2(a.seed, (uint64_t)i, g, e0, e1);
        diam = -(a.scale_mm) * log1p(-e0);
        if (diam > 20.0) diam = -(a.scale_mm) * log1p(-e1);
    }
    diam = diam / 1000.0;                                               // :157
    const double height = (u_h - 0.5) * diam;                           // :160 uniform(-d/2, d/2)
    const double r = sqrt((diam / 2) * (diam / 2) - height * height);   // :163
    SgCand c;
    c.x = x; c.y = y; c.r = r;
    c.valid = (x * x + y * y > r * r) && (r > 0) ? 1 : 0;               // :166 (and a zero-radius slice is no disk)
    c.state = c.valid ? 0 : 2;
    a.cand[i] = c;
    int cx = (int)floor((x + a.R0) / SG_SAMP_CELL), cy = (int)floor((y + a.R0) / SG_SAMP_CELL);
    cx = cx < 0 ? 0 : (cx >= a.grid ? a.grid - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= a.grid ? a.grid - 1 : cy);
    const int cell = cy * a.grid + cx;
    a.cell_of[i] = cell;
    atomicAdd(&a.cell_count[cell], 1);
}

// exclusive scan of the cell counts (one block; ~1e5 cells)
__global__ __launch_bounds__(1024) void k_samp_scan_cells(SampArgs a)
{
    __shared__ int part[1024];
    const int n_cells = a.grid * a.grid, tid = threadIdx.x;
    const int per = (n_cells + 1023) / 1024, lo = tid * per, hi = min(lo + per, n_cells);
    int s = 0;
    for (int c = lo; c < hi; ++c) s += a.cell_count[c];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int c = lo; c < hi; ++c) { a.cell_start[c] = run; run += a.cell_count[c]; a.cell_count[c] = 0; }
    if (tid == 1023) a.cell_start[n_cells] = part[1023];
}

__global__ __launch_bounds__(SB) void k_samp_fill(SampArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= a.n) return;
    const int cell = a.cell_of[i];
    const int slot = atomicAdd(&a.cell_count[cell], 1);
    a.cell_items[a.cell_start[cell] + slot] = (int32_t)i;
}

// lower-index candidates whose disk overlaps candidate i (sampling.py:170)
__global__ __launch_bounds__(SB) void k_samp_overlap(SampArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= a.n) return;
    int32_t *cf = a.conf + i * SG_SAMP_MAXCONF;
    for (int k = 0; k < SG_SAMP_MAXCONF; ++k) cf[k] = -1;
    const SgCand me = a.cand[i];
    if (!me.valid) return;
    const int cell = a.cell_of[i], cx = cell % a.grid, cy = cell / a.grid;
    int n = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int x = cx + dx, y = cy + dy;
            if (x < 0 || y < 0 || x >= a.grid || y >= a.grid) continue;
            const int c = y * a.grid + x;
            for (int s = a.cell_start[c]; s < a.cell_start[c + 1]; ++s) {
                const int j = a.cell_items[s];
                if (j >= i) continue;
                const SgCand o = a.cand[j];
                if (!o.valid) continue;
                const double ddx = o.x - me.x, ddy = o.y - me.y, rr = o.r + me.r;
                if (ddx * ddx + ddy * ddy <= rr * rr) {
                    if (n < SG_SAMP_MAXCONF) cf[n] = j; else atomicExch(&a.flags[0], 1);
                    ++n;
                }
            }
        }
    if (n == 0) a.cand[i].state = 1;             // nothing earlier in the way: accepted whatever happens elsewhere
}

// acc[i] = valid_i and no ACCEPTED earlier dart overlaps it; swept until nothing is undecided
__global__ __launch_bounds__(SB) void k_samp_resolve(SampArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= a.n) return;
    if (a.cand[i].state != 0) return;
    const int32_t *cf = a.conf + i * SG_SAMP_MAXCONF;
    bool blocked = false, pending = false;
    for (int k = 0; k < SG_SAMP_MAXCONF; ++k) {
        const int j = cf[k];
        if (j < 0) continue;
        const int st = a.cand[j].state;          // only ever moves 0 -> 1 or 0 -> 2: a stale 0 just delays us a sweep
        if (st == 1) blocked = true;
        else if (st == 0) pending = true;
    }
    if (blocked) { a.cand[i].state = 2; atomicExch(&a.flags[2], 1); }
    else if (!pending) { a.cand[i].state = 1; atomicExch(&a.flags[2], 1); }
    else atomicExch(&a.flags[1], 1);
}

// cumulative accepted area in index order and the reference's stop rule (sampling.py:142, :181-183); one block
__global__ __launch_bounds__(1024) void k_samp_cut(SampArgs a)
{
    __shared__ double part[1024];
    __shared__ long long cut_s;
    const int tid = threadIdx.x;
    const int64_t per = (a.n + 1023) / 1024, lo = tid * per, hi = lo + per < a.n ? lo + per : a.n;
    double s = 0.0;
    for (int64_t i = lo; i < hi; ++i) if (a.cand[i].state == 1) s += SG_PI * a.cand[i].r * a.cand[i].r;
    part[tid] = s;
    if (tid == 0) cut_s = -1;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const double v = tid >= d ? part[tid - d] : 0.0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    double run = part[tid] - s;
    long long mine = -1;
    for (int64_t i = lo; i < hi; ++i) {
        if (a.cand[i].state == 1) {
            const bool before = run < a.target_area;                     // the loop condition when this dart is thrown
            run += SG_PI * a.cand[i].r * a.cand[i].r;
            if (before && run >= a.target_area && mine < 0) mine = i;    // first dart that fills the target: kept, then stop
        }
        a.area_scan[i] = run;
    }
    if (mine >= 0) atomicMax(&cut_s, mine);                              // at most one thread finds it
    __syncthreads();
    if (tid == 0) a.out_n[1] = cut_s;
}

// rows of the accepted darts up to the cut, in index order (one block, ballot compaction)
__global__ __launch_bounds__(1024) void k_samp_emit(SampArgs a)
{
    __shared__ int wave_cnt[16];
    __shared__ long long base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long cut = a.out_n[1];
    if (tid == 0) base_s = 0;
    __syncthreads();
    if (cut < 0) { if (tid == 0) a.out_n[0] = -1; return; }
    for (int64_t i0 = 0; i0 <= cut; i0 += 1024) {
        const int64_t i = i0 + tid;
        const bool keep = i <= cut && a.cand[i].state == 1;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) off += wave_cnt[w]; tot += wave_cnt[w]; }
        if (keep) {
            const long long row = base_s + off + __popcll(m & ((1ull << lane) - 1ull));
            if (row < a.out_cap) {
                a.out_xyr[3 * row] = a.cand[i].x; a.out_xyr[3 * row + 1] = a.cand[i].y; a.out_xyr[3 * row + 2] = a.cand[i].r;
            }
        }
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    if (tid == 0) a.out_n[0] = base_s;
}

// ---- host ------------------------------------------------------------------------------------------------------------
#define SCHK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { rc = (int)e__; goto done; } } while (0)

// Returns 0 and *n_rows (rows written to d_xyr, capacity cap) ; -2: target area not reached with n_cand candidates;
// -3: a candidate overlapped more than SG_SAMP_MAXCONF earlier ones; -4: acceptance did not settle; >0: hipError_t.
extern "C" int sg_sample_table(double occupancy, double scale_mm, double R0, uint64_t seed, int64_t n_cand, double *d_xyr,
                               int64_t cap, int64_t *n_rows, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    SampArgs a{};
    a.n = n_cand; a.R0 = R0; a.scale_mm = scale_mm; a.seed = seed;
    a.grid = (int)ceil(2.0 * R0 / SG_SAMP_CELL) + 1;
    a.target_area = occupancy * SG_PI * (R0 * R0);                       // sampling.py:124
    a.out_xyr = d_xyr; a.out_cap = cap;
    const size_t n_cells = (size_t)a.grid * (size_t)a.grid;
    void *bufs[9] = {};
    int32_t h_flags[3] = {0, 0, 0};
    int64_t h_out[2] = {0, -1};
    const unsigned blocks = (unsigned)((n_cand + SB - 1) / SB);
    SCHK(hipMalloc(&bufs[0], sizeof(SgCand) * (size_t)n_cand));
    SCHK(hipMalloc(&bufs[1], sizeof(int32_t) * (size_t)n_cand));
    SCHK(hipMalloc(&bufs[2], sizeof(int32_t) * (n_cells + 1)));
    SCHK(hipMalloc(&bufs[3], sizeof(int32_t) * (n_cells + 1)));
    SCHK(hipMalloc(&bufs[4], sizeof(int32_t) * (size_t)n_cand));
    SCHK(hipMalloc(&bufs[5], sizeof(int32_t) * (size_t)n_cand * SG_SAMP_MAXCONF));
    SCHK(hipMalloc(&bufs[6], sizeof(int32_t) * 4));
    SCHK(hipMalloc(&bufs[7], sizeof(double) * (size_t)n_cand));
    SCHK(hipMalloc(&bufs[8], sizeof(int64_t) * 2));
    a.cand = (SgCand *)bufs[0]; a.cell_of = (int32_t *)bufs[1]; a.cell_count = (int32_t *)bufs[2];
    a.cell_start = (int32_t *)bufs[3]; a.cell_items = (int32_t *)bufs[4]; a.conf = (int32_t *)bufs[5];
    a.flags = (int32_t *)bufs[6]; a.area_scan = (double *)bufs[7]; a.out_n = (int64_t *)bufs[8];
    SCHK(hipMemsetAsync(a.cell_count, 0, sizeof(int32_t) * (n_cells + 1), st));
    SCHK(hipMemsetAsync(a.flags, 0, sizeof(int32_t) * 4, st));
    hipLaunchKernelGGL(k_samp_gen, dim3(blocks), dim3(SB), 0, st, a);
    hipLaunchKernelGGL(k_samp_scan_cells, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_samp_fill, dim3(blocks), dim3(SB), 0, st, a);
    hipLaunchKernelGGL(k_samp_overlap, dim3(blocks), dim3(SB), 0, st, a);
    SCHK(hipGetLastError());
    for (int sweep = 0; sweep < 64; ++sweep) {
        SCHK(hipMemsetAsync(a.flags + 1, 0, sizeof(int32_t) * 2, st));
        hipLaunchKernelGGL(k_samp_resolve, dim3(blocks), dim3(SB), 0, st, a);
        SCHK(hipMemcpyAsync(h_flags, a.flags, sizeof(h_flags), hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        if (h_flags[0]) { rc = -3; goto done; }
        if (!h_flags[1]) break;                              // nothing undecided any more
        if (!h_flags[2] && sweep > 0) { rc = -4; goto done; }
    }
    if (h_flags[1]) { rc = -4; goto done; }
    hipLaunchKernelGGL(k_samp_cut, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_samp_emit, dim3(1), dim3(1024), 0, st, a);
    SCHK(hipGetLastError());
    SCHK(hipMemcpyAsync(h_out, a.out_n, sizeof(h_out), hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    if (h_out[1] < 0) { rc = -2; goto done; }
    *n_rows = h_out[0];
done:
    for (void *p : bufs) if (p) (void)hipFree(p);
    return rc;
}
