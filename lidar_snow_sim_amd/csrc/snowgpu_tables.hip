// snowgpu_tables.hip -- filing a snowflake table on the device.
//
// A table (K rows of x, y, disk radius; tools/snowfall/sampling.py:183) is "filed" once before it is used: the per-flake,
// beam-independent quantities of get_occlusions are hoisted out of the beam loop (simulation.py:332, :351-352: range and
// azimuth; geometry.py:138-190: tangents from the origin; geometry.py:32-80: tangent angles, ordered (right, left)), and
// the flakes are binned by azimuth (2048 bins, every bin a flake's angular interval +- 1e-6 rad touches) and sorted by
// range inside each bin.  snowgpu_upload_table does this on the host with glibc's libm -- the bit-exact path for the
// reference's .npy tables.  This file does the same on the device for tables that are BORN there (snowgpu_sample_table),
// so that they never visit the host: derive -> per-bin histogram -> scan -> scatter -> per-bin rank sort.  atan2 / atan are
// rounded correctly here (sg_atan_cr.h: double-double arithmetic), which glibc's are not quite -- 0.5056 ULP observed, 7 in
// 10^4 inputs differ in the last bit -- so a device-filed table equals the host-filed one up to that (and asin, OCML's, only
// widens the bin range a flake is filed under); for a sampled table, whose flakes have no reference counterpart, it is
// immaterial either way (DESIGN.md "on-device sampler").
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "sg_common.h"
#include "sg_atan_cr.h"

#define TB 256

__device__ __forceinline__ bool tb_forward(double ray, double centre)     // geometry.py:59-66 (as do_angles_intersect_particles)
{
    const double d = ray - centre;
    return (fabs(d) < SG_PI / 2) || (fabs(d - SG_TWO_PI) < SG_PI / 2) || (fabs(d + SG_TWO_PI) < SG_PI / 2);
}

// The device twin of derive_flake (snowgpu_api.cpp): same operations in the same order.
__device__ __forceinline__ bool tb_derive(double x, double y, double r, SgEntry &f)
{
    if (!(isfinite(x) && isfinite(y) && isfinite(r)) || !(r > 0)) return false;
    f.x = x; f.y = y; f.r = r;
    f.rho = sqrt(x * x + y * y);                                          // simulation.py:332
    if (!(f.rho > r)) return false;                                       // disk contains the origin
    f.phi = sg_atan2_cr(y, x);                                            // simulation.py:351 (rounded correctly, sg_atan_cr.h)
    if (f.phi < 0) f.phi = f.phi + SG_TWO_PI;                             // :352
    double a[2], b[2];
    const double disc = r * sqrt(x * x + y * y - r * r);                  // geometry.py:161
    if (fabs(x) - r == 0) {                                               // geometry.py:166-176
        a[0] = 1.0; b[0] = 0.0;
        a[1] = (y * y - x * x) / (2 * x * y); b[1] = -1.0;
    } else {
        a[0] = (-x * y + disc) / (r * r - x * x);                         // geometry.py:178-184
        a[1] = (-x * y - disc) / (r * r - x * x);
        b[0] = b[1] = -1.0;
    }
    double ang[2];
    for (int i = 0; i < 2; ++i) {                                         // geometry.py:47-72
        double ray1 = sg_atan_cr(-a[i] / b[i]);
        double ray2 = ray1 + SG_PI;
        if (ray1 < 0) ray1 = ray1 + SG_TWO_PI;
        ray1 = fabs(ray1);
        if (b[i] == 0) { ray1 = SG_PI / 2; ray2 = 3 * SG_PI / 2; }
        const bool ok1 = tb_forward(ray1, f.phi), ok2 = tb_forward(ray2, f.phi);
        if (ok1 == ok2) return false;
        ang[i] = ok1 ? ray1 : ray2;
    }
    const double lo = fmin(ang[0], ang[1]), hi = fmax(ang[0], ang[1]);   // geometry.py:74-78
    if (hi - lo > SG_PI) { f.t0 = hi; f.t1 = lo; } else { f.t0 = lo; f.t1 = hi; }
    return true;
}

__device__ __forceinline__ int tb_bin_of(double theta, double inv_w, int nb)
{
    theta = fmod(theta, SG_TWO_PI);
    if (theta < 0) theta += SG_TWO_PI;
    int b = (int)floor(theta * inv_w);
    if (b < 0) b = 0;
    if (b >= nb) b = nb - 1;
    return b;
}

// 1. per flake: derived record, first bin and bin count; per-bin histogram.  bad[0] = first row that is not a disk clear of
// the origin (or -1).
__global__ __launch_bounds__(TB) void k_file_derive(const double *__restrict__ xyr, int64_t k, SgEntry *__restrict__ fl,
                                                    int32_t *__restrict__ b0, int32_t *__restrict__ span,
                                                    uint32_t *__restrict__ count, int32_t *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= k) return;
    const int nb = SG_NBINS;
    const double inv_w = nb / SG_TWO_PI;
    SgEntry f;
    f.flags = 0; f.src = (uint32_t)i;
    if (!tb_derive(xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2], f)) {
        atomicMax(bad, (int32_t)(0x7fffffff - (i < 0x7fffffff ? i : 0x7ffffffe)));   // smallest offending row wins
        span[i] = 0;
        return;
    }
    const double alpha = asin(fmin(1.0, f.r / f.rho));
    const double lo = f.phi - alpha - SG_BIN_MARGIN, hi = f.phi + alpha + SG_BIN_MARGIN;
    int first, s;
    if (hi - lo >= SG_TWO_PI - 2.0 / inv_w) { first = 0; s = nb; }
    else {
        const int bl = tb_bin_of(lo, inv_w, nb), bh = tb_bin_of(hi, inv_w, nb);
        first = bl;
        s = bh - bl;
        if (s < 0) s += nb;
        s += 1;
    }
    fl[i] = f; b0[i] = first; span[i] = s;
    for (int t = 0; t < s; ++t) atomicAdd(&count[(first + t) % nb], 1u);
}

// 2. exclusive scan of the 2048 bin counts (one block); start[nb] = records in all bins; fill cursors zeroed
__global__ __launch_bounds__(1024) void k_file_scan(const uint32_t *__restrict__ count, uint32_t *__restrict__ start,
                                                    uint32_t *__restrict__ fill, uint32_t *__restrict__ max_bin)
{
    __shared__ uint32_t s[1024];
    const int t = threadIdx.x;
    const uint32_t c0 = count[2 * t], c1 = count[2 * t + 1];
    s[t] = c0 + c1;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const uint32_t add = t >= d ? s[t - d] : 0; __syncthreads(); s[t] += add; __syncthreads(); }
    const uint32_t base = s[t] - (c0 + c1);
    start[2 * t] = base; start[2 * t + 1] = base + c0;
    fill[2 * t] = 0; fill[2 * t + 1] = 0;
    atomicMax(max_bin, c0 > c1 ? c0 : c1);
    if (t == 1023) start[SG_NBINS] = s[1023];
}

// 3. every flake into every bin of its range (arbitrary order inside a bin; flags bit 0 = first bin of the range)
__global__ __launch_bounds__(TB) void k_file_scatter(int64_t k, const SgEntry *__restrict__ fl, const int32_t *__restrict__ b0,
                                                     const int32_t *__restrict__ span, const uint32_t *__restrict__ start,
                                                     uint32_t *__restrict__ fill, SgEntry *__restrict__ tmp)
{
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= k) return;
    const int s = span[i];
    if (s <= 0) return;
    SgEntry e = fl[i];
    const int first = b0[i];
    for (int t = 0; t < s; ++t) {
        const int b = (first + t) % SG_NBINS;
        e.flags = t == 0 ? 1u : 0u;
        tmp[start[b] + atomicAdd(&fill[b], 1u)] = e;
    }
}

// 4. one wave per bin: rank sort by (range, table row) -- the order std::sort gives the host-filed bins
__global__ __launch_bounds__(64) void k_file_sort(const uint32_t *__restrict__ start, const SgEntry *__restrict__ tmp,
                                                  SgEntry *__restrict__ out)
{
    const int b = blockIdx.x;
    const uint32_t e0 = start[b], m = start[b + 1] - e0;
    for (uint32_t i = threadIdx.x; i < m; i += 64) {
        const SgEntry me = tmp[e0 + i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const double rj = tmp[e0 + j].rho;
            const uint32_t sj = tmp[e0 + j].src;
            rank += (rj < me.rho) || (rj == me.rho && sj < me.src);
        }
        out[e0 + rank] = me;
    }
    if (b == 0 && threadIdx.x == 0) {                     // the spare record behind the last bin (the scan prefetches e + 1)
        SgEntry z{};
        out[start[SG_NBINS]] = z;
    }
}

// Coarse range index of a filed table: q[b][k] = records of bin b nearer than SG_QSTEP_M * k metres.  The scan pass starts
// its search for "records nearer than the target" from the two entries around the target's range instead of the whole bin.
__global__ __launch_bounds__(64) void k_table_index(const SgEntry *__restrict__ entries, const uint32_t *__restrict__ start,
                                                    uint32_t *__restrict__ q)
{
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= SG_QSTEPS) return;
    const uint32_t e0 = start[b];
    uint32_t lo = e0, hi = start[b + 1];
    const double lim = SG_QSTEP_M * (double)k;
    while (lo < hi) {
        const uint32_t m = (lo + hi) >> 1;
        if (entries[m].rho < lim) lo = m + 1; else hi = m;
    }
    q[b * SG_QSTEPS + k] = lo - e0;
}

// per-flake quantities of a filed table by table row (debug tap): the copy filed under the flake's first bin
__global__ __launch_bounds__(TB) void k_table_dump(const SgEntry *__restrict__ entries, uint32_t n_entries, double *__restrict__ out /* K x 4 */)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_entries) return;
    const SgEntry e = entries[i];
    if (!(e.flags & 1u)) return;
    double *o = out + (size_t)e.src * 4;
    o[0] = e.rho; o[1] = e.phi; o[2] = e.t0; o[3] = e.t1;
}

#define TCHK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

// Stage A (derive + histogram + scan).  scratch: fl K records, b0 / span K ints, count / start / fill nb + 1 each,
// misc[0] = bad-row marker, misc[1] = longest bin.  The caller then reads start[nb] (and misc) and allocates the bins.
extern "C" int sg_file_table_stage_a(const double *d_xyr, int64_t k, SgEntry *fl, int32_t *b0, int32_t *span, uint32_t *count,
                                     uint32_t *start, uint32_t *fill, int32_t *misc, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, sizeof(uint32_t) * (SG_NBINS + 1), st) != hipSuccess) return (int)hipGetLastError();
    if (hipMemsetAsync(misc, 0, sizeof(int32_t) * 2, st) != hipSuccess) return (int)hipGetLastError();
    if (k > 0) {
        hipLaunchKernelGGL(k_file_derive, dim3((unsigned)((k + TB - 1) / TB)), dim3(TB), 0, st, d_xyr, k, fl, b0, span, count, misc);
        TCHK();
    }
    hipLaunchKernelGGL(k_file_scan, dim3(1), dim3(1024), 0, st, count, start, fill, (uint32_t *)(misc + 1));
    TCHK();
    return 0;
}

// Stage B (scatter + per-bin sort) into `entries` (start[nb] + 1 records); tmp holds start[nb] records.
extern "C" int sg_file_table_stage_b(int64_t k, const SgEntry *fl, const int32_t *b0, const int32_t *span, const uint32_t *start,
                                     uint32_t *fill, SgEntry *tmp, SgEntry *entries, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (k > 0) {
        hipLaunchKernelGGL(k_file_scatter, dim3((unsigned)((k + TB - 1) / TB)), dim3(TB), 0, st, k, fl, b0, span, start, fill, tmp);
        TCHK();
    }
    hipLaunchKernelGGL(k_file_sort, dim3(SG_NBINS), dim3(64), 0, st, start, tmp, entries);
    TCHK();
    return 0;
}

extern "C" int sg_table_index(const SgEntry *entries, const uint32_t *start, uint32_t *q, void *stream)
{
    hipLaunchKernelGGL(k_table_index, dim3(SG_NBINS), dim3(64), 0, (hipStream_t)stream, entries, start, q);
    TCHK();
    return 0;
}

extern "C" int sg_table_dump(const SgEntry *entries, uint32_t n_entries, double *d_out, void *stream)
{
    if (n_entries == 0) return 0;
    hipLaunchKernelGGL(k_table_dump, dim3((n_entries + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, entries, n_entries, d_out);
    TCHK();
    return 0;
}
