// snowgpu_kernels.hip -- gfx950 kernels of the snowfall-augmentation engine and their launch wrappers.
//
//   k_sort_*     stable counting sort of every frame's rows by channel                      (simulation.py:447)
//   k_seg_*      launch order of the first pass: (table, frame, channel) segments
//   k_beams      one thread per beam: candidate scan + occlusion dict; the first pass over all rows stops there and
//                hands beams with flakes to k_power, the later capacity tiers run received power + decision in place
//                                                                                           (simulation.py:50-424)
//   k_list_*     ordered work lists from flag bytes (overflowed beams, beams for k_power)
//   k_power      received power on the 10 cm grid, first maximum, attenuate-or-scatter        (simulation.py:135-188)
//   k_compact_*  noise-floor filter, stable stream compaction, stats                         (simulation.py:516-530)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see lidar_snow_sim_amd/build.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "sg_beam.h"

#define SG_BLOCK 256

__device__ __forceinline__ int sg_find_frame(const int64_t *__restrict__ off, int n_frames, int64_t g)
{
    int lo = 0, hi = n_frames - 1;   // largest f with off[f] <= g
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ unsigned long long sg_lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// ------------------------------------------------------------------------------------------------
// Stable counting sort by channel, per frame.  grid = (tiles per frame, frames), 256 threads, a tile
// is 1024 consecutive rows; wave w owns rows [256 w, 256 w + 256) of the tile in 4 rounds of 64 so
// that "earlier row" == "earlier (wave, round, lane)".
template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_sort_hist(const T *__restrict__ rows, const int64_t *__restrict__ frame_off,
                                                        int32_t *__restrict__ tile_hist, uint16_t *__restrict__ rank,
                                                        int32_t *__restrict__ status, int64_t max_tiles)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    __shared__ volatile int cnt[4][256];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 4 * 256; i += SG_BLOCK) ((volatile int *)cnt)[i] = 0;
    __syncthreads();
    int my_bucket[4], my_rank[4];
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + w * 256 + q * 64 + lane;
        const bool valid = r < n;
        int bucket = -1;
        if (valid) {
            const T c = rows[(base + r) * 5 + 4];
            const int ci = (int)c;
            if ((T)ci == c && ci >= 0 && ci < 256) bucket = ci;
            else { atomicCAS(&status[0], 0, 5 /* SNOWGPU_E_CHANNELS */); bucket = 255; }
        }
        my_bucket[q] = bucket;
        my_rank[q] = 0;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int v = __shfl(bucket, leader);
            const unsigned long long m = __ballot(valid && bucket == v);
            const int before = cnt[w][v];
            if (valid && bucket == v) my_rank[q] = before + __popcll(m & sg_lanemask_lt());
            if (lane == leader) cnt[w][v] = before + __popcll(m);
            todo &= ~m;
        }
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + w * 256 + q * 64 + lane;
        if (r < n) {
            int off = my_rank[q];
            for (int ww = 0; ww < w; ++ww) off += cnt[ww][my_bucket[q]];
            rank[base + r] = (uint16_t)off;
        }
    }
    int32_t *h = tile_hist + ((int64_t)f * max_tiles + blockIdx.x) * 256;
    h[tid] = cnt[0][tid] + cnt[1][tid] + cnt[2][tid] + cnt[3][tid];
}

// One block per frame, thread v owns bucket v: tile_base[t][v] = (rows of smaller buckets) + (rows of
// bucket v in earlier tiles).
__global__ __launch_bounds__(SG_BLOCK) void k_sort_scan(const int64_t *__restrict__ frame_off,
                                                        const int32_t *__restrict__ tile_hist,
                                                        int32_t *__restrict__ tile_base, int64_t max_tiles)
{
    const int f = blockIdx.x, v = threadIdx.x;
    const int64_t n = frame_off[f + 1] - frame_off[f];
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    const int32_t *h = tile_hist + (int64_t)f * max_tiles * 256;
    int32_t *b = tile_base + (int64_t)f * max_tiles * 256;
    int total = 0;
#pragma unroll 8                                  // eight loads in flight: the loop is a chain of global-load latencies otherwise
    for (int64_t t = 0; t < tiles; ++t) total += h[t * 256 + v];
    __shared__ int s[256];
    s[v] = total;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {          // Hillis-Steele inclusive scan over the 256 buckets
        int add = v >= d ? s[v - d] : 0;
        __syncthreads();
        s[v] += add;
        __syncthreads();
    }
    int run = s[v] - total;
#pragma unroll 8
    for (int64_t t = 0; t < tiles; ++t) { b[t * 256 + v] = run; run += h[t * 256 + v]; }
}

template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_sort_scatter(const T *__restrict__ rows, const int64_t *__restrict__ frame_off,
                                                           const int32_t *__restrict__ tile_base,
                                                           const uint16_t *__restrict__ rank, int32_t *__restrict__ perm,
                                                           int64_t max_tiles)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const int32_t *b = tile_base + ((int64_t)f * max_tiles + blockIdx.x) * 256;
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * SG_BLOCK + threadIdx.x;
        if (r < n) {
            const T c = rows[(base + r) * 5 + 4];
            int ci = (int)c;
            if (!((T)ci == c && ci >= 0 && ci < 256)) ci = 255;
            perm[base + b[ci] + rank[base + r]] = (int32_t)r;
        }
    }
}

// ---- frame-level epilogue of one row (simulation.py:516): rounded intensity, label ------------------------------
template <typename T>
__device__ __forceinline__ void sg_store_row(const SgBeamArgs &a, int64_t g, int f, T px, T py, T pz, const SgBeamOut &o)
{
    T *orow = (T *)a.tmp_rows + g * 5;
    T oi;
    if constexpr (SgReal<T>::is_f32) {
        orow[0] = (float)o.x; orow[1] = (float)o.y; orow[2] = (float)o.z;   // :178-180 store float64 -> float32
        oi = rintf((float)o.intensity);                                     // :516 np.round (half to even)
    } else {
        orow[0] = o.x; orow[1] = o.y; orow[2] = o.z;
        oi = rint(o.intensity);
    }
    orow[3] = oi;
    orow[4] = (T)o.label;
    (void)f; (void)px; (void)py; (void)pz;
    // The noise-floor decision (:518-520) is taken by k_compact_count: rows that are not scattered keep their
    // coordinates, so the original range is still in the row -- and the per-beam kernels do not have to wait for the
    // threshold polynomial of the prepass.
}

// ------------------------------------------------------------------------------------------------
// The per-beam kernel.  Dynamic LDS: four per-thread lists of LMAX + 1 float64 entries, strided by the block size.
//   LIST = false  direct mode, the first pass over all rows: phases 1-2 (scan, occlusion dict); beams that met a
//                 flake are handed to k_power through their dict and a flag byte, overflowed beams are flagged
//   LIST = true   a later capacity tier over a device-side list: phases 1-3 in place
template <typename T, int LMAX, int BLOCK, bool LIST>
__global__ __launch_bounds__(BLOCK) void k_beams(SgBeamArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const double *s_rgrid = a.rgrid;                  // 9.8 KB, read with consecutive bins per lane: L1-resident
    double *s_a1 = (double *)smem;                    // LMAX + 1 rows each: the hard target is entry n_flakes <= LMAX
    double *s_a2 = s_a1 + (LMAX + 1) * BLOCK;
    double *s_rho = s_a2 + (LMAX + 1) * BLOCK;
    double *s_ratio = s_rho + (LMAX + 1) * BLOCK;
    const int tid = threadIdx.x;
    // Direct mode: one block per 256 consecutive channel-sorted rows.  List mode (a later capacity tier): the
    // number of queued beams is only known on the device, so a small fixed grid strides over the queue -- a block
    // with this much LDS cannot share its CU, and thousands of empty ones would each cost a dispatch slot.
    int64_t work_n = a.n_total;
    if (LIST) {
        work_n = *a.work_count;
        if (work_n > a.work_cap) work_n = a.work_cap;
    }
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    int seg_f = -1;                                   // segment-ordered direct mode: the block's frame
    int64_t seg_g = -1;
    if (!LIST && a.seg_blk) {
        const int blk = (int)blockIdx.x;
        if (blk >= a.seg_n[1]) return;                // surplus block (the grid is an upper bound)
        const int lo = a.seg_of_blk[blk];             // one load instead of a 13-step dependent search per block
        const int off = (blk - a.seg_blk[lo]) * BLOCK + tid;
        seg_f = a.seg_frame[lo];
        if (off < a.seg_cnt[lo]) seg_g = a.seg_start[lo] + off;
    }
    int64_t chunk = (int64_t)blockIdx.x * BLOCK;
    if (LIST && chunk >= work_n) return;
    do {                                              // direct mode: exactly one trip, and the compiler must see that
    int64_t g = -1;
    if (LIST) {
        if (chunk + tid < work_n) g = a.work_list[chunk + tid];
    } else if (seg_f >= 0) {
        g = seg_g;
    } else {
        g = chunk + tid;
        if (g >= a.n_total) g = -1;
    }
    unsigned long long *ph = a.phase_cycles ? a.phase_cycles + 8 * (LMAX == 4 ? 0 : LMAX == 8 ? 1 : LMAX == 16 ? 2 : 3) : nullptr;
    const unsigned long long tcs = ph ? wall_clock64() : 0;
    // No early return from here on: wave-wide ballots / shuffles follow (overflow queue, per-frame sums).
    const bool live = g >= 0;
    int f = 0;
    T px = 0, py = 0, pz = 0, pint = 0, pch = 0;
    int ch = 0;
    bool simulated = false;
    const int n_las = a.las->n;
    if (live) {
        if (!LIST && seg_f >= 0) f = seg_f;
        else if (a.uniform_rows > 0) {               // equal-sized frames: no search (float estimate, integer fix-up)
            const unsigned rows_u = (unsigned)a.uniform_rows, gu = (unsigned)g;
            int fe = (int)((float)gu * a.inv_uniform_rows);
            if (fe >= a.n_frames) fe = a.n_frames - 1;
            while (fe > 0 && gu < (unsigned)fe * rows_u) --fe;
            while (fe + 1 < a.n_frames && gu >= (unsigned)(fe + 1) * rows_u) ++fe;
            f = fe;
        } else f = sg_find_frame(a.frame_off, a.n_frames, g);
        const int64_t src = a.frame_off[f] + a.perm[g];
        const T *row = (const T *)a.rows + src * 5;
        px = row[0]; py = row[1]; pz = row[2]; pint = row[3]; pch = row[4];
        ch = (int)pch;
        simulated = ((T)ch == pch) && ch >= 0 && ch < n_las;            // simulation.py:80, :482 (Q5)
    }
    const unsigned long long tc0 = ph ? wall_clock64() : 0;
    SgBeamOut o;
    o.x = (double)px; o.y = (double)py; o.z = (double)pz; o.intensity = (double)pint; o.label = (double)pch;
    o.overflow = 0; o.range_error = 0; o.diff2 = 0.0; o.has_power = 0; o.n_flakes = 0;
    bool write_row = live;
    if (simulated) {
        const SgTable tab = a.frame_tables[(int64_t)f * n_las + ch];   // resolved per (frame, channel) by k_resolve_tables
        if (tab.entries == nullptr) {
            atomicCAS(&a.status[0], 0, 1 /* SNOWGPU_E_INVALID */);
            write_row = false;
        } else {
            int32_t *dc = a.dbg_count ? a.dbg_count + g : nullptr;
            double *drj = a.dbg_count ? a.dbg_rj + g * a.dbg_cap : nullptr;
            double *dra = a.dbg_count ? a.dbg_ratio + g * a.dbg_cap : nullptr;
            sg_beam<T, LMAX, BLOCK, !LIST>(px, py, pz, pint, ch, tab, a.las, s_rgrid, a.beam_div_deg, s_a1, s_a2, s_rho,
                                           s_ratio, tid, o, a.dbg_cap, dc, drj, dra, ph, a.exact_math != 0);
            if (o.overflow) {
                write_row = false;                        // a later pass with a longer list writes this row
                o.has_power = 0;
            } else if (o.range_error) {
                atomicCAS(&a.status[0], 0, 4 /* SNOWGPU_E_RANGE */);
                atomicCAS(&a.status[1], -1, (int32_t)g);
            }
        }
    }
    if (!LIST && LMAX < SG_LCAP) {
        // Direct mode: an overflowed beam is only flagged (keep[g] = 2); k_list_* then build the next pass's list in
        // sorted-row order, so that the lanes of its waves stay neighbours in channel and azimuth -- one table, nearby
        // bins.  (An atomic queue hands a wave 64 beams of as many channels, i.e. tables: every load a miss.)
        if (o.overflow) a.keep[g] = 2;
    } else {   // queue the overflowed beams of this wave with ONE atomic (a per-lane atomic on a single counter
        // serialises in L2 and stalls every other memory request behind it)
        const unsigned long long om = __ballot(o.overflow != 0);
        if (om) {
            if (LMAX >= SG_LCAP) {
                if (o.overflow) {
                    atomicCAS(&a.status[0], 0, 6 /* SNOWGPU_E_OVERFLOW */);
                    atomicCAS(&a.status[1], -1, (int32_t)g);
                }
            } else {
                int base = 0;
                const int leader = __ffsll((long long)om) - 1;
                if ((tid & 63) == leader) base = atomicAdd(a.ovf_count, (int)__popcll(om));
                base = __shfl(base, leader);
                if (o.overflow) {
                    const int slot = base + (int)__popcll(om & sg_lanemask_lt());
                    if (slot < a.ovf_cap) a.ovf_list[slot] = (int32_t)g;
                }
            }
        }
    }
    const unsigned long long tc1 = ph ? wall_clock64() : 0;
    double best = 0.0;
    int k_best = 0;
    if constexpr (!LIST) {
        // ---- direct mode: hand the beams that met a flake, with their occlusion dicts, to k_power ----------------
        // Only a fraction of the beams gets here; walking phase 3 in place would keep most lanes of every wave idle.
        // No queue counter: the beam is flagged (keep[g] = 16 + n_flakes), its dict goes to the slot of its own sorted
        // position, and k_list_* build the list in sorted-row order -- no atomics, and the lanes of a k_power wave stay
        // neighbours (one frame, one channel).
        if (o.has_power) {
            write_row = false;                                // k_power writes this row
            a.keep[g] = (uint8_t)(16 + o.n_flakes);
            double *dd = a.pq_dict + g * a.pq_stride;
            for (int t = 0; t <= o.n_flakes; ++t) { dd[2 * t] = s_rho[t * BLOCK + tid]; dd[2 * t + 1] = s_ratio[t * BLOCK + tid]; }
            o.has_power = 0;
        }
    } else {
    // ---- phase 3b: received power (per lane; s_ratio is dead after phase 3a and carries the work lists) ----
    constexpr int NB = LMAX <= 4 ? 4 : 8;    // bins carried together
    if (o.has_power) {
        int st[2] = {0, 0};
        if (a.exact_math) sg_lane_power<BLOCK, true, NB, LMAX>(o.n_flakes, s_rgrid, s_a1, s_a2, s_rho, s_ratio, tid, best, k_best);
        else sg_lane_power<BLOCK, false, NB, LMAX>(o.n_flakes, s_rgrid, s_a1, s_a2, s_rho, s_ratio, tid, best, k_best, ph ? st : nullptr);
        if (ph) {                                         // experiment: trip counts of the two stages, per wave
            int mx[2], sm[2];
            for (int q = 0; q < 2; ++q) {
                mx[q] = st[q]; sm[q] = st[q];
                for (int off = 32; off > 0; off >>= 1) { mx[q] = max(mx[q], __shfl_xor(mx[q], off)); sm[q] += __shfl_xor(sm[q], off); }
            }
            const unsigned long long am = __ballot(1);
            if ((tid & 63) == __ffsll((long long)am) - 1) {
                unsigned long long *p2 = a.phase_cycles + 32 + 8 * (LMAX == 4 ? 0 : LMAX == 8 ? 1 : LMAX == 16 ? 2 : 3);
                atomicAdd(&p2[0], 1ull); atomicAdd(&p2[1], (unsigned long long)__popcll(am));
                atomicAdd(&p2[2], (unsigned long long)mx[0]); atomicAdd(&p2[3], (unsigned long long)mx[1]);
                atomicAdd(&p2[4], (unsigned long long)sm[0]); atomicAdd(&p2[5], (unsigned long long)sm[1]);
            }
        }
    }
    }
    if (ph && (tid & 63) == 0) {
        atomicAdd(&ph[0], tc0 - tcs);                  // frame lookup + row load
        atomicAdd(&ph[4], wall_clock64() - tc1);       // phase 3b (list mode) / queueing (direct mode)
        atomicAdd(&ph[5], 1ull);                       // waves
    }
    if (o.has_power) sg_beam_decide<T>(px, py, pz, ch, a.las, best, k_best, o);
    {   // intensity_diff_sum (simulation.py:170, :512): one atomic per wave and frame, not one per beam
        long long d2 = write_row ? (long long)o.diff2 : 0;
        const int f0 = __shfl(f, 0);
        const bool same = __all(!live || f == f0);
        if (same) {
            for (int off = 32; off > 0; off >>= 1) d2 += __shfl_down(d2, off);
            if ((tid & 63) == 0 && d2 != 0) atomicAdd(&a.diff2[f0], (unsigned long long)d2);
        } else if (d2 != 0) {
            atomicAdd(&a.diff2[f], (unsigned long long)d2);      // wave straddling two frames
        }
    }
    if (!write_row) continue;
    sg_store_row<T>(a, g, f, px, py, pz, o);
    } while (LIST && (chunk += stride) < work_n);
}

// ------------------------------------------------------------------------------------------------
// Received power for the beams the direct-mode pass queued: one thread per queue slot, every lane busy.
// Phase 3a (amplitudes, windows), 3b (pruned power profile and its first maximum), 3c (decision), row epilogue.
template <typename T, int LMAX, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_power(SgBeamArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *s_a1 = (double *)smem;
    double *s_a2 = s_a1 + (LMAX + 1) * BLOCK;
    double *s_rho = s_a2 + (LMAX + 1) * BLOCK;
    double *s_ratio = s_rho + (LMAX + 1) * BLOCK;
    const int tid = threadIdx.x;
    int n = *a.pq_count;
    if (n > a.pq_cap) n = a.pq_cap;
    const int64_t slot = (int64_t)blockIdx.x * BLOCK + tid;
    if ((int64_t)blockIdx.x * BLOCK >= n) return;     // surplus block (the grid covers every row)
    const bool live = slot < n;
    int64_t g = 0;
    int f = 0, S = 0, ch = 0;
    T px = 0, py = 0, pz = 0, pch = 0;
    if (live) {
        g = a.pq_list[slot];
        S = (int)a.keep[g] - 16;
        if (a.uniform_rows > 0) {                    // equal-sized frames: no search (float estimate, integer fix-up)
            const unsigned rows_u = (unsigned)a.uniform_rows, gu = (unsigned)g;
            int fe = (int)((float)gu * a.inv_uniform_rows);
            if (fe >= a.n_frames) fe = a.n_frames - 1;
            while (fe > 0 && gu < (unsigned)fe * rows_u) --fe;
            while (fe + 1 < a.n_frames && gu >= (unsigned)(fe + 1) * rows_u) ++fe;
            f = fe;
        } else f = sg_find_frame(a.frame_off, a.n_frames, g);
        const int64_t src = a.frame_off[f] + a.perm[g];
        const T *row = (const T *)a.rows + src * 5;
        px = row[0]; py = row[1]; pz = row[2]; pch = row[4];
        ch = (int)pch;
        const double *dd = a.pq_dict + g * a.pq_stride;
        for (int t = 0; t <= S; ++t) { s_rho[t * BLOCK + tid] = dd[2 * t]; s_ratio[t * BLOCK + tid] = dd[2 * t + 1]; }
    }
    SgBeamOut o;
    o.x = (double)px; o.y = (double)py; o.z = (double)pz; o.intensity = 0.0; o.label = 0.0;
    o.overflow = 0; o.range_error = 0; o.diff2 = 0.0; o.has_power = 0; o.n_flakes = S;
    double best = 0.0;
    int k_best = 0;
    if (live) {
        T d_t;
        if constexpr (SgReal<T>::is_f32) d_t = sqrtf((px * px + py * py) + pz * pz);    // simulation.py:89
        else d_t = sqrt((px * px + py * py) + pz * pz);
        sg_beam_amp<T, LMAX, BLOCK>(d_t, S, ch, a.las, s_a1, s_a2, s_rho, s_ratio, tid, o);
        if (o.range_error) {
            atomicCAS(&a.status[0], 0, 4 /* SNOWGPU_E_RANGE */);
            atomicCAS(&a.status[1], -1, (int32_t)g);
        }
        constexpr int NB = LMAX <= 4 ? 4 : 8;
        if (a.exact_math) sg_lane_power<BLOCK, true, NB, LMAX>(S, a.rgrid, s_a1, s_a2, s_rho, s_ratio, tid, best, k_best);
        else sg_lane_power<BLOCK, false, NB, LMAX>(S, a.rgrid, s_a1, s_a2, s_rho, s_ratio, tid, best, k_best);
        sg_beam_decide<T>(px, py, pz, ch, a.las, best, k_best, o);
    }
    {   // intensity_diff_sum (simulation.py:170, :512): one atomic per wave and frame (same-address atomics from every
        // lane would serialise in L2); the list is in sorted-row order, so a wave rarely holds more than one frame
        long long d2 = live ? (long long)o.diff2 : 0;
        unsigned long long todo = __ballot(live && d2 != 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int fl = __shfl(f, leader);
            const bool mine = live && f == fl;
            long long part = mine ? d2 : 0;
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if ((tid & 63) == leader && part != 0) atomicAdd(&a.diff2[fl], (unsigned long long)part);
            todo &= ~__ballot(mine);
        }
    }
    if (live) sg_store_row<T>(a, g, f, px, py, pz, o);
}

// ------------------------------------------------------------------------------------------------
// Segment order of the first pass.  A segment = the rows of one (frame, channel) pair in the channel-sorted order,
// i.e. all beams of a frame that look up the same flake table.  Segments are ordered by table, so that the ~1000
// blocks resident at any moment use one or two tables (2-3 MB each: L2-resident) instead of all 64 of a frame.
// Three small kernels over the n_frames * 256 pairs: count segments and blocks per table (one packed 64-bit atomic
// per pair: segments << 32 | blocks), exclusive scan over the tables, place every pair (a second packed atomic gives
// its segment slot and its first block inside the table's range).  The order inside a table is whatever the atomics
// give -- results do not depend on the launch order.
struct SgPair { int64_t start; int rows; int key; };

__device__ __forceinline__ SgPair sg_pair(int p, const int64_t *__restrict__ frame_off, const int32_t *__restrict__ tile_base, int64_t max_tiles,
                                          const int32_t *__restrict__ table_ids, int n_las, int n_tables)
{
    SgPair r;
    const int f = p >> 8, c = p & 255;
    const int64_t n = frame_off[f + 1] - frame_off[f];
    r.start = frame_off[f]; r.rows = 0; r.key = n_tables;
    if (n <= 0) return r;                             // the sort wrote nothing for an empty frame
    const int32_t *b = tile_base + (int64_t)f * max_tiles * 256;
    const int64_t s0 = b[c], s1 = c < 255 ? (int64_t)b[c + 1] : n;
    r.start += s0;
    r.rows = (int)(s1 - s0);
    if (c < n_las) {
        const int id = table_ids[(int64_t)f * n_las + c];
        if (id >= 0 && id < n_tables) r.key = id;     // unknown ids and channels without a laser go last
    }
    return r;
}

__global__ __launch_bounds__(256) void k_seg_count(const int64_t *__restrict__ frame_off, int n_frames, const int32_t *__restrict__ tile_base,
                                                   int64_t max_tiles, const int32_t *__restrict__ table_ids, int n_las, int n_tables, int blk,
                                                   unsigned long long *__restrict__ tbl_cnt)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_frames * 256) return;
    const SgPair r = sg_pair(p, frame_off, tile_base, max_tiles, table_ids, n_las, n_tables);
    if (r.rows > 0) atomicAdd(&tbl_cnt[r.key], (1ull << 32) | (unsigned long long)((r.rows + blk - 1) / blk));
}

// exclusive scan of the packed per-table counts (both halves at once: neither overflows 32 bits); leaves the counts zero
// so that k_seg_place can use them as cursors
__global__ __launch_bounds__(1024) void k_seg_scan(unsigned long long *__restrict__ tbl_cnt, unsigned long long *__restrict__ tbl_base, int n,
                                                   int32_t *__restrict__ seg_n)
{
    __shared__ unsigned long long sc[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024, b0 = t * per, b1 = b0 + per < n ? b0 + per : n;
    unsigned long long sum = 0;
    for (int k = b0; k < b1; ++k) sum += tbl_cnt[k];
    sc[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const unsigned long long add = t >= d ? sc[t - d] : 0; __syncthreads(); sc[t] += add; __syncthreads(); }
    unsigned long long run = sc[t] - sum;
    for (int k = b0; k < b1; ++k) { const unsigned long long c = tbl_cnt[k]; tbl_base[k] = run; run += c; tbl_cnt[k] = 0; }
    if (t == 1023) { seg_n[0] = (int32_t)(sc[1023] >> 32); seg_n[1] = (int32_t)(sc[1023] & 0xffffffffull); }
}

__global__ __launch_bounds__(256) void k_seg_place(const int64_t *__restrict__ frame_off, int n_frames, const int32_t *__restrict__ tile_base,
                                                   int64_t max_tiles, const int32_t *__restrict__ table_ids, int n_las, int n_tables, int blk,
                                                   const unsigned long long *__restrict__ tbl_base, unsigned long long *__restrict__ tbl_cur,
                                                   int64_t *__restrict__ seg_start, int32_t *__restrict__ seg_cnt, int32_t *__restrict__ seg_frame,
                                                   int32_t *__restrict__ seg_blk, int32_t *__restrict__ seg_of_blk)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_frames * 256) return;
    const SgPair r = sg_pair(p, frame_off, tile_base, max_tiles, table_ids, n_las, n_tables);
    if (r.rows <= 0) return;
    const int nb = (r.rows + blk - 1) / blk;
    const unsigned long long c = atomicAdd(&tbl_cur[r.key], (1ull << 32) | (unsigned long long)nb), base = tbl_base[r.key];
    const int slot = (int)(base >> 32) + (int)(c >> 32);
    const int b0 = (int)(base & 0xffffffffull) + (int)(c & 0xffffffffull);
    seg_start[slot] = r.start; seg_cnt[slot] = r.rows; seg_frame[slot] = p >> 8; seg_blk[slot] = b0;
    for (int q = 0; q < nb; ++q) seg_of_blk[b0 + q] = slot;      // block -> segment: one load per block in k_beams
}

// ------------------------------------------------------------------------------------------------
// Ordered lists of the first pass, built from the flag bytes: positions g with keep[g] in class 1 = [lo1, hi1], ascending,
// followed by those in class 2 = [lo2, hi2], ascending (lo2 > hi2: no second class).  keep[g] == 2: overflowed beams
// for the next capacity tier; 16 + n_flakes: beams with an occlusion dict for k_power (one flake first, then the rest).
// Three kernels over tiles of SG_TILE positions: count, one-block scan, scatter.  (Folding the scan into the last
// block of the count kernel was tried: the device-scope fence it needs writes back the L2 of the block's XCD, and
// 16 000 of those cost more than the launch they save.)
__global__ __launch_bounds__(SG_BLOCK) void k_list_count(const uint8_t *__restrict__ keep, int64_t n_total, int32_t *__restrict__ tile_cnt,
                                                         int lo1, int hi1, int lo2, int hi2)
{
    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * SG_TILE + (int64_t)tid * 4;
    int c1 = 0, c2 = 0;
    uint32_t v = 0;
    if (g0 + 3 < n_total) v = *(const uint32_t *)(keep + g0);
    else for (int q = 0; q < 4; ++q) if (g0 + q < n_total) v |= (uint32_t)keep[g0 + q] << (8 * q);
    for (int q = 0; q < 4; ++q) {
        const int b = g0 + q < n_total ? (int)((v >> (8 * q)) & 0xff) : -1;
        c1 += (b >= lo1 && b <= hi1);
        c2 += (b >= lo2 && b <= hi2);
    }
    __shared__ int s1[SG_BLOCK / 64], s2[SG_BLOCK / 64];
    for (int o = 32; o > 0; o >>= 1) { c1 += __shfl_down(c1, o); c2 += __shfl_down(c2, o); }
    if ((tid & 63) == 0) { s1[tid >> 6] = c1; s2[tid >> 6] = c2; }
    __syncthreads();
    if (tid == 0) {
        int t1 = 0, t2 = 0;
        for (int w = 0; w < SG_BLOCK / 64; ++w) { t1 += s1[w]; t2 += s2[w]; }
        tile_cnt[2 * blockIdx.x] = t1; tile_cnt[2 * blockIdx.x + 1] = t2;
    }
}

__global__ __launch_bounds__(1024) void k_list_scan(const int32_t *__restrict__ tile_cnt, int32_t *__restrict__ tile_base, int tiles,
                                                    int32_t *__restrict__ count_out)
{
    __shared__ int s1[1024], s2[1024];
    const int tid = threadIdx.x;
    const int per = (tiles + 1023) / 1024, b0 = tid * per, b1 = b0 + per < tiles ? b0 + per : tiles;
    int sum1 = 0, sum2 = 0;
    for (int i = b0; i < b1; ++i) { sum1 += tile_cnt[2 * i]; sum2 += tile_cnt[2 * i + 1]; }
    s1[tid] = sum1; s2[tid] = sum2;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int a1 = tid >= d ? s1[tid - d] : 0, a2 = tid >= d ? s2[tid - d] : 0;
        __syncthreads();
        s1[tid] += a1; s2[tid] += a2;
        __syncthreads();
    }
    const int total1 = s1[1023], total2 = s2[1023];
    int run1 = s1[tid] - sum1, run2 = total1 + s2[tid] - sum2;      // class 2 follows class 1 in the list
    for (int i = b0; i < b1; ++i) {
        tile_base[2 * i] = run1; tile_base[2 * i + 1] = run2;
        run1 += tile_cnt[2 * i]; run2 += tile_cnt[2 * i + 1];
    }
    if (tid == 0) *count_out = total1 + total2;
}

__global__ __launch_bounds__(SG_BLOCK) void k_list_scatter(const uint8_t *__restrict__ keep, int64_t n_total, const int32_t *__restrict__ tile_base,
                                                           int32_t *__restrict__ list, int32_t cap, int lo1, int hi1, int lo2, int hi2)
{
    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * SG_TILE + (int64_t)tid * 4;
    int cls[4];
    int c1 = 0, c2 = 0;
    for (int q = 0; q < 4; ++q) {
        const int b = g0 + q < n_total ? (int)keep[g0 + q] : -1;
        cls[q] = (b >= lo1 && b <= hi1) ? 1 : ((b >= lo2 && b <= hi2) ? 2 : 0);
        c1 += cls[q] == 1; c2 += cls[q] == 2;
    }
    // exclusive prefixes of c1, c2 over the block: wave scan + wave totals
    int i1 = c1, i2 = c2;
    for (int o = 1; o < 64; o <<= 1) {
        const int v1 = __shfl_up(i1, o), v2 = __shfl_up(i2, o);
        if ((tid & 63) >= o) { i1 += v1; i2 += v2; }
    }
    __shared__ int s1[SG_BLOCK / 64], s2[SG_BLOCK / 64];
    if ((tid & 63) == 63) { s1[tid >> 6] = i1; s2[tid >> 6] = i2; }
    __syncthreads();
    int slot1 = tile_base[2 * blockIdx.x] + i1 - c1, slot2 = tile_base[2 * blockIdx.x + 1] + i2 - c2;
    for (int w = 0; w < (tid >> 6); ++w) { slot1 += s1[w]; slot2 += s2[w]; }
    for (int q = 0; q < 4; ++q) {
        if (cls[q] == 1) { if (slot1 < cap) list[slot1] = (int32_t)(g0 + q); ++slot1; }
        else if (cls[q] == 2) { if (slot2 < cap) list[slot2] = (int32_t)(g0 + q); ++slot2; }
    }
}

// ------------------------------------------------------------------------------------------------
// Stable compaction of kept rows, per frame.
template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_compact_count(const T *__restrict__ tmp_rows, const double *__restrict__ thr_poly,
                                                            uint8_t *__restrict__ keep, const int64_t *__restrict__ frame_off,
                                                            int32_t *__restrict__ tile_cnt, int64_t max_tiles)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const double p0 = thr_poly[(int64_t)f * 3], p1 = thr_poly[(int64_t)f * 3 + 1], p2 = thr_poly[(int64_t)f * 3 + 2];
    int c = 0;
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * SG_BLOCK + threadIdx.x;
        if (r >= n) continue;
        // keep = (label == 2) | (intensity > p0 d^2 + p1 d + p2), d the ORIGINAL range, d^2 in the row dtype
        // (simulation.py:465, :469, :518-520); rows with label != 2 still hold their original coordinates
        const T *row = tmp_rows + (base + r) * 5;
        const T x = row[0], y = row[1], z = row[2], oi = row[3], lab = row[4];
        T dd;
        if constexpr (sizeof(T) == 4) dd = sqrtf((x * x + y * y) + z * z);
        else dd = sqrt((x * x + y * y) + z * z);
        const T dd2 = dd * dd;
        const double thr = (p0 * (double)dd2 + p1 * (double)dd) + p2;
        const bool k = (lab == (T)2) || ((double)oi > thr);
        keep[base + r] = k ? 1 : 0;
        c += k;
    }
    __shared__ int s[4];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[(int64_t)f * max_tiles + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(SG_BLOCK) void k_compact_scan(const int64_t *__restrict__ frame_off,
                                                           const int32_t *__restrict__ tile_cnt,
                                                           int32_t *__restrict__ tile_base, int64_t *__restrict__ out_counts,
                                                           int64_t *__restrict__ out_stats, int64_t max_tiles)
{
    const int f = blockIdx.x;
    const int64_t n = frame_off[f + 1] - frame_off[f];
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    if (threadIdx.x == 0) {                      // <= a few hundred tiles per frame: a serial scan is fine
        int run = 0;
        for (int64_t t = 0; t < tiles; ++t) {
            tile_base[(int64_t)f * max_tiles + t] = run;
            run += tile_cnt[(int64_t)f * max_tiles + t];
        }
        out_counts[f] = run;
        out_stats[f * 3 + 0] = 0;                // num_attenuated: filled by k_compact_scatter
        out_stats[f * 3 + 1] = n - run;          // num_removed (simulation.py:522)
        out_stats[f * 3 + 2] = 0;
    }
}

template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_compact_scatter(const T *__restrict__ tmp_rows, const uint8_t *__restrict__ keep,
                                                              const int32_t *__restrict__ perm,
                                                              const int64_t *__restrict__ frame_off,
                                                              const int32_t *__restrict__ tile_base, T *__restrict__ out_rows,
                                                              int32_t *__restrict__ out_src, int64_t *__restrict__ out_stats,
                                                              int64_t max_tiles)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    __shared__ int wave_cnt[4][4];               // [round][wave]
    const int tid = threadIdx.x, w = tid >> 6;
    bool k[4];
    int pre[4];
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * SG_BLOCK + tid;
        k[q] = (r < n) && keep[base + r];
        const unsigned long long m = __ballot(k[q]);
        pre[q] = __popcll(m & sg_lanemask_lt());
        if ((tid & 63) == 0) wave_cnt[q][w] = __popcll(m);
    }
    __syncthreads();
    int run = tile_base[(int64_t)f * max_tiles + blockIdx.x];
    int att = 0;
    for (int q = 0; q < 4; ++q) {
        int off = run;
        for (int ww = 0; ww < w; ++ww) off += wave_cnt[q][ww];
        if (k[q]) {
            const int64_t r = base + tile0 + q * SG_BLOCK + tid;
            const int64_t dst = base + off + pre[q];
            const T *s = tmp_rows + r * 5;
            T *d = out_rows + dst * 5;
            const T lab = s[4];
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; d[4] = lab;
            out_src[dst] = perm[r];
            if (lab == (T)1) ++att;              // simulation.py:525
        }
        run += wave_cnt[q][0] + wave_cnt[q][1] + wave_cnt[q][2] + wave_cnt[q][3];
    }
    for (int o = 32; o > 0; o >>= 1) att += __shfl_down(att, o);
    if ((tid & 63) == 0 && att) atomicAdd((unsigned long long *)&out_stats[f * 3 + 0], (unsigned long long)att);
}

__global__ void k_stats_final(int n_frames, int64_t *__restrict__ out_stats, const unsigned long long *__restrict__ diff2)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const int64_t att = out_stats[f * 3 + 0];
    const double diff_sum = (double)(long long)diff2[f] / 2.0;
    out_stats[f * 3 + 2] = att > 0 ? (int64_t)(diff_sum / (double)att) : 0;   // simulation.py:527-530 int()
}

// table_ids[frame][channel] -> the table descriptor itself, so that a beam needs one load instead of two dependent ones
__global__ void k_resolve_tables(const SgTable *__restrict__ tables, int n_tables, const int32_t *__restrict__ table_ids,
                                 int64_t n, SgTable *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = table_ids[i];
    SgTable d{};
    if (t >= 0 && t < n_tables) d = tables[t];
    out[i] = d;
}

extern "C" int sg_launch_resolve_tables(const SgTable *tables, int n_tables, const int32_t *table_ids, int64_t n, SgTable *out,
                                        void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_resolve_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tables, n_tables,
                       table_ids, n, out);
    return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// launch wrappers (C linkage, called from snowgpu_api.cpp)

#define SG_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return (int)e__;            \
    } while (0)

extern "C" int sg_launch_sort(const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t n_total,
                              int32_t *tile_hist, int32_t *tile_base, uint16_t *rank, int32_t *perm, int32_t *status,
                              int64_t max_tiles, void *stream)
{
    (void)n_total;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_sort_hist<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)rows, frame_off, tile_hist, rank, status, max_tiles);
    else hipLaunchKernelGGL(k_sort_hist<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)rows, frame_off, tile_hist, rank, status, max_tiles);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sort_scan, dim3(n_frames), dim3(SG_BLOCK), 0, st, frame_off, tile_hist, tile_base, max_tiles);
    SG_CHECK_LAUNCH();
    if (dtype == 0) hipLaunchKernelGGL(k_sort_scatter<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)rows, frame_off, tile_base, rank, perm, max_tiles);
    else hipLaunchKernelGGL(k_sort_scatter<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)rows, frame_off, tile_base, rank, perm, max_tiles);
    SG_CHECK_LAUNCH();
    return 0;
}

template <typename T, int LMAX, int BLOCK, bool LIST>
static int launch_beams_tl(const SgBeamArgs *a, int64_t n_threads, hipStream_t st)
{
    const size_t lds = sizeof(double) * ((size_t)BLOCK * 4 * (LMAX + 1));
    static bool attr_set[64] = {};                       // per device: several contexts may live in one process
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        hipError_t e = hipFuncSetAttribute((const void *)k_beams<T, LMAX, BLOCK, LIST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    unsigned blocks = (unsigned)((n_threads + BLOCK - 1) / BLOCK);
    if (!LIST && a->seg_blk) blocks = (unsigned)a->grid_blocks;
    if (blocks == 0) return 0;
    if (LIST) {                                          // list mode: at most what the chip can hold at once
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
        const unsigned per_cu = (unsigned)std::max<size_t>(1, (size_t)(160 * 1024) / lds);
        blocks = std::min(blocks, (unsigned)cus * per_cu);
    }
    hipLaunchKernelGGL((k_beams<T, LMAX, BLOCK, LIST>), dim3(blocks), dim3(BLOCK), lds, st, *a);
    SG_CHECK_LAUNCH();
    return 0;
}

template <typename T, int LMAX, int BLOCK>
static int launch_beams_t(const SgBeamArgs *a, int64_t n_threads, hipStream_t st)
{
    return a->work_list ? launch_beams_tl<T, LMAX, BLOCK, true>(a, n_threads, st)
                        : launch_beams_tl<T, LMAX, BLOCK, false>(a, n_threads, st);
}

template <typename T, int LMAX, int BLOCK>
static int launch_power_t(const SgBeamArgs *a, hipStream_t st)
{
    const size_t lds = sizeof(double) * ((size_t)BLOCK * 4 * (LMAX + 1));
    static bool attr_set[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
        hipError_t e = hipFuncSetAttribute((const void *)k_power<T, LMAX, BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    const unsigned blocks = (unsigned)((a->n_total + BLOCK - 1) / BLOCK);     // upper bound: surplus blocks leave at once
    if (blocks == 0) return 0;
    hipLaunchKernelGGL((k_power<T, LMAX, BLOCK>), dim3(blocks), dim3(BLOCK), lds, st, *a);
    SG_CHECK_LAUNCH();
    return 0;
}

// the received-power kernel for the queue a direct-mode pass of capacity lmax filled
extern "C" int sg_launch_power(const SgBeamArgs *a, int dtype, int lmax, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        if (lmax == 4) return launch_power_t<float, 4, 256>(a, st);
        if (lmax == 8) return launch_power_t<float, 8, 64>(a, st);
        if (lmax == 16) return launch_power_t<float, 16, 64>(a, st);
        return launch_power_t<float, SG_LCAP, 64>(a, st);
    }
    if (lmax == 4) return launch_power_t<double, 4, 256>(a, st);
    if (lmax == 8) return launch_power_t<double, 8, 64>(a, st);
    if (lmax == 16) return launch_power_t<double, 16, 64>(a, st);
    return launch_power_t<double, SG_LCAP, 64>(a, st);
}

// threads per block of the pass with list capacity lmax (the segment builder counts blocks of this size)
extern "C" int sg_beams_block(int lmax) { return lmax == 4 ? 256 : (lmax == 32 ? 128 : 64); }

// lmax = per-beam list capacity of this pass: 4 (160 B of LDS per beam: 16 waves per CU), 8, 16 or 63 (the hard
// cap).  Beams that exceed it are flagged (direct mode) or queued (list mode) for the next pass.  With a->work_list
// set, a chip-sized grid strides over the list, whose length is only known on the device.
extern "C" int sg_launch_beams(const SgBeamArgs *a, int dtype, int lmax, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = a->work_list ? (int64_t)a->work_cap : a->n_total;
    if (dtype == 0) {
        if (lmax == 4) return launch_beams_t<float, 4, 256>(a, n, st);
        if (lmax == 8) return launch_beams_t<float, 8, 64>(a, n, st);
        if (lmax == 16) return launch_beams_t<float, 16, 64>(a, n, st);
        if (lmax == 32) return launch_beams_t<float, 32, 128>(a, n, st);
        return launch_beams_t<float, SG_LCAP, 64>(a, n, st);
    } else {
        if (lmax == 4) return launch_beams_t<double, 4, 256>(a, n, st);
        if (lmax == 8) return launch_beams_t<double, 8, 64>(a, n, st);
        if (lmax == 16) return launch_beams_t<double, 16, 64>(a, n, st);
        if (lmax == 32) return launch_beams_t<double, 32, 128>(a, n, st);
        return launch_beams_t<double, SG_LCAP, 64>(a, n, st);
    }
}

extern "C" int sg_launch_segments(const int64_t *frame_off, int n_frames, const int32_t *tile_base, int64_t max_tiles, const int32_t *table_ids,
                                  int n_las, int n_tables, int block, unsigned long long *tbl_cnt, unsigned long long *tbl_base, int32_t *seg_blk,
                                  int64_t *seg_start, int32_t *seg_cnt, int32_t *seg_frame, int32_t *seg_n, int32_t *seg_of_blk, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)n_frames;         // 256 pairs per frame, one thread each
    if (hipMemsetAsync(tbl_cnt, 0, sizeof(unsigned long long) * ((size_t)n_tables + 1), st) != hipSuccess) return (int)hipGetLastError();
    hipLaunchKernelGGL(k_seg_count, dim3(grid), dim3(256), 0, st, frame_off, n_frames, tile_base, max_tiles, table_ids, n_las, n_tables, block, tbl_cnt);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1024), 0, st, tbl_cnt, tbl_base, n_tables + 1, seg_n);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_seg_place, dim3(grid), dim3(256), 0, st, frame_off, n_frames, tile_base, max_tiles, table_ids, n_las, n_tables, block,
                       tbl_base, tbl_cnt, seg_start, seg_cnt, seg_frame, seg_blk, seg_of_blk);
    SG_CHECK_LAUNCH();
    return 0;
}

extern "C" int sg_launch_list(const uint8_t *keep, int64_t n_total, int32_t *tile_cnt, int32_t *tile_base, int32_t *list, int32_t *count,
                              int32_t cap, int lo1, int hi1, int lo2, int hi2, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int64_t tiles = (n_total + SG_TILE - 1) / SG_TILE;
    if (tiles == 0) return 0;
    hipLaunchKernelGGL(k_list_count, dim3((unsigned)tiles), dim3(SG_BLOCK), 0, st, keep, n_total, tile_cnt, lo1, hi1, lo2, hi2);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_list_scan, dim3(1), dim3(1024), 0, st, tile_cnt, tile_base, (int)tiles, count);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_list_scatter, dim3((unsigned)tiles), dim3(SG_BLOCK), 0, st, keep, n_total, tile_base, list, cap, lo1, hi1, lo2, hi2);
    SG_CHECK_LAUNCH();
    return 0;
}

extern "C" int sg_launch_compact(const void *tmp_rows, int dtype, const double *thr_poly, uint8_t *keep, const int32_t *perm,
                                 const int64_t *frame_off, int n_frames, int64_t n_total, int32_t *tile_cnt,
                                 int32_t *tile_base, void *out_rows, int32_t *out_src, int64_t *out_counts,
                                 int64_t *out_stats, const unsigned long long *diff2, int64_t max_tiles, void *stream)
{
    (void)n_total;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_compact_count<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)tmp_rows, thr_poly, keep, frame_off, tile_cnt, max_tiles);
    else hipLaunchKernelGGL(k_compact_count<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)tmp_rows, thr_poly, keep, frame_off, tile_cnt, max_tiles);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_compact_scan, dim3(n_frames), dim3(SG_BLOCK), 0, st, frame_off, tile_cnt, tile_base, out_counts, out_stats, max_tiles);
    SG_CHECK_LAUNCH();
    if (dtype == 0)
        hipLaunchKernelGGL(k_compact_scatter<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)tmp_rows, keep, perm, frame_off, tile_base, (float *)out_rows, out_src, out_stats, max_tiles);
    else
        hipLaunchKernelGGL(k_compact_scatter<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)tmp_rows, keep, perm, frame_off, tile_base, (double *)out_rows, out_src, out_stats, max_tiles);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_stats_final, dim3((n_frames + 63) / 64), dim3(64), 0, st, n_frames, out_stats, diff2);
    SG_CHECK_LAUNCH();
    return 0;
}
