// snowgpu_kernels.hip -- gfx950 kernels of the snowfall-augmentation engine and their launch wrappers.
//
//   k_sort_*     stable counting sort of every frame's rows by channel                      (simulation.py:447)
//   k_seg_*      launch order of the first pass: (table, frame, channel) segments
//   k_beams      one thread per beam: candidate scan + occlusion dict; beams with flakes hand their dict to k_power
//                through a compact queue, beams with more flakes than the list holds go on the list of the capacity
//                tier that takes them                                                       (simulation.py:50-424)
//   k_tier_scan_direct  the scan of a later tier whose lists the first pass could not keep
//   k_power      received power on the 10 cm grid, first maximum, attenuate-or-scatter        (simulation.py:135-188)
//   k_beams_huge the global-list tier (more than 63 flakes in one beam)
//   k_compact_*  output rows from original rows + 4-byte result records, noise-floor filter, camera-FOV crop, stable
//                stream compaction, stats                                                   (simulation.py:516-540)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see lidar_snow_sim_amd/build.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "sg_beam.h"
#include "sg_few.h"
#include "sg_kutil.h"
#include "sg_lean.h"
#include "sg_row.h"

#define SG_BLOCK 256
#ifndef SG_NB4
#define SG_NB4 4          /* bins of the power profile evaluated together by the 4-entry tier (1, 2, 3 measured slower) ... */
#endif
#ifndef SG_KP_WIN
#define SG_KP_WIN 3       /* k_power: a work item of the multi-flake beams is this many waves' worth of slots, taken in order of flake count */
#endif
#ifndef SG_FP_WAVES
#define SG_FP_WAVES 6     /* waves per SIMD the pass over all rows is compiled for (<= 80 VGPRs; its LDS -- 19 KB per block -- would allow 8) */
#endif
#ifndef SG_NB_TIERS
#define SG_NB_TIERS 4     /* ... and by the later tiers (8 until the dict's endpoints moved into registers; since then, same box, 8 / 4 / 3 / 2:
                             C2 3.96 / 3.90 / 3.93 / 3.93 ms, C2far 7.61 / 7.53 - 7.59 / 7.53 / 7.63, C1 7.21 / 7.16) */
#endif
#ifndef SG_KP_THREE_MAX
#define SG_KP_THREE_MAX 16    /* k_power keeps three list columns in LDS (not four) for capacities SG_KP_THREE_MIN .. SG_KP_THREE_MAX; 0: never */
#endif
#ifndef SG_KP_THREE_MIN
#define SG_KP_THREE_MIN 8     /* (the 4-entry k_power gains no occupancy from it -- it takes half of each CU by design -- and pays for the ranges' second trip: 5.17 vs 4.90 ms per C2 step) */
#endif
#ifndef SG_KP_WAVES_TIERS
#define SG_KP_WAVES_TIERS 2   /* waves per SIMD the 8- and 16-entry k_power are compiled for.  With the dict's endpoints sorted in registers (sg_beam.h:
                                 SG_DICT_SWEEP) the 16-entry kernel wants more than 256 registers: held to 256 -- 18 spilled, 44 B of scratch per lane -- it keeps
                                 its two waves per SIMD and C2far gains 7 %; left alone it drops to one wave and C2far loses 5 %.  (Round 5, before that
                                 dict: 1, 2 and 3 measured the same) */
#endif
#ifndef SG_KP_WAVES
#define SG_KP_WAVES 3     /* waves per SIMD k_power<4> is compiled for.  3 = 168 VGPRs, 10 spilled (40 B of scratch per lane): same box, against 2 (184 VGPRs,
                             none spilled) C2 3.71 -> 3.67 ms, C1 6.92 -> 6.86, C2far the same -- the kernel waits on latency, and a third wave per SIMD is worth
                             more than the spill costs; at 4 (128 VGPRs) it spilled 33 registers and wrote 0.65 GB of scratch per step */
#endif
// Beams per wave of the per-beam kernels by list capacity (the LDS lists are strided by it).  The first capacity fills
// whole 256-thread blocks; 8 and 16 entries run full 64-lane waves; the 63-entry tier runs 16 live lanes per wave (32 KB of
// LDS per block instead of 131 KB: a block that needs most of a CU's LDS waits until one has drained and holds up what is
// queued behind it).  Narrower waves for the 8- / 16-entry tiers (32 / 16 lanes: more waves per SIMD, a wave waits for its
// slowest lane among fewer) were measured: no gain on the long-list workload, a loss where such a tier runs over all rows.
#ifndef SG_LANES_8
#define SG_LANES_8 64
#endif
#ifndef SG_LANES_16
#define SG_LANES_16 64
#endif
#ifndef SG_LANES_63
#define SG_LANES_63 16
#endif

// ------------------------------------------------------------------------------------------------
// Stable counting sort by channel, per frame.  grid = (tiles per frame, frames), 256 threads, a tile
// is 1024 consecutive rows; wave w owns rows [256 w, 256 w + 256) of the tile in 4 rounds of 64 so
// that "earlier row" == "earlier (wave, round, lane)".
// STATS: the tile's rows are here anyway -- the per-tile statistics of the noise-threshold prepass (sg_lean.h; needs the ground plane,
// i.e. a plane that is known when the sort starts) ride along: one pass over the rows less per step (0.67 GB of 256 sweeps).
// Ranks: every lane finds the lanes of its wave that hold the same channel by eight ballots, one per bit of the channel byte -- the
// same cost whether the 64 rows are of one channel (a channel-major sweep) or of 64 (firing order: an STF .bin interleaves the
// channels, precompute.py:78).  A loop with one round per DISTINCT channel of the wave took 1.23 ms of a 256-sweep step on firing-order
// rows against 0.35 ms on channel-major ones (profiles/r05_C2fire_*).
// tile_unsorted: 1 if a row of this tile has a smaller channel than the row before it (k_sort_scan folds the tiles of a frame: a
// frame without such a row is channel-sorted as it stands, its permutation is the identity and nobody makes or reads a copy of it).
template <typename T, bool STATS>
__global__ __launch_bounds__(SG_BLOCK) void k_sort_hist(const T *__restrict__ rows, const int64_t *__restrict__ frame_off,
                                                        int32_t *__restrict__ tile_hist, uint16_t *__restrict__ rank,
                                                        uint8_t *__restrict__ ch8, int32_t *__restrict__ status, int64_t max_tiles, SgLeanTile lean,
                                                        int32_t *__restrict__ tile_unsorted)
{
    const int f = blockIdx.y;
    // status[1] ("first offending row", -1 = none: nobody writes it before the per-beam kernels) is set here, so that ONE fill clears the
    // status words of a batch instead of two (each fill is a launch on the chain of a small batch)
    if (blockIdx.x == 0 && f == 0 && threadIdx.x == 0) status[1] = -1;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    __shared__ volatile int cnt[4][256];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 4 * 256; i += SG_BLOCK) ((volatile int *)cnt)[i] = 0;
    __syncthreads();
    int my_bucket[4], my_rank[4];
    [[maybe_unused]] T sx[4], sy[4], sz[4], si[4];
    [[maybe_unused]] bool sv[4];
    int descends = 0;
    // every load of the thread's four rows first (the rounds below are chains of ballots and LDS updates: a load inside one waits its turn)
    T sc[4], scp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + w * 256 + q * 64 + lane;
        const bool valid = r < n;
        const T *p = rows + (base + (valid ? r : 0)) * 5;
        if constexpr (STATS) { sx[q] = p[0]; sy[q] = p[1]; sz[q] = p[2]; si[q] = p[3]; sv[q] = valid; }
        sc[q] = p[4];
        scp[q] = (valid && lane == 0 && r > 0) ? p[-1] : (T)0;        // lane 0: the channel of the row before this round's first (earlier round, wave or tile)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + w * 256 + q * 64 + lane;
        const bool valid = r < n;
        int bucket = -1;
        const T c_prev = scp[q];
        if (valid) {
            const T c = sc[q];
            const int ci = (int)c;
            if ((T)ci == c && ci >= 0 && ci < 256) bucket = ci;
            else { atomicCAS(&status[0], 0, 5 /* SNOWGPU_E_CHANNELS */); bucket = 255; }
        }
        {
            int before_b = __shfl_up(bucket, 1);
            if (lane == 0) before_b = r > 0 ? (int)c_prev : bucket;
            if (valid && bucket < before_b) descends = 1;
        }
        my_bucket[q] = bucket;
        my_rank[q] = 0;
        if (valid) ch8[base + r] = (uint8_t)bucket;     // the scatter pass reads 1 byte per row instead of the row again
        unsigned long long same = __ballot(valid);      // lanes of this wave with my channel
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool set = (bucket >> bit) & 1;
            const unsigned long long bb = __ballot(set);
            same &= set ? bb : ~bb;
        }
        if (valid) {
            const int before = cnt[w][bucket];          // (a wave's LDS operations keep their order: every read precedes the leaders' writes)
            my_rank[q] = before + __popcll(same & sg_lanemask_lt());
            if ((same & sg_lanemask_lt()) == 0) cnt[w][bucket] = before + __popcll(same);
        }
    }
    const int any_descends = __syncthreads_or(descends);
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
    if (tid == 0) tile_unsorted[(int64_t)f * max_tiles + blockIdx.x] = any_descends ? 1 : 0;
    if constexpr (STATS) {
        __shared__ double sm[58];
        lean_tile_stats<T>(lean, f, blockIdx.x, sx, sy, sz, si, sv, sm);
    }
}

// One block per frame, thread v owns bucket v: tile_base[t][v] = (rows of smaller buckets) + (rows of
// bucket v in earlier tiles); frame_unsorted[f] = some tile of the frame saw a descending channel.
__global__ __launch_bounds__(SG_BLOCK) void k_sort_scan(const int64_t *__restrict__ frame_off,
                                                        const int32_t *__restrict__ tile_hist,
                                                        int32_t *__restrict__ tile_base, int64_t max_tiles,
                                                        const int32_t *__restrict__ tile_unsorted, int32_t *__restrict__ frame_unsorted)
{
    const int f = blockIdx.x, v = threadIdx.x;
    const int64_t n = frame_off[f + 1] - frame_off[f];
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    const int32_t *h = tile_hist + (int64_t)f * max_tiles * 256;
    int32_t *b = tile_base + (int64_t)f * max_tiles * 256;
    int total = 0, uns = 0;
    for (int64_t t = v; t < tiles; t += SG_BLOCK) uns |= tile_unsorted[(int64_t)f * max_tiles + t];
#pragma unroll 8                                  // eight loads in flight: the loop is a chain of global-load latencies otherwise
    for (int64_t t = 0; t < tiles; ++t) total += h[t * 256 + v];
    __shared__ int s[256];
    s[v] = total;
    uns = __syncthreads_or(uns);
    if (v == 0) frame_unsorted[f] = uns ? 1 : 0;
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

// Second pass of the sort, for the frames that need it (frame_unsorted[f]; a channel-sorted frame is read in place): the tile's rows
// go to their places in the SORTED COPY of the frame, and perm gets their source rows.  The tile is staged through LDS in sorted
// order first, so that the stores walk whole runs -- in firing order a tile holds 16 rows of each of 64 channels, i.e. 64 runs of
// 320 contiguous bytes -- instead of scattering 20-byte rows lane by lane.  The per-beam kernels and the compaction then read sorted
// position g as row g of the copy: no gather through perm anywhere (measured on firing-order rows before this: the scan 1.83 instead of
// 1.60 ms, the compaction's scatter 0.69 instead of 0.36 ms, this pass -- 4-byte stores scattered over 64 channel runs -- 0.32 ms).
// identity_perm: the debug tap wants the permutation of every frame, sorted ones too.
template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_sort_scatter(const T *__restrict__ rows, const uint8_t *__restrict__ ch8, const int64_t *__restrict__ frame_off,
                                                           const int32_t *__restrict__ tile_hist, const int32_t *__restrict__ tile_base,
                                                           const uint16_t *__restrict__ rank, int32_t *__restrict__ perm, T *__restrict__ srows,
                                                           const int32_t *__restrict__ frame_unsorted, int identity_perm, int64_t max_tiles)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const int tid = threadIdx.x;
    if (!frame_unsorted[f]) {
        if (identity_perm)
            for (int q = 0; q < 4; ++q) { const int64_t r = tile0 + q * SG_BLOCK + tid; if (r < n) perm[base + r] = (int32_t)r; }
        return;
    }
    __shared__ T stage[SG_TILE * 5];
    __shared__ int s_dest[SG_TILE];
    __shared__ uint16_t s_src[SG_TILE];
    __shared__ int s_start[256];
    const int32_t *b = tile_base + ((int64_t)f * max_tiles + blockIdx.x) * 256;
    {   // where each channel's run starts inside the tile's sorted image: exclusive scan of the tile's histogram
        const int c = tile_hist[((int64_t)f * max_tiles + blockIdx.x) * 256 + tid];
        s_start[tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int add = tid >= d ? s_start[tid - d] : 0;
            __syncthreads();
            s_start[tid] += add;
            __syncthreads();
        }
        const int excl = s_start[tid] - c;
        __syncthreads();
        s_start[tid] = excl;
        __syncthreads();
    }
    const int m = (int)(n - tile0 < SG_TILE ? n - tile0 : SG_TILE);      // rows of this tile
    for (int q = 0; q < 4; ++q) {
        const int i = q * SG_BLOCK + tid;
        if (i < m) {
            const int64_t r = tile0 + i;
            const int ch = ch8[base + r], rk = rank[base + r];
            const int sp = s_start[ch] + rk;                                 // position in the tile's sorted image
            const T *p = rows + (base + r) * 5;
            const T v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3], v4 = p[4];
            T *d = stage + sp * 5;
            d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3; d[4] = v4;
            s_dest[sp] = b[ch] + rk;                                         // frame-local sorted position
            s_src[sp] = (uint16_t)i;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < m * 5; idx += SG_BLOCK) {
        const int sp = idx / 5, j = idx - sp * 5;
        srows[(base + s_dest[sp]) * 5 + j] = stage[idx];
    }
    for (int sp = tid; sp < m; sp += SG_BLOCK) perm[base + s_dest[sp]] = (int32_t)(tile0 + s_src[sp]);
}

// The sorted copy for a caller-supplied permutation (no device sort): a plain gather; every frame counts as unsorted.
template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_gather_rows(const T *__restrict__ rows, const int64_t *__restrict__ frame_off, const int32_t *__restrict__ perm,
                                                          T *__restrict__ srows, int32_t *__restrict__ frame_unsorted, int32_t *__restrict__ status)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    if (blockIdx.x == 0 && threadIdx.x == 0) frame_unsorted[f] = 1;
    if (blockIdx.x == 0 && f == 0 && threadIdx.x == 0) status[1] = -1;       // (see k_sort_hist)
    for (int64_t r = (int64_t)blockIdx.x * SG_BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * SG_BLOCK) {
        const T *p = rows + (base + perm[base + r]) * 5;
        T *d = srows + (base + r) * 5;
        d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3]; d[4] = p[4];
    }
}

// ------------------------------------------------------------------------------------------------
// Hand-over queues (scan -> k_power) are "blocked SoA": 64 consecutive slots form a group, a group holds its P planes back to
// back.  A beam hands over what phase 2 needs: plane 0 = its range, plane 1 = its azimuth, planes 2 + 3 j .. 4 + 3 j = the
// interval angles and the range of its j-th intersecting flake (near -> far); P = 3 capacity + 2.  A wave reads / writes 512
// contiguous bytes per plane (coalesced like plain SoA), and everything one beam needs sits within P x 512 bytes.
#define SG_QPLANES(LMAX) (3 * (LMAX) + 2)
template <int P>
__device__ __forceinline__ int64_t sg_qaddr(int64_t slot, int plane)
{
    return (slot >> 6) * (int64_t)(P * 64) + (int64_t)plane * 64 + (slot & 63);
}

// ------------------------------------------------------------------------------------------------
// The per-beam kernel.  Dynamic LDS: per-thread lists, strided by the block size -- four of LMAX + 1 float64 entries where phases 1-3 run
// in place, three of LMAX where the list is handed on, and in the pass over all rows one float64 (range) and one 4-byte word per entry.
//   DICT          0: phases 1-3 in place; 1: the list is handed to k_power; 2: as 1, every distance test by the reference's expression
//                 (exact-math mode of the pass over all rows: see sg_beam.h, sg_near_ray).
//   LIST = false  direct mode, the pass over all rows: phase 1 (scan).  A beam without flakes is
//                 finished (record written); a beam with flakes hands its flake list to k_power (which builds the dict: phase 2) through its region's slice
//                 of the dict queue; a beam with more flakes than the list holds is appended to the list of the capacity tier that
//                 takes all of them (the scan counts on, so the count is exact).
//   LIST = true   a later capacity tier over its class of the tier lists; DICT = true: dict hand-over to
//                 k_power<.., LISTQ> (entry i of the class -> slot i), DICT = false: phases 1-3 in place (the entries
//                 beyond the hand-over buffer).  A fixed grid strides over the list, whose length only the device knows.
//   BLOCK         beams per block = stride of the LDS lists.  BLOCK = 16 (the 63-entry tier) still launches one wave: 16
//                 live lanes, 32 KB of LDS per block instead of 131 KB -- a block that needs most of a CU's LDS waits until
//                 one has drained, and meanwhile holds up everything queued behind it.
template <typename T, int LMAX, int BLOCK, bool LIST, int DICT>
__global__ __launch_bounds__(BLOCK < 64 ? 64 : BLOCK, (!LIST && DICT == 1 && BLOCK == 256) ? SG_FP_WAVES : 1) void k_beams(SgBeamArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // hand-over passes keep three LMAX-entry lists (interval angles, range); the in-place passes four of LMAX + 1 entries
    // (the dict and the scatterer list of phase 3 carry the hard target as entry n_flakes <= LMAX)
    constexpr int ROWS = DICT ? LMAX : LMAX + 1;
    // the pass over all rows keeps ONE word per listed flake instead of its two interval angles (sg_beam.h: sg_wave_scan, COMPACT): ranges,
    // words, scan order, counts, bin starts = 19 KB per 256 beams (three double columns: 31 KB, and a fifth of the waves the registers allow)
    constexpr bool COMPACT = !LIST && DICT != 0;
    double *s_rho = COMPACT ? (double *)smem : (double *)smem + 2 * ROWS * BLOCK;
    double *s_a1 = COMPACT ? s_rho + ROWS * BLOCK : (double *)smem;       // COMPACT: the words (uint32_t), half a double column
    double *s_a2 = COMPACT ? nullptr : s_a1 + ROWS * BLOCK;
    double *s_ratio = DICT ? nullptr : s_rho + ROWS * BLOCK;
    int *s_cnt = DICT ? (COMPACT ? (int *)(reinterpret_cast<uint32_t *>(s_a1) + ROWS * BLOCK) : (int *)(s_rho + ROWS * BLOCK)) : nullptr;   // wave scan: flakes met per beam ...
    int *s_key = DICT ? s_cnt + (BLOCK < 64 ? 64 : BLOCK) : nullptr;      // ... and the scan order of the stored ones
    int *s_st = DICT ? s_key + LMAX * BLOCK : nullptr;                    // ... and where its two bins start (two ints per lane)
    int *s_mark = DICT ? s_st + 2 * (BLOCK < 64 ? 64 : BLOCK) : nullptr;  // ... and the owner marks of a trip of its pair loop (sg_pair_owner)
    const int tid = threadIdx.x;
    const int n_las = a.las->n;
    int64_t work_n = 0, work_off = 0;
    if (LIST) {
        work_n = a.tier_info[a.cls];
        if (work_n > a.work_hi) work_n = a.work_hi;
        work_off = (int64_t)a.cls * a.tier_stride;
    }
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    int seg_f = -1, seg_ch = -1;                      // segment-ordered direct mode: the block's frame and channel
    int64_t seg_g = -1, q_base = 0;
    int q_size = 0, region = 0;
    [[maybe_unused]] int seg_blk0 = 0;                // segment-ordered direct mode: first block of the block's segment
    int64_t blk = blockIdx.x;                         // direct mode: this launch walks blocks [lo, hi) of the pass
    if (!LIST) {
        const int64_t lo = a.chunk_blk ? a.chunk_blk[a.chunk] : a.blk_lo, hi = a.chunk_blk ? a.chunk_blk[a.chunk + 1] : a.blk_hi;
        blk += lo;
        if (blk >= hi) return;                        // surplus block (the grid is an upper bound)
        blk = (int64_t)__builtin_amdgcn_readfirstlane((int)blk);   // (scalar: see below)
        if (a.seg_blk) {
            // block-uniform values, pinned to scalar registers (left to itself the compiler kept the region index in a vector
            // register pair for the whole kernel -- and spilled it when the kernel was held to five waves per SIMD)
            // the block's record -- segment, its first sorted position, its rows, frame | channel, its first block: ONE round trip (block ->
            // segment -> the segment's four arrays were two, on the chain of a dozen that a wave of this pass is)
            const int32_t *br = a.seg_of_blk + blk * SG_BLKREC;
            const int sg = __builtin_amdgcn_readfirstlane(br[0]);
            const int blk0 = __builtin_amdgcn_readfirstlane(br[4]);
            seg_blk0 = blk0;
            const int off = ((int)blk - blk0) * BLOCK + tid;
            const int fc = __builtin_amdgcn_readfirstlane(br[3]);
            seg_f = fc & 0x3fffff; seg_ch = (int)((unsigned)fc >> 22);
            q_base = (int64_t)__builtin_amdgcn_readfirstlane(br[1]);      // segments exist for n_total < 2^31
            q_size = __builtin_amdgcn_readfirstlane(br[2]); region = sg;
            if (tid < BLOCK && off < q_size) seg_g = q_base + off;
        } else {
            region = (int)((blk * BLOCK) / a.q_chunk);
            q_base = (int64_t)region * a.q_chunk;
            q_size = (int)(a.n_total - q_base < a.q_chunk ? a.n_total - q_base : a.q_chunk);
        }
    }
    int64_t chunk = LIST ? (int64_t)a.work_lo + (int64_t)blockIdx.x * BLOCK : blk * BLOCK;
    if (LIST && chunk >= work_n) return;
    do {                                              // direct mode: exactly one trip, and the compiler must see that
    int64_t g = -1;
    if (LIST) {
        if (tid < BLOCK && chunk + tid < work_n) g = a.tier_list[work_off + chunk + tid];
    } else if (seg_f >= 0) {
        g = seg_g;
    } else {
        g = chunk + tid;
        if (tid >= BLOCK || g >= a.n_total) g = -1;
    }
    // No early return from here on: wave-wide ballots / shuffles follow (queue slots, per-frame sums).
    const bool live = g >= 0;
    int f = 0, ch = 0;
    T px = 0, py = 0, pz = 0;
    [[maybe_unused]] T pint = 0;
    bool simulated = false;
    if (live) {
        f = (!LIST && seg_f >= 0) ? seg_f : sg_frame_of(a, g);
        const T *row = sg_row<T>(a, f, g);
        px = row[0]; py = row[1]; pz = row[2];
        if constexpr (!LIST && DICT) pint = row[3];   // the pass over all rows leaves range + intensity for the noise-floor pass
        if (!LIST && seg_f >= 0) {                    // the device sort only builds segments of integer channels
            ch = seg_ch;
            simulated = ch < n_las;
        } else {
            const T pch = row[4];
            ch = (int)pch;
            simulated = ((T)ch == pch) && ch >= 0 && ch < n_las;            // simulation.py:80, :482 (Q5)
        }
    }
    SgBeamOut o;
    o.overflow = 0; o.range_error = 0; o.diff2 = 0.0; o.has_power = 0; o.n_flakes = 0; o.n_hits = 0; o.label = 0; o.new_i = 0; o.k_best = 0;
    uint32_t rec = (live && !simulated) ? SG_REC_COPY : 0u;
    bool pending = false;                             // another kernel writes this row's record
    int L = 0;                                        // hand-over passes: flakes in the list
    T d_t = 0;
    double theta_c = 0.0;
    SgTable tab{};
    bool act = false;                                 // this lane simulates a beam
    [[maybe_unused]] int tier_k = -1;                 // the pass over all rows: the later tier this beam goes to
    if (simulated) {
        tab = a.frame_tables[(int64_t)f * n_las + ch];   // resolved per (frame, channel) by k_resolve_tables
        if (tab.entries == nullptr) atomicCAS(&a.status[0], 0, 1 /* SNOWGPU_E_INVALID */);
        else act = true;
    }
    if constexpr (DICT) {
        // The pass over all rows scans as a wave (every lane takes part, whether it has a beam or not): its beams' lists
        // differ 20-fold in length.  The later tiers hold beams with long lists of similar length; one beam per lane is a
        // little faster there (measured 0.37 vs 0.41 ms for tier 8).
        if (LIST && a.per_lane_scan >= 0) {
            if (act) L = sg_beam_scan<T, LMAX, BLOCK>(px, py, pz, tab, a.beam_div_deg, s_a1, s_a2, s_rho, tid, o, d_t, theta_c, a.exact_math != 0);
        } else {
            // overflow slot of this block's column 0 (sorted positions follow the columns) -- the pass over all rows only
            double *ov_blk = nullptr;
            if constexpr (!LIST) {
                if (a.ov_cap > 0) ov_blk = a.ov + (size_t)(seg_f >= 0 ? q_base + (blk - seg_blk0) * BLOCK : chunk) * SG_OV_STRIDE;
            }
            // DICT == 1: distance tests too close to call are not decided here (sg_beam.h: sg_near_ray); DICT == 2 (exact-math mode) and the
            // wave scan in a tier: every test by the reference's expression, in place
            L = sg_wave_scan<T, LMAX, BLOCK, !LIST && DICT == 1, COMPACT>(act, px, py, pz, tab, a.beam_div_deg, s_a1, s_a2, s_rho, s_cnt, s_key, s_st, tid, o, d_t, theta_c,
                                                                          a.exact_math != 0, ov_blk, ov_blk ? a.ov_cap : 0, s_mark);
            if (ov_blk && act && o.overflow && o.n_hits <= a.ov_cap) {   // header and the flakes the LDS list holds: the slot is complete
                double *sp = ov_blk + (size_t)tid * SG_OV_STRIDE;
                sp[0] = (double)d_t; sp[1] = theta_c;
                if constexpr (COMPACT) {
                    double th_r, th_l;
                    sg_beam_limits(theta_c, a.beam_div_deg, th_r, th_l);
#pragma unroll
                    for (int j = 0; j < LMAX; ++j) {
                        double x1, x2;
                        sg_hit_angles(reinterpret_cast<const uint32_t *>(s_a1)[j * BLOCK + tid], tab.entries, th_r, th_l, x1, x2);
                        sp[2 + 3 * j] = x1; sp[3 + 3 * j] = x2; sp[4 + 3 * j] = s_rho[j * BLOCK + tid];
                    }
                } else {
                    for (int j = 0; j < LMAX; ++j) {
                        sp[2 + 3 * j] = s_a1[j * BLOCK + tid]; sp[3 + 3 * j] = s_a2[j * BLOCK + tid]; sp[4 + 3 * j] = s_rho[j * BLOCK + tid];
                    }
                }
                a.ov_sc[g] = (uint16_t)(o.n_hits | (ch << 8));
            }
        }
        if constexpr (!LIST) {
            if (act && a.rng) ((T *)a.rng)[g] = d_t;  // simulation.py:89, for the noise-floor pass (:465-469, :518-520)
        }
        if (act) {
            o.has_power = !o.overflow && L > 0;       // k_power builds the dict (phase 2) and everything after it
            if (!o.overflow && L == 0 && a.dbg_count) {   // debug tap: the dict of a clear beam is its hard target alone
                a.dbg_count[g] = 1;
                a.dbg_rj[g * a.dbg_cap] = (double)d_t;
                a.dbg_ratio[g * a.dbg_cap] = sg_clear_beam_ratio(theta_c, a.beam_div_deg);
            }
        }
    } else if (act) {
        int32_t *dc = a.dbg_count ? a.dbg_count + g : nullptr;
        double *drj = a.dbg_count ? a.dbg_rj + g * a.dbg_cap : nullptr;
        double *dra = a.dbg_count ? a.dbg_ratio + g * a.dbg_cap : nullptr;
        sg_beam<T, LMAX, BLOCK>(px, py, pz, ch, tab, a.las, a.beam_div_deg, s_a1, s_a2, s_rho, s_ratio, tid, o, a.dbg_cap, dc, drj,
                                dra, a.exact_math != 0);
    }
    if (act) {
        if (o.overflow) {
            o.has_power = 0;
            int k = 0;                                // the first later tier that holds every flake of this beam
            while (k < a.n_cls && o.n_hits > a.cls_cap[k]) ++k;
            if (!LIST && (o.n_hits & SG_HITS_UNDECIDED)) k = a.n_cls - 1;   // an undecided distance test: the global-list tier scans this beam again
            if (LIST || k >= a.n_cls) {               // a listed beam fits its tier by construction
                atomicCAS(&a.status[0], 0, 6 /* SNOWGPU_E_OVERFLOW */);
                atomicCAS(&a.status[1], -1, (int32_t)g);
            } else {
                tier_k = k;                           // appended to that tier's list below
                pending = true;
            }
        } else if (o.range_error) {
            atomicCAS(&a.status[0], 0, 4 /* SNOWGPU_E_RANGE */);
            atomicCAS(&a.status[1], -1, (int32_t)g);
        }
    }
    if constexpr (!LIST) {
        // ---- tier lists: an over-full beam goes on its REGION's slice of its tier's list -- one atomic per wave and class on the
        // region's counter (thousands of addresses; one counter per class for the whole batch was measured: 300 000 atomics on one
        // address, device scope, made this pass 40 % slower).  k_power_plan / k_tier_gather close the slices up afterwards.
        // Nothing to do for the 96 % of the waves without an over-full beam.
        if (__ballot(tier_k >= 0)) {
            for (int k = 0; k < a.n_cls; ++k) {
                const unsigned long long m = __ballot(tier_k == k);
                if (!m) continue;
                const int leader = __ffsll((long long)m) - 1;
                int base = 0;
                if ((tid & 63) == leader) base = atomicAdd(&a.tn[(int64_t)__builtin_amdgcn_readfirstlane(region) * SG_MAX_CLASSES + k], (int)__popcll(m));
                base = __shfl(base, leader);
                if (tier_k == k) a.tier_sparse[(int64_t)k * a.tier_stride + q_base + base + (int)__popcll(m & sg_lanemask_lt())] = (int32_t)g;
            }
        }
    }
    if constexpr (DICT) {
        // ---- hand the beams that met a flake, with their flake lists, to k_power ------------------------------------
        // Only a fraction of the beams gets here; walking phases 2-3 in place would keep most lanes of every wave idle
        // (measured: phase 2 alone was 30 % of this pass at 28 % of the lanes).
        constexpr int P = SG_QPLANES(LMAX);
        auto hand_over = [&](double *q, int64_t slot) {
            q[sg_qaddr<P>(slot, 0)] = (double)d_t;
            q[sg_qaddr<P>(slot, 1)] = theta_c;
            if constexpr (COMPACT) {
                // the interval angles from the words: the beam's limits, or the records' tangent angles (read here, by the few beams that
                // listed a flake -- the records' lines are what the scan just read)
                double th_r, th_l;
                sg_beam_limits(theta_c, a.beam_div_deg, th_r, th_l);
                if constexpr (LMAX <= 4) {
                    double x1[LMAX], x2[LMAX];
                    uint32_t hw[LMAX];
#pragma unroll
                    for (int j = 0; j < LMAX; ++j) hw[j] = reinterpret_cast<const uint32_t *>(s_a1)[j * BLOCK + tid];
                    sg_hit_angles_all<LMAX>(hw, L, tab.entries, th_r, th_l, x1, x2);
#pragma unroll
                    for (int j = 0; j < LMAX; ++j)
                        if (j < L) {
                            q[sg_qaddr<P>(slot, 2 + 3 * j)] = x1[j];
                            q[sg_qaddr<P>(slot, 3 + 3 * j)] = x2[j];
                            q[sg_qaddr<P>(slot, 4 + 3 * j)] = s_rho[j * BLOCK + tid];
                        }
                } else {
                    for (int j = 0; j < L; ++j) {
                        double x1, x2;
                        sg_hit_angles(reinterpret_cast<const uint32_t *>(s_a1)[j * BLOCK + tid], tab.entries, th_r, th_l, x1, x2);
                        q[sg_qaddr<P>(slot, 2 + 3 * j)] = x1;
                        q[sg_qaddr<P>(slot, 3 + 3 * j)] = x2;
                        q[sg_qaddr<P>(slot, 4 + 3 * j)] = s_rho[j * BLOCK + tid];
                    }
                }
            } else {
                for (int j = 0; j < L; ++j) {
                    q[sg_qaddr<P>(slot, 2 + 3 * j)] = s_a1[j * BLOCK + tid];
                    q[sg_qaddr<P>(slot, 3 + 3 * j)] = s_a2[j * BLOCK + tid];
                    q[sg_qaddr<P>(slot, 4 + 3 * j)] = s_rho[j * BLOCK + tid];
                }
            }
        };
        if constexpr (!LIST) {
            // The region's slice of the queue: beams with one flake from the front, the others from the back -- one
            // packed 64-bit atomic per wave on the region's counter (thousands of distinct addresses: no serialisation).
            const bool front = o.has_power && L <= a.front_max, back = o.has_power && L > a.front_max;
            const unsigned long long mf = __ballot(front), mb = __ballot(back);
            if (mf | mb) {
                const int leader = __ffsll((long long)(mf | mb)) - 1;
                unsigned long long base = 0;
                if ((tid & 63) == leader)
                    base = atomicAdd(&a.qn[__builtin_amdgcn_readfirstlane(region)], (unsigned long long)__popcll(mf) | ((unsigned long long)__popcll(mb) << 32));
                const unsigned blo = __shfl((unsigned)(base & 0xffffffffull), leader), bhi = __shfl((unsigned)(base >> 32), leader);
                if (o.has_power) {
                    const int64_t slot = front ? q_base + (int)blo + (int)__popcll(mf & sg_lanemask_lt())
                                               : q_base + q_size - 1 - ((int)bhi + (int)__popcll(mb & sg_lanemask_lt()));
                    hand_over(a.dq, slot);
                    a.dq_g[slot] = (int32_t)g;
                    a.dq_sc[slot] = (uint16_t)(L | (ch << 8));
                    // this row's record is a reference to its queue slot: k_power writes its result there, slot after slot
                    // (a 4-byte store per beam scattered over the sorted positions costs a whole memory sector each)
                    rec = SG_REC_SLOT | (uint32_t)slot;
                }
            }
        } else if (live) {
            const int64_t slot = chunk + tid;         // entry i of the class -> slot i
            uint16_t sc = 0xffff;                     // no flake list: the record below is final
            if (o.has_power) {
                hand_over(a.tq, slot);
                sc = (uint16_t)(L | (ch << 8));
                pending = true;
            }
            a.tq_sc[slot] = sc;
        }
    } else {
        // ---- phases 3b / 3c in place (s_ratio is dead after phase 3a and carries the work lists) -----------------
        constexpr int NB = LMAX <= 4 ? 4 : 8;    // bins carried together
        if (o.has_power) {
            double best = 0.0;
            int k_best = 0;
            if (a.exact_math) sg_lane_power<BLOCK, true, NB, LMAX>(o.n_flakes, a.rgrid, s_a1, s_a2, s_rho, s_ratio, tid, best, k_best);
            else sg_lane_power<BLOCK, false, NB, LMAX>(o.n_flakes, a.rgrid, s_a1, s_a2, s_rho, s_ratio, tid, best, k_best);
            if constexpr (SgReal<T>::is_f32) d_t = sqrtf((px * px + py * py) + pz * pz);
            else d_t = sqrt((px * px + py * py) + pz * pz);
            sg_beam_decide((double)d_t, ch, a.las, best, k_best, o);
            rec = sg_pack_record(o);
        }
        sg_add_diff2(a.diff2, live, f, (long long)o.diff2);
    }
    if constexpr (!LIST && DICT) {
        // a beam this pass finishes itself (no flake met: label 0) carries its original intensity in the record, if that is an
        // integer in [0, 255] as in every STF sweep: together with the range above the noise-floor pass then never reads the row
        if (live && !pending && rec == 0u && act && a.rng) {
            const int iv = (int)pint;
            if ((T)iv == pint && iv >= 0 && iv <= 255) rec = SG_REC_HAS_I | (uint32_t)iv;
        }
    }
    if (live && !pending) a.rec[g] = rec;
    } while (LIST && (chunk += stride) < work_n);
}

// ------------------------------------------------------------------------------------------------
// The scan of a later capacity tier WITHOUT lists in LDS: one beam per lane walks its bins (simulation.py:338-410) and writes
// every intersecting flake straight into the beam's slot of the tier's hand-over buffer, in the order it meets them; the
// near -> far order (simulation.py:413-417) is made by k_power<.., LISTQ> when it loads the slot into its own LDS lists
// (insertion by range, scan order on equal ranges -- the order sg_beam_scan's insertion produces).  The scan is a chain of
// dependent record loads (12 % VALU issue, 65 % waiting: profiles/r04a_*_pmc.txt); with three LMAX-entry lists per beam in LDS
// it ran at 1.25 (16 entries) to 2.5 (8 entries) waves per SIMD, without them at what its registers allow.  Stores are
// coalesced like the old hand-over's: lane = slot, so the lanes of a wave that meet their h-th flake together write neighbours.
template <typename T, int LMAX>
__global__ __launch_bounds__(256, 4) void k_tier_scan_direct(SgBeamArgs a)
{
    constexpr int P = SG_QPLANES(LMAX);
    // The rare long-list class: a handful of beams, each a chain of dependent record loads, beside persistent kernels that fill every SIMD --
    // and everything behind this chain (the step's last link since the prepass got shorter) waits for it.  Its waves go first.
    if constexpr (LMAX > 16) __builtin_amdgcn_s_setprio(3);
    const int n_las = a.las->n;
    int64_t work_n = a.tier_info[a.cls];
    if (work_n > a.work_hi) work_n = a.work_hi;
    const int64_t work_off = (int64_t)a.cls * a.tier_stride;
    for (int64_t i = (int64_t)a.work_lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < work_n; i += (int64_t)gridDim.x * 256) {
        const int32_t g = a.tier_list[work_off + i];
        const int f = sg_frame_of(a, g);
        const T *row = sg_row<T>(a, f, g);
        const T px = row[0], py = row[1], pz = row[2];
        const int ch = (int)row[4];                                       // a flagged beam was simulated: valid channel
        const SgTable tab = a.frame_tables[(int64_t)f * n_las + ch];
        const int64_t slot = i;                                           // entry i of the class -> slot i
        if (tab.entries == nullptr) { atomicCAS(&a.status[0], 0, 1 /* SNOWGPU_E_INVALID */); a.tq_sc[slot] = 0xffff; a.rec[g] = 0u; continue; }
        T d_t;
        const SgBeamGeo geo = sg_beam_geometry<T>(px, py, pz, a.beam_div_deg, a.exact_math != 0, d_t);
        const int nb = (int)tab.n_bins;
        const int b_lo = sg_bin_of(geo.theta_r - SG_BEAM_MARGIN, tab.inv_bin_w, nb);
        const int b_hi = sg_bin_of(geo.theta_l + SG_BEAM_MARGIN, tab.inv_bin_w, nb);
        int span = b_hi - b_lo;
        if (span < 0) span += nb;
        int hits = 0, b = b_lo;
        double *q = a.tq + (slot >> 6) * (int64_t)(P * 64) + (slot & 63);   // plane p of this slot: q[p * 64]
        for (int s = 0; s <= span; ++s) {
            const uint32_t e0 = tab.bin_start[b], e1 = tab.bin_start[b + 1];
            SgEntry nxt = tab.entries[e0];                              // (one spare record at the end of the array: e + 1 is always readable)
            for (uint32_t e = e0; e < e1; ++e) {
                const SgEntry fl = nxt;
                nxt = tab.entries[e + 1];
                if (!(fl.rho < geo.d)) break;                           // :345 (bins are sorted by range)
                if (s > 0 && !(fl.flags & 1u)) continue;                // already met in an earlier bin
                double na1, na2;
                if (!sg_flake_hits(geo, fl, na1, na2)) continue;
                if (hits < LMAX) { q[(2 + 3 * hits) * 64] = na1; q[(3 + 3 * hits) * 64] = na2; q[(4 + 3 * hits) * 64] = fl.rho; }
                ++hits;
            }
            if (++b == nb) b = 0;
        }
        if (hits > LMAX) {                                                // a listed beam fits its tier by construction
            atomicCAS(&a.status[0], 0, 6 /* SNOWGPU_E_OVERFLOW */);
            atomicCAS(&a.status[1], -1, g);
            a.tq_sc[slot] = 0xffff; a.rec[g] = 0u;
        } else if (hits == 0) {
            a.tq_sc[slot] = 0xffff;                                       // no flake list: the record is final
            a.rec[g] = 0u;
            if (a.dbg_count) {                                            // debug tap: the dict of a clear beam is its hard target alone
                a.dbg_count[g] = 1;
                a.dbg_rj[(int64_t)g * a.dbg_cap] = (double)d_t;
                a.dbg_ratio[(int64_t)g * a.dbg_cap] = sg_clear_beam_ratio(geo.theta_c, a.beam_div_deg);
            }
        } else {
            q[0] = (double)d_t; q[64] = geo.theta_c;
            a.tq_sc[slot] = (uint16_t)(hits | (ch << 8));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Work items of k_power for the queue of a direct-mode pass: one item = up to `lanes` consecutive live slots of one
// region (its front run, then its back run).  One thread per region; items are appended with one atomic per wave.
// item = {first slot, count | (frame + 1) << 10}; the back run (beams with several flakes) in items of lanes_back slots.
// region r of this launch's block range: its slice [q_base, q_base + q_size) of the sorted positions, its frame + 1 (segments)
__device__ __forceinline__ bool sg_region(const SgBeamArgs &a, int r, int n_regions_ub, int64_t &q_base, int &q_size, int &f1)
{
    const int64_t lo = a.chunk_blk ? a.chunk_blk[a.chunk] : a.blk_lo, hi = a.chunk_blk ? a.chunk_blk[a.chunk + 1] : a.blk_hi;
    bool mine = false;
    q_base = 0; q_size = 0; f1 = 0;
    if (a.seg_blk) {
        if (r < a.seg_n[0]) {
            const int64_t b0 = a.seg_blk[r];
            mine = b0 >= lo && b0 < hi;               // chunks are cut at segment starts
            if (mine) { q_base = a.seg_start[r]; q_size = a.seg_cnt[r]; f1 = (a.seg_frame[r] & 0x3fffff) + 1; }
        }
    } else if (r < n_regions_ub) {
        q_base = (int64_t)r * a.q_chunk;
        const int64_t b0 = q_base / a.blk_rows;
        mine = q_base < a.n_total && b0 >= lo && b0 < hi;
        if (mine) q_size = (int)(a.n_total - q_base < a.q_chunk ? a.n_total - q_base : a.q_chunk);
    }
    return mine;
}

__global__ __launch_bounds__(256) void k_power_plan(SgBeamArgs a, int lanes, int lanes_back, int n_regions_ub)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    int n_items = 0, nf = 0, nb = 0, q_size = 0, f1 = 0;
    int64_t q_base = 0;
    const bool mine = sg_region(a, r, n_regions_ub, q_base, q_size, f1);
    // the later tiers' lists: where this region's slice of each goes in the closed-up list (tier_info[k] ends up as the length)
    for (int k = 0; k < a.n_cls; ++k) {
        const int c = mine ? a.tn[(int64_t)r * SG_MAX_CLASSES + k] : 0;
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if ((int)(threadIdx.x & 63) >= o) inc += v; }
        const int total = __shfl(inc, 63);
        int base = 0;
        if ((threadIdx.x & 63) == 63 && total > 0) base = atomicAdd(&a.tier_info[k], total);
        base = __shfl(base, 63) + inc - c;
        if (mine) a.tbase[(int64_t)r * SG_MAX_CLASSES + k] = base;
    }
    int n_one = 0;                                    // items of k_power_few (beams with few flakes: the front of the region's slice)
    if (mine) {
        const unsigned long long c = a.qn[r];
        nf = (int)(c & 0xffffffffull); nb = (int)(c >> 32);
        n_one = (nf + lanes - 1) / lanes;
        n_items = (nb + lanes_back - 1) / lanes_back;
        if (!a.pw_items1) { n_items += n_one; n_one = 0; }
    }
    auto reserve = [&](int n, int32_t *counter) {    // inclusive scan over the wave, one atomic for its total
        int inc = n;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if ((int)(threadIdx.x & 63) >= o) inc += v; }
        const int total = __shfl(inc, 63);
        int b = 0;
        if ((threadIdx.x & 63) == 63 && total > 0) b = atomicAdd(counter, total);
        return __shfl(b, 63) + inc - n;
    };
    if (a.back_list) {                                // one list for k_power_all: the region's back run goes to bbase[r] (k_tier_gather copies it)
        const int bb = reserve(nb, a.pw_count + 2);
        if (mine) a.bbase[r] = bb;
        n_items -= (nb + lanes_back - 1) / lanes_back;
        nb = 0;                                       // (no per-region items for the back runs)
    }
    int base = reserve(n_items, a.pw_count);
    if (a.pw_items1) {
        int base1 = reserve(n_one, a.pw_count + 1);
        for (int k = 0; k < nf; k += lanes)
            a.pw_items1[base1++] = make_int2((int)(q_base + k), (nf - k < lanes ? nf - k : lanes) | (f1 << 10));
    } else {
        for (int k = 0; k < nf; k += lanes)
            a.pw_items[base++] = make_int2((int)(q_base + k), (nf - k < lanes ? nf - k : lanes) | (f1 << 10));
    }
    for (int k = 0; k < nb; k += lanes_back)
        a.pw_items[base++] = make_int2((int)(q_base + q_size - nb + k), (nb - k < lanes_back ? nb - k : lanes_back) | (f1 << 10));
}

// The regions' slices of the tier lists, closed up: one wave per region (bases from k_power_plan); reports the list lengths.
__global__ __launch_bounds__(256) void k_tier_gather(SgBeamArgs a, int n_regions_ub)
{
    const int r = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    if (blockIdx.x == 0 && threadIdx.x < SG_MAX_CLASSES) a.status[2 + threadIdx.x] = a.tier_info[threadIdx.x];   // beams per later tier
    if (blockIdx.x == 0 && threadIdx.x < SG_MAX_CLASSES && a.tier_hint) a.tier_hint[threadIdx.x] = a.tier_info[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.tier_hint) a.tier_hint[SG_MAX_CLASSES] = (int32_t)((a.n_total >> 10) + 1);   // ... of a batch of this many K rows
    int q_size = 0, f1 = 0;
    int64_t q_base = 0;
    if (!sg_region(a, r, n_regions_ub, q_base, q_size, f1)) return;
    for (int k = 0; k < a.n_cls; ++k) {
        const int c = a.tn[(int64_t)r * SG_MAX_CLASSES + k];
        if (c == 0) continue;
        const int32_t *src = a.tier_sparse + (int64_t)k * a.tier_stride + q_base;
        int32_t *dst = a.tier_list + (int64_t)k * a.tier_stride + a.tbase[(int64_t)r * SG_MAX_CLASSES + k];
        for (int i = lane; i < c; i += 64) dst[i] = src[i];
    }
    if (a.back_list) {                                // the region's multi-flake beams: the last nb slots of its slice of the queue
        const int nb = (int)(a.qn[r] >> 32);
        int32_t *dst = a.back_list + a.bbase[r];
        const int first = (int)(q_base + q_size - nb);
        for (int i = lane; i < nb; i += 64) dst[i] = first + i;
    }
}

// ------------------------------------------------------------------------------------------------
// Everything after the scan for the beams a hand-over pass queued: one lane per queue slot, every lane busy.
// Phase 2 (occlusion dict), 3a (amplitudes, windows), 3b (pruned power profile and its first maximum), 3c (decision),
// result record.  Nothing is read but the queue: range, azimuth and flake list of the beam; its channel rides in the slot.
//   LISTQ = false  the direct-mode pass's queue: work items from k_power_plan (runs of live slots: 64 one-flake slots, or a
//                  window of SG_KP_WIN x 64 multi-flake slots)
//   LISTQ = true   a list-mode pass's hand-over buffer: item i = window i of the class
// ONE work item of the received-power phase: `cnt` hand-over slots from `start` on (a run of live slots of the direct-mode queue, LISTQ = false;
// a window of a list-mode class, LISTQ = true, whose lists lie in the overflow slots (ov_list) or in the class' hand-over buffer, in scan
// order if tq_unsorted), by the calling WAVE, whose LDS columns start at `smem` (BLOCK >= 64: the block's columns, the wave takes its 64).
// item_f: the item's frame (direct-mode items carry it), or -1.  Called by k_power -- one capacity per launch -- and by k_power_all, where a
// wave takes items of any capacity from one list.
template <typename T, int LMAX, int BLOCK, bool LISTQ>
__device__ __forceinline__ void sg_kp_item(const SgBeamArgs &a, char *smem, const int ov_list, const int tq_unsorted, const int64_t work_off,
                                           const int start, const int cnt, const int item_f, const int32_t *__restrict__ slot_list = nullptr)
{
    // slot_list (LISTQ = false): the item is entries [start, start + cnt) of this list of queue slots (k_power_all: the closed-up back runs)
    constexpr int LANES = BLOCK < 64 ? BLOCK : 64;
    constexpr int WIN = BLOCK < 64 ? 1 : SG_KP_WIN;       // waves' worth of slots per work item
    constexpr int P = SG_QPLANES(LMAX);
    double *s_a1 = (double *)smem;
    // Three list columns for capacities up to 16 (THREE): interval angles a1, a2 and the ratio / range column -- the flakes' ranges
    // stay in the hand-over queue until the dict is done (sg_beam_dict<.., KEEP_RHO = false> returns which list entry each dict
    // entry came from), then amplitude -> a1's cells, bin window + work list -> a2's, range -> the ratio's.  One column less is
    // one more block per CU where the lists are long (16 entries: 26 KB instead of 35 KB per wave: six waves per CU, not four).
    constexpr bool THREE = SG_KP_THREE_MAX > 0 && LMAX <= SG_KP_THREE_MAX && LMAX >= SG_KP_THREE_MIN;
    double *s_a2 = s_a1 + (LMAX + 1) * BLOCK;
    double *s_rho = s_a2 + (LMAX + 1) * BLOCK;                            // THREE: the ratio column first, the ranges afterwards
    double *s_ratio = THREE ? s_rho : s_rho + (LMAX + 1) * BLOCK;
    double *s_work = THREE ? s_a2 : s_ratio;                              // stage-A work list: low words of the window column / own column
    const int tid = threadIdx.x, lane = tid & 63;
    const int ltid = BLOCK < 64 ? lane : tid;          // column of the LDS lists
    const double *planes = LISTQ ? a.tq : a.dq;
    const uint16_t *scs = LISTQ ? a.tq_sc : a.dq_sc;
    static_assert(WIN <= 4, "the window's permutation passes through one 64-double row segment of the wave");
    uint16_t *s_perm = (uint16_t *)(s_ratio + (BLOCK < 64 ? 0 : (tid & ~63)));   // dead between two beams: this wave's columns of row 0
    int perm[WIN];
    // A window of several waves' worth of slots is taken in order of flake count: the cost of phases 2 and 3 grows
    // steeply with the list length, and a wave is as slow as its longest list (counting sort by ballots; the slots of
    // a window are neighbours in every plane of the queue, so the permuted reads touch the same lines).
    const bool sorted = WIN > 1 && cnt > LANES;          // wave-uniform
    if constexpr (WIN > 1) {
        if (sorted) {
            int key[WIN], rank[WIN];
            // the window's flake counts in one round of loads per level of indirection: every lane reads a valid slot (slot 0 of the window
            // stands in for those past its end) -- under `idx < cnt` each of the WIN loads waited for the one before
            int64_t at[WIN];
            unsigned scw[WIN];
#pragma unroll
            for (int r = 0; r < WIN; ++r) { const int idx = r * 64 + lane; at[r] = (int64_t)start + (idx < cnt ? idx : 0); }
            if (LISTQ && ov_list) {                       // (the branch outside the loops: a branch per slot put a wait behind every load)
#pragma unroll
                for (int r = 0; r < WIN; ++r) at[r] = (int64_t)a.tier_list[work_off + at[r]];
#pragma unroll
                for (int r = 0; r < WIN; ++r) scw[r] = (unsigned)a.ov_sc[at[r]];
            } else {
                if (!LISTQ && slot_list) {
#pragma unroll
                    for (int r = 0; r < WIN; ++r) at[r] = (int64_t)slot_list[at[r]];
                }
#pragma unroll
                for (int r = 0; r < WIN; ++r) scw[r] = (unsigned)scs[at[r]];
            }
#pragma unroll
            for (int r = 0; r < WIN; ++r) {
                const int idx = r * 64 + lane;
                key[r] = 255;                             // past the end of the window: last
                if (idx < cnt) key[r] = scw[r] == 0xffffu ? 254 : (int)(scw[r] & 255u);
                rank[r] = 0;
            }
            int base = 0;
            auto place = [&](int c) {
#pragma unroll
                for (int r = 0; r < WIN; ++r) {
                    const unsigned long long m = __ballot(key[r] == c);
                    if (key[r] == c) rank[r] = base + (int)__popcll(m & sg_lanemask_lt());
                    base += (int)__popcll(m);
                }
            };
            for (int c = 0; c <= LMAX; ++c) place(c);
            place(254); place(255);
#pragma unroll
            for (int r = 0; r < WIN; ++r) s_perm[rank[r]] = (uint16_t)(r * 64 + lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // written and read by the lanes of one wave: no barrier, but keep the order
#pragma unroll
            for (int r = 0; r < WIN; ++r) perm[r] = s_perm[r * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the lists of the first beam overwrite it)
        }
    }
    const int rounds = (cnt + LANES - 1) / LANES;
    for (int r = 0; r < rounds; ++r) {
        const int idx = r * LANES + lane;
        const bool in = lane < LANES && idx < cnt;
        int pos = idx;
        if constexpr (WIN > 1) {
            if (sorted) {
#pragma unroll
                for (int q = 0; q < WIN; ++q) if (q == r) pos = perm[q];
            }
        }
        int64_t slot = (int64_t)start + pos;
        if constexpr (!LISTQ) { if (slot_list && in) slot = slot_list[slot]; }
        unsigned sc = 0xffffu;
        int32_t g = 0;
        double d = 0.0, tc = 0.0, f_a1 = 0.0, f_a2 = 0.0, f_rho = 0.0;
        // where the beam's hand-over data lies: plane p at qb[p * qs] -- a slot of a blocked-SoA queue (stride 64), or the
        // overflow slot of its sorted position (stride 1)
        const double *qb = planes;
        int qs = 64;
        if (in) {                                         // everything a beam surely has, in one round of loads
            g = LISTQ ? a.tier_list[work_off + slot] : a.dq_g[slot];
            if (LISTQ && ov_list) { qb = a.ov + (size_t)g * SG_OV_STRIDE; qs = 1; sc = a.ov_sc[g]; }
            else { qb = planes + sg_qaddr<P>(slot, 0); sc = scs[slot]; }
            d = qb[0];                                    // the beam's range (simulation.py:89), widened from the row dtype
            tc = qb[qs];
            f_a1 = qb[2 * qs]; f_a2 = qb[3 * qs]; f_rho = qb[4 * qs];
        }
        const bool live = in && sc != 0xffffu;            // 0xffff: a listed beam without a flake (its record is final)
        const int L = (int)(sc & 255u), ch = (int)(sc >> 8);
        int f = 0;
        SgBeamOut o;
        o.overflow = 0; o.range_error = 0; o.diff2 = 0.0; o.has_power = 0; o.n_flakes = 0; o.n_hits = 0; o.label = 0; o.new_i = 0; o.k_best = 0;
        constexpr int NB = LMAX <= 4 ? SG_NB4 : SG_NB_TIERS;   // bins of the power profile carried together
        int S = 0, nw = 0, k_best = 0;
        double best = 0.0;
        if (live) {
            f = item_f >= 0 ? item_f : sg_frame_of(a, g);
            // ord: 4 bits per list entry -- the hit (scan order) it came from; identity unless the scan left the flakes unsorted
            unsigned long long ord = 0xfedcba9876543210ull;
            if (LISTQ && (tq_unsorted || ov_list)) {
                // the flakes as the scan met them: insertion by range, scan order on equal ranges (simulation.py:413-417).  The
                // ranges pass through the ratio / range column (free until the dict); THREE: they go back to being fetched from
                // the queue afterwards, through `ord`.
                s_a1[ltid] = f_a1; s_a2[ltid] = f_a2; s_rho[ltid] = f_rho;
                if constexpr (THREE) ord = 0;
                // first every flake into the columns as it lies in the queue -- independent loads, all in flight together --, then
                // the insertion sort on LDS alone (a load per step of the sort would put a memory latency into every step)
                // (eight flakes per round, every load of a round issued before its first store: `#pragma unroll 4` left a remainder loop of up to
                // three trips that waited for each trip's loads -- seven flakes took four rounds, the 63-entry class up to eighteen)
                for (int h0 = 1; h0 < L; h0 += 8) {
                    double xa[8], xb[8], xr[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int h = h0 + c < L ? h0 + c : 0;     // (past the list: plane 0 again, not stored)
                        xa[c] = qb[(2 + 3 * h) * qs]; xb[c] = qb[(3 + 3 * h) * qs]; xr[c] = qb[(4 + 3 * h) * qs];
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (h0 + c < L) { s_a1[(h0 + c) * BLOCK + ltid] = xa[c]; s_a2[(h0 + c) * BLOCK + ltid] = xb[c]; s_rho[(h0 + c) * BLOCK + ltid] = xr[c]; }
                }
                for (int h = 1; h < L; ++h) {
                    const double x1 = s_a1[h * BLOCK + ltid], x2 = s_a2[h * BLOCK + ltid], r = s_rho[h * BLOCK + ltid];
                    int q = h;
                    // (equal ranges of different flakes -- probability zero for sampled tables -- fall back on the interval angles, so
                    // that the order never depends on the order in which the scan's lanes reached the slot)
                    while (q > 0 && (s_rho[(q - 1) * BLOCK + ltid] > r ||
                                     (s_rho[(q - 1) * BLOCK + ltid] == r && ov_list &&
                                      (s_a1[(q - 1) * BLOCK + ltid] > x1 || (s_a1[(q - 1) * BLOCK + ltid] == x1 && s_a2[(q - 1) * BLOCK + ltid] > x2))))) {
                        s_rho[q * BLOCK + ltid] = s_rho[(q - 1) * BLOCK + ltid]; s_a1[q * BLOCK + ltid] = s_a1[(q - 1) * BLOCK + ltid];
                        s_a2[q * BLOCK + ltid] = s_a2[(q - 1) * BLOCK + ltid];
                        --q;
                    }
                    if (q != h) { s_rho[q * BLOCK + ltid] = r; s_a1[q * BLOCK + ltid] = x1; s_a2[q * BLOCK + ltid] = x2; }
                    if constexpr (THREE) {
                        const unsigned long long low = (1ull << (4 * q)) - 1ull;
                        ord = (ord & low) | ((unsigned long long)h << (4 * q)) | ((ord & ~low) << 4);
                    }
                }
            } else {
                s_a1[ltid] = f_a1; s_a2[ltid] = f_a2;
                if constexpr (!THREE) s_rho[ltid] = f_rho;
                if constexpr (LMAX <= 4 && !THREE) {
                    // every plane of a slot exists, filled or not: the whole list in ONE round of loads (a loop of L - 1 trips waited for each trip's three)
                    double xa[LMAX - 1], xb[LMAX - 1], xr[LMAX - 1];
#pragma unroll
                    for (int j = 1; j < LMAX; ++j) { xa[j - 1] = qb[(2 + 3 * j) * qs]; xb[j - 1] = qb[(3 + 3 * j) * qs]; xr[j - 1] = qb[(4 + 3 * j) * qs]; }
#pragma unroll
                    for (int j = 1; j < LMAX; ++j)
                        if (j < L) { s_a1[j * BLOCK + ltid] = xa[j - 1]; s_a2[j * BLOCK + ltid] = xb[j - 1]; s_rho[j * BLOCK + ltid] = xr[j - 1]; }
                } else {
                    for (int j = 1; j < L; ++j) {
                        s_a1[j * BLOCK + ltid] = qb[(2 + 3 * j) * qs];
                        s_a2[j * BLOCK + ltid] = qb[(3 + 3 * j) * qs];
                        if constexpr (!THREE) s_rho[j * BLOCK + ltid] = qb[(4 + 3 * j) * qs];
                    }
                }
            }
            auto queue_rho = [&](int j) -> double {       // range of list entry j (THREE: from the queue)
                const int h = (int)((ord >> (4 * j)) & 15ull);
                return h == 0 ? f_rho : qb[(4 + 3 * h) * qs];
            };
            int32_t *dc = a.dbg_count ? a.dbg_count + g : nullptr;
            double *drj = a.dbg_count ? a.dbg_rj + (int64_t)g * a.dbg_cap : nullptr;
            double *dra = a.dbg_count ? a.dbg_ratio + (int64_t)g * a.dbg_cap : nullptr;
            unsigned long long srcmap = 0;
            if constexpr (THREE) S = sg_beam_dict<LMAX, BLOCK, false>(L, tc, d, a.beam_div_deg, s_a1, s_a2, nullptr, s_ratio, ltid, 0, nullptr, nullptr, nullptr, 0, &srcmap);
            else S = sg_beam_dict<LMAX, BLOCK>(L, tc, d, a.beam_div_deg, s_a1, s_a2, s_rho, s_ratio, ltid, a.dbg_cap, dc, drj, dra);
            if constexpr (THREE) {
                if (dc) {                                   // debug tap: ratios now, ranges from the queue
                    *dc = S + 1;
                    for (int t = 0; t <= S && t < a.dbg_cap; ++t) {
                        const int j = (int)((srcmap >> (4 * t)) & 15ull);
                        drj[t] = t == S ? d : queue_rho(j);
                        dra[t] = s_ratio[t * BLOCK + ltid];
                    }
                }
            }
            if (S > 0) {                                // S == 0: no flake owns a slot -> label 0 (simulation.py:133)
                const T d_t = (T)d;                     // exact
                if constexpr (THREE) {
                    sg_beam_amp3<T, LMAX, BLOCK>(d_t, S, ch, a.las, s_a1, s_a2, s_rho, ltid, o,
                                                 [&](int t) { return queue_rho((int)((srcmap >> (4 * t)) & 15ull)); });
                } else sg_beam_amp<T, LMAX, BLOCK>(d_t, S, ch, a.las, s_a1, s_a2, s_rho, s_ratio, ltid, o, 0);
                if (o.range_error) {
                    atomicCAS(&a.status[0], 0, 4 /* SNOWGPU_E_RANGE */);
                    atomicCAS(&a.status[1], -1, g);
                }
                // stage A of the received-power profile, lane by lane: the few groups of bins that can hold its maximum
                if (a.exact_math) nw = sg_power_plan<BLOCK, true, NB, LMAX>(S, a.rgrid, s_a1, s_a2, s_rho, s_work, ltid, best, k_best);
                else nw = sg_power_plan<BLOCK, false, NB, LMAX>(S, a.rgrid, s_a1, s_a2, s_rho, s_work, ltid, best, k_best);
            } else S = 0;
        }
        // stage B, the whole wave over the groups of its 64 beams
        {
            const int colbase = BLOCK < 64 ? 0 : (tid & ~63);
            if (a.exact_math) sg_wave_eval<BLOCK, true, NB>(nw, S, a.rgrid, s_a1, s_a2, s_rho, s_work, colbase, best, k_best);
            else sg_wave_eval<BLOCK, false, NB>(nw, S, a.rgrid, s_a1, s_a2, s_rho, s_work, colbase, best, k_best);
        }
        if (live) {
            uint32_t rec = 0;
            if (S > 0) {
                sg_beam_decide(d, ch, a.las, best, k_best, o);
                rec = sg_pack_record(o);
            }
            if (LISTQ) a.rec[g] = rec;
            else a.rec_q[slot] = rec;                   // the row's record points here (SG_REC_SLOT)
        }
        sg_add_diff2(a.diff2, live, f, live ? (long long)o.diff2 : 0);
    }
}

// PERSISTENT WAVES: the grid is what the chip holds at once, and every wave strides over the items on its own (no block
// barrier anywhere).  A queue of many short items keeps few live waves resident if each item is its own block -- blocks
// that turn out empty, and blocks that wait for their slowest wave, hold the LDS the next ones need -- and the kernel is
// then bound by the latency of its first loads.  (Requesting the slot data of a wave's NEXT item before it computes the
// current one was measured too: since phase 2 moved here the registers that costs outweigh the latency it hides.)
template <typename T, int LMAX, int BLOCK, bool LISTQ>
__global__ __launch_bounds__(BLOCK < 64 ? 64 : BLOCK, LMAX <= 4 ? SG_KP_WAVES : (LMAX <= 16 ? SG_KP_WAVES_TIERS : 1)) void k_power(SgBeamArgs a)
{
    constexpr int THREADS = BLOCK < 64 ? 64 : BLOCK, WAVES = THREADS / 64, LANES = BLOCK < 64 ? BLOCK : 64;
    constexpr int WIN = BLOCK < 64 ? 1 : SG_KP_WIN;       // waves' worth of slots per work item
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (LMAX > 16) __builtin_amdgcn_s_setprio(3);   // (the rare long-list class: see k_tier_scan_direct)
    const int tid = threadIdx.x;
    int64_t work_n = 0, work_off = 0;
    int n_items;
    const int step = (int)gridDim.x * WAVES;
    // A class that the resident waves take in ONE round of single-wave items gains nothing from windows (a wave would walk
    // three rounds while two thirds of the chip idle): small batches and the rare long-list tiers run one item per wave.
    int item_slots = LANES * WIN;
    if (LISTQ) {
        work_n = a.tier_info[a.cls];
        if (work_n > a.work_hi) work_n = a.work_hi;
        work_off = (int64_t)a.cls * a.tier_stride;
        if (WIN > 1 && work_n <= (int64_t)step * LANES) item_slots = LANES;
        n_items = (int)((work_n + item_slots - 1) / item_slots);
    } else {
        n_items = *a.pw_count;
    }
    // (the item index is wave-uniform; said so explicitly, it and everything read through it live in scalar registers)
    for (int i = (int)blockIdx.x * WAVES + __builtin_amdgcn_readfirstlane(tid >> 6); i < n_items; i += step) {
        int start, cnt, item_f = -1;
        if (LISTQ) {
            start = i * item_slots;
            cnt = (int)(work_n - start < item_slots ? work_n - start : item_slots);
        } else {
            const int2 d = a.pw_items[i];
            start = __builtin_amdgcn_readfirstlane(d.x);
            const int dy = __builtin_amdgcn_readfirstlane(d.y);
            cnt = dy & 1023; item_f = (dy >> 10) - 1;
        }
        sg_kp_item<T, LMAX, BLOCK, LISTQ>(a, smem, a.ov_list, a.tq_unsorted, work_off, start, cnt, item_f);
    }
}

// ------------------------------------------------------------------------------------------------
// ONE work queue for everything k_power_few left (large batches).  Rounds 2 - 5 ran k_power<4> (main queue, half of every CU), k_power<8> and
// k_power<16> (overflow slots) as three persistent kernels on three streams: each sized its grid for a chip of its own, and whichever got
// to a CU first kept it -- the 16-entry class' waves lived 0.1 ms of a kernel 1 ms long, the rest of which it queued for CUs
// (profiles/r05_timeline_one_step.txt).  Here one grid of one-wave blocks takes the items of all three from one item space, longest
// lists first (a long item started last is the tail of the phase):
//     [0, it16)            the 16-entry class: L16 lanes per item (3 x 17 x L16 doubles of LDS: 19.6 KB at 48 lanes -- eight waves per CU)
//     [it16, it16 + it8)   the 8-entry class: windows of SG_KP_WIN x 64 entries, taken in order of flake count
//     [.., + it4)          the multi-flake beams of the main queue: windows of the CLOSED-UP back list (SgBeamArgs::back_list) -- per
//                          region a back run is ~80 slots at C2, a round of 64 and a round of 16 lanes: a third of k_power<4>'s rounds
//                          ran a quarter full
// Items by striding, or (ticket) by an atomic cursor: a wave that drew short items takes more of them.
struct SgKpAll {
    int32_t cls8, cls16;         // index of the class in the tier lists, -1: none
    int32_t ticket;
};

template <typename T, int L16>
__global__ __launch_bounds__(64, 2) void k_power_all(SgBeamArgs a, SgKpAll u)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WIN = SG_KP_WIN;
    const int n_waves = (int)gridDim.x;
    const int lane = (int)(threadIdx.x & 63);
    int n16 = u.cls16 >= 0 ? a.tier_info[u.cls16] : 0, n8 = u.cls8 >= 0 ? a.tier_info[u.cls8] : 0;
    if (n16 > a.work_hi) n16 = a.work_hi;
    if (n8 > a.work_hi) n8 = a.work_hi;
    const int n4 = a.back_list ? a.pw_count[2] : 0;
    // (a class the resident waves take in one round of single-wave items gains nothing from windows: see k_power)
    const int slots8 = n8 <= n_waves * 64 ? 64 : 64 * WIN, slots4 = n4 <= n_waves * 64 ? 64 : 64 * WIN;
    const int it16 = (n16 + L16 - 1) / L16, it8 = (n8 + slots8 - 1) / slots8, it4 = (n4 + slots4 - 1) / slots4;
    const int total = it16 + it8 + it4;
    int i = (int)blockIdx.x;
    if (u.ticket) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&a.pw_count[3], 1);
        i = __builtin_amdgcn_readfirstlane(t);
    }
    while (i < total) {
        if (i < it16) {
            const int start = i * L16;
            sg_kp_item<T, 16, L16, true>(a, smem, 1, 0, (int64_t)u.cls16 * a.tier_stride, start, n16 - start < L16 ? n16 - start : L16, -1);
        } else if (i < it16 + it8) {
            const int start = (i - it16) * slots8;
            sg_kp_item<T, 8, 64, true>(a, smem, 1, 0, (int64_t)u.cls8 * a.tier_stride, start, n8 - start < slots8 ? n8 - start : slots8, -1);
        } else {
            const int start = (i - it16 - it8) * slots4;
            sg_kp_item<T, 4, 64, false>(a, smem, 0, 0, 0, start, n4 - start < slots4 ? n4 - start : slots4, -1, a.back_list);
        }
        if (u.ticket) {
            int t = 0;
            if (lane == 0) t = atomicAdd(&a.pw_count[3], 1);
            i = __builtin_amdgcn_readfirstlane(t);
        } else {
            i += n_waves;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The beams of the pass over all rows that met only a FEW flakes -- up to N; six of ten beams that met any met one, eight of ten
// one or two, at 2.5 mm/h -- in their own kernel (sg_few.h): no flake lists in LDS, two thirds of k_power's registers, twice its
// waves per SIMD.  They fill the front of every region's slice of the queue, and k_power is left with the beams that met more.
// Items from k_power_plan: up to 64 neighbouring queue slots of one region each; persistent waves stride over them.  A beam has one
// or two bins to evaluate when its scatterers' windows are apart and up to a dozen when they overlap, so the wave numbers the
// (beam, bin) pairs of its 64 beams by a prefix sum and takes them 64 at a time, whichever beam they belong to (the owner's
// scatterers through LDS); a segmented reduction over the lanes folds each beam's bins (first maximum, simulation.py:151).
// Measured (256 sweeps, same box): N = 2: C2 4.53 -> 4.34 ms, C4 10.6 -> 10.2 ms, C2far 9.05 -> 9.22 ms; N = 1: 4.53, 10.6, 9.05 (and
// 9.58 / 11.03 without any such kernel); N = 3: 4.32, 9.77, 9.48 (and C1: 8.98 without, 8.58 with N = 2, 8.78 with N = 3).
// Tried and dropped: the kernel beside k_power on another stream, two or three blocks per CU instead of four (all 1 - 10 % slower:
// the persistent kernels of this phase do best when each has the chip to itself for its turn).
template <typename T, int N>
__global__ __launch_bounds__(256, 4) void k_power_few(SgBeamArgs a, int qplanes)   // (N = 3: 127 registers and eight spilled; at three waves per SIMD, 143 registers, it was slower on every workload)
{
    __shared__ double s_amp[N + 1][256], s_rho[N + 1][256], s_best[256];      // scatterer N: the hard target
    __shared__ int s_win[N + 1][256], s_zone[N + 1][256], s_k[256];           // k0 | k1 << 16;  first bin | bins << 16
    __shared__ int s_mark[256];                                               // owner marks of a trip of the pair loop (sg_pair_owner)
    const int tid = (int)threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    const int n_items = a.pw_count[1];
    const int step = (int)gridDim.x * 4;
    for (int i = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6); i < n_items; i += step) {
        const int2 it = a.pw_items1[i];
        const int start = __builtin_amdgcn_readfirstlane(it.x), dy = __builtin_amdgcn_readfirstlane(it.y);
        const int cnt = dy & 1023, item_f = (dy >> 10) - 1;
        const bool live = lane < cnt;
        const int64_t slot = (int64_t)start + lane;
        SgBeamOut o;
        o.overflow = 0; o.range_error = 0; o.diff2 = 0.0; o.has_power = 0; o.n_flakes = 0; o.n_hits = 0; o.label = 0; o.new_i = 0; o.k_best = 0;
        int f = 0, ch = 0, S = 0, g = 0;
        SgFew<N> P{};
        int zone[N + 1], n = 0;
#pragma unroll
        for (int z = 0; z <= N; ++z) zone[z] = 0;
        if (live) {
            const double *q = a.dq + (slot >> 6) * (int64_t)(qplanes * 64) + (slot & 63);
            const unsigned sc = a.dq_sc[slot];
            const double d = q[0], tc = q[64];
            double a1[N], a2[N], rho[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {             // every plane of the slot exists, filled or not: all loads in one flight
                a1[j] = q[(2 + 3 * j) * 64]; a2[j] = q[(3 + 3 * j) * 64]; rho[j] = q[(4 + 3 * j) * 64];
            }
            ch = (int)(sc >> 8);
            g = a.dq_g[slot];
            f = item_f >= 0 ? item_f : sg_frame_of(a, g);
            S = sg_few_prep<T, N>(d, tc, (int)(sc & 255u), a1, a2, rho, ch, a.las, a.beam_div_deg, P, o);
            if (S) {                                  // stage A: the bins of every scatterer's window that can hold the maximum
                int ka, kb;
                sg_few_zone<N, 0>(P, 0.0, ka, kb);
                if (kb >= ka) { zone[0] = ka | ((kb - ka + 1) << 16); n += kb - ka + 1; }
                if constexpr (N >= 2) { sg_few_zone<N, 1>(P, 0.0, ka, kb); if (kb >= ka) { zone[1] = ka | ((kb - ka + 1) << 16); n += kb - ka + 1; } }
                if constexpr (N >= 3) { sg_few_zone<N, 2>(P, 0.0, ka, kb); if (kb >= ka) { zone[2] = ka | ((kb - ka + 1) << 16); n += kb - ka + 1; } }
                sg_few_zone<N, N>(P, 0.0, ka, kb);
                if (kb >= ka) { zone[N] = ka | ((kb - ka + 1) << 16); n += kb - ka + 1; }
            }
        }
#pragma unroll
        for (int t = 0; t < N; ++t) { s_amp[t][tid] = P.amp[t]; s_rho[t][tid] = P.rho[t]; s_win[t][tid] = P.k0[t] | (P.k1[t] << 16); }
        s_amp[N][tid] = P.tamp; s_rho[N][tid] = P.d; s_win[N][tid] = P.tk0 | (P.tk1 << 16);
#pragma unroll
        for (int z = 0; z <= N; ++z) s_zone[z][tid] = zone[z];
        s_best[tid] = 0.0; s_k[tid] = 0;
        asm volatile("" ::: "memory");                // written and read by the lanes of one wave: LDS keeps a wave's operations in order
        const int incl = sg_wave_incl_add(n);
        const int excl = incl - n;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        for (int base = 0; base < total; base += 64) {
            const int p = base + lane;
            const bool valid = p < total;
#if !defined(SG_FEW_OWNER_SEARCH)
            const int ow = sg_pair_owner(s_mark, tid, base, excl, incl);   // first lane whose inclusive count exceeds p
#else
            int lo = 0, hi = 63;                      // owner = first lane whose inclusive count exceeds p
            for (int s6 = 0; s6 < 6; ++s6) {
                const int mid = (lo + hi) >> 1;
                const int v = __shfl(incl, mid);
                if (v > p) hi = mid; else lo = mid + 1;
            }
            const int ow = lo & 63;
#endif
            int j = p - __shfl(excl, ow);
            double sm = -1.0;
            int kk = 0x7fffffff, oo = 64 + lane;      // (a lane without a pair: a run of its own)
            if (valid) {
                const int col = wbase + ow;
                int k = 0;
                bool found = false;
#pragma unroll
                for (int z = 0; z <= N; ++z) {        // which zone of its beam the pair falls in
                    const int zc = s_zone[z][col], zn = (int)((unsigned)zc >> 16);
                    if (!found) { if (j < zn) { k = (zc & 0xffff) + j; found = true; } else j -= zn; }
                }
                double amp[N], rho[N];
                int k0[N], k1[N];
#pragma unroll
                for (int t = 0; t < N; ++t) {
                    amp[t] = s_amp[t][col]; rho[t] = s_rho[t][col];
                    const int w = s_win[t][col];
                    k0[t] = w & 0xffff; k1[t] = (int)((unsigned)w >> 16);
                }
                const int tw = s_win[N][col];
                sm = a.exact_math ? sg_few_bin<true, N>(amp, rho, k0, k1, s_amp[N][col], s_rho[N][col], tw & 0xffff, (int)((unsigned)tw >> 16), k, a.rgrid)
                                  : sg_few_bin<false, N>(amp, rho, k0, k1, s_amp[N][col], s_rho[N][col], tw & 0xffff, (int)((unsigned)tw >> 16), k, a.rgrid);
                kk = k; oo = ow;
            }
            // the pairs of one beam are neighbours: fold them towards the last lane of the run (larger sum; equal sums: smaller bin)
            sg_fold_runs(sm, kk, oo);
            const int on = sg_dpp<0x130, 0xf>(oo, oo);      // wave_shl:1 -- the next lane's owner (lane 63: its own, and the test below knows)
            if (valid && (lane == 63 || on != oo)) {  // ... which folds it into the beam's cell (its bins may come in two rounds)
                volatile double *vb = s_best;
                volatile int *vk = s_k;
                const int col = wbase + oo;
                const double b0 = vb[col];
                const int kc = vk[col];
                if (sm > b0 || (sm == b0 && kk < kc)) { vb[col] = sm; vk[col] = kk; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (live) {
            uint32_t rec = 0;
            if (S) {
                const double best = ((volatile double *)s_best)[tid];
                const int k_best = ((volatile int *)s_k)[tid];
                if (o.range_error) {
                    atomicCAS(&a.status[0], 0, 4 /* SNOWGPU_E_RANGE */);
                    atomicCAS(&a.status[1], -1, g);
                }
                sg_beam_decide(P.d, ch, a.las, best, k_best, o);
                rec = sg_pack_record(o);
            }
            a.rec_q[slot] = rec;
        }
        sg_add_diff2(a.diff2, live, f, live ? (long long)o.diff2 : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next item overwrites the cells
    }
}

// ------------------------------------------------------------------------------------------------
// The global-list tier: beams that met more flakes than the largest LDS list holds (SNOWGPU_MAX_FLAKES_PER_BEAM).  The
// reference's lists are unbounded (simulation.py:413-419); here each lane keeps its lists in global memory, h_cap + 1
// entries per column, strided by the lane count -- the same code as the LDS tiers, slower, and only ever run for the
// handful of beams that need it.  h_lanes lanes stride over the class.
template <typename T>
__global__ __launch_bounds__(64) void k_beams_huge(SgBeamArgs a)
{
    const int tid = (int)(blockIdx.x * 64 + threadIdx.x);             // lane of the tier = column of the lists
    const int rstride = a.h_lanes;
    const size_t col = (size_t)(a.h_cap + 1) * (size_t)a.h_lanes;
    double *s_a1 = a.h_lists, *s_a2 = s_a1 + col, *s_rho = s_a2 + col, *s_ratio = s_rho + col;
    const int n_las = a.las->n;
    int64_t work_n = a.tier_info[a.cls];
    const int64_t work_off = (int64_t)a.cls * a.tier_stride;
    int64_t chunk = (int64_t)blockIdx.x * 64;
    if (chunk >= work_n) return;
    do {
    const bool live = chunk + threadIdx.x < work_n;
    int64_t g = 0;
    int f = 0;
    SgBeamOut o;
    o.overflow = 0; o.range_error = 0; o.diff2 = 0.0; o.has_power = 0; o.n_flakes = 0; o.n_hits = 0; o.label = 0; o.new_i = 0; o.k_best = 0;
    if (live) {
        g = a.tier_list[work_off + chunk + threadIdx.x];
        f = sg_frame_of(a, g);
        const T *row = sg_row<T>(a, f, g);
        const T px = row[0], py = row[1], pz = row[2];
        const int ch = (int)row[4];                                    // a flagged beam was simulated: valid channel
        const SgTable tab = a.frame_tables[(int64_t)f * n_las + ch];
        int32_t *dc = a.dbg_count ? a.dbg_count + g : nullptr;
        double *drj = a.dbg_count ? a.dbg_rj + g * a.dbg_cap : nullptr;
        double *dra = a.dbg_count ? a.dbg_ratio + g * a.dbg_cap : nullptr;
        sg_beam<T, 0, 0>(px, py, pz, ch, tab, a.las, a.beam_div_deg, s_a1, s_a2, s_rho, s_ratio, tid, o, a.dbg_cap, dc, drj,
                                dra, a.exact_math != 0, rstride, a.h_cap);
        uint32_t rec = 0;
        if (o.overflow) {
            atomicCAS(&a.status[0], 0, 6 /* SNOWGPU_E_OVERFLOW */);
            atomicCAS(&a.status[1], -1, (int32_t)g);
        } else if (o.range_error) {
            atomicCAS(&a.status[0], 0, 4 /* SNOWGPU_E_RANGE */);
            atomicCAS(&a.status[1], -1, (int32_t)g);
        }
        if (o.has_power) {
            double best = 0.0;
            int k_best = 0;
            if (a.exact_math) sg_lane_power<0, true, 8, 0>(o.n_flakes, a.rgrid, s_a1, s_a2, s_rho, s_ratio, tid, best, k_best, rstride, a.h_cap);
            else sg_lane_power<0, false, 8, 0>(o.n_flakes, a.rgrid, s_a1, s_a2, s_rho, s_ratio, tid, best, k_best, rstride, a.h_cap);
            T d_t;
            if constexpr (SgReal<T>::is_f32) d_t = sqrtf((px * px + py * py) + pz * pz);
            else d_t = sqrt((px * px + py * py) + pz * pz);
            sg_beam_decide((double)d_t, ch, a.las, best, k_best, o);
            rec = sg_pack_record(o);
        }
        a.rec[g] = rec;
    }
    sg_add_diff2(a.diff2, live, f, live ? (long long)o.diff2 : 0);
    } while ((chunk += (int64_t)gridDim.x * 64) < work_n);
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
                                                   unsigned long long *__restrict__ tbl_cnt, const SgTable *__restrict__ tables,
                                                   SgTable *__restrict__ resolved)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_frames * 256) return;
    if (resolved && (p & 255) < n_las) {              // table descriptor of (frame, channel): what k_resolve_tables does, one launch less
        const int64_t i = (int64_t)(p >> 8) * n_las + (p & 255);
        const int t = table_ids[i];
        SgTable d{};
        if (t >= 0 && t < n_tables) d = tables[t];
        resolved[i] = d;
    }
    const SgPair r = sg_pair(p, frame_off, tile_base, max_tiles, table_ids, n_las, n_tables);
    if (r.rows > 0) atomicAdd(&tbl_cnt[(size_t)r.key * SG_TBL_STRIDE], (1ull << 32) | (unsigned long long)((r.rows + blk - 1) / blk));
}

// exclusive scan of the packed per-table counts (both halves at once: neither overflows 32 bits); leaves the counts zero
// so that k_seg_place can use them as cursors
__global__ __launch_bounds__(1024) void k_seg_scan(unsigned long long *__restrict__ tbl_cnt, unsigned long long *__restrict__ tbl_base, int n,
                                                   int32_t *__restrict__ seg_n, int32_t *__restrict__ one_chunk_blk)
{
    __shared__ unsigned long long sc[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024, b0 = t * per, b1 = b0 + per < n ? b0 + per : n;
    unsigned long long sum = 0;
    for (int k = b0; k < b1; ++k) sum += tbl_cnt[(size_t)k * SG_TBL_STRIDE];
    sc[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const unsigned long long add = t >= d ? sc[t - d] : 0; __syncthreads(); sc[t] += add; __syncthreads(); }
    unsigned long long run = sc[t] - sum;
    for (int k = b0; k < b1; ++k) { const unsigned long long c = tbl_cnt[(size_t)k * SG_TBL_STRIDE]; tbl_base[k] = run; run += c; tbl_cnt[(size_t)k * SG_TBL_STRIDE] = 0; }
    if (t == 1023) {
        seg_n[0] = (int32_t)(sc[1023] >> 32); seg_n[1] = (int32_t)(sc[1023] & 0xffffffffull);
        if (one_chunk_blk) { one_chunk_blk[0] = 0; one_chunk_blk[1] = (int32_t)(sc[1023] & 0xffffffffull); }   // the pass as ONE launch: all blocks
    }
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
    const unsigned long long c = atomicAdd(&tbl_cur[(size_t)r.key * SG_TBL_STRIDE], (1ull << 32) | (unsigned long long)nb), base = tbl_base[r.key];
    const int slot = (int)(base >> 32) + (int)(c >> 32);
    const int b0 = (int)(base & 0xffffffffull) + (int)(c & 0xffffffffull);
    seg_start[slot] = r.start; seg_cnt[slot] = r.rows; seg_frame[slot] = (p >> 8) | ((p & 255) << 22); seg_blk[slot] = b0;
    for (int q = 0; q < nb; ++q) {                               // block -> what k_beams needs of its segment: one round trip per block there
        int32_t *br = seg_of_blk + (int64_t)(b0 + q) * SG_BLKREC;
        br[0] = slot; br[1] = (int32_t)r.start; br[2] = r.rows; br[3] = (p >> 8) | ((p & 255) << 22); br[4] = b0;
    }
}

// The three kernels above as ONE block for batches of up to four frames (1024 (frame, channel) pairs) and up to SG_SEG_SMALL_TABLES
// tables: per-table counts, their scan and the placement through LDS, and on the way the fill that clears everything the step counts up
// from zero (`zero`, n_zero 64-bit words).  A small batch is bound by its chain of dependent launches: this is one link instead of five
// (fill, three kernels, fill) and it runs on the caller's stream, so the scan needs no hop to a side stream and back (55 us between the
// end of the sort and the start of the scan in a single sweep's trace, ~10 us now).  Same segments as the three kernels build (the order
// inside a table is whatever the atomics give, there as here).
#define SG_SEG_SMALL_TABLES 4096
__global__ __launch_bounds__(1024) void k_seg_small(const int64_t *__restrict__ frame_off, int n_frames, const int32_t *__restrict__ tile_base,
                                                    int64_t max_tiles, const int32_t *__restrict__ table_ids, int n_las, int n_tables, int blk,
                                                    int64_t *__restrict__ seg_start, int32_t *__restrict__ seg_cnt, int32_t *__restrict__ seg_frame,
                                                    int32_t *__restrict__ seg_blk, int32_t *__restrict__ seg_of_blk, int32_t *__restrict__ seg_n,
                                                    int32_t *__restrict__ one_chunk_blk, const SgTable *__restrict__ tables, SgTable *__restrict__ resolved,
                                                    unsigned long long *__restrict__ zero, int64_t n_zero)
{
    __shared__ unsigned long long cnt[SG_SEG_SMALL_TABLES + 1], sc[1024];
    const int t = threadIdx.x, p = t;
    for (int64_t i = t; i < n_zero; i += 1024) zero[i] = 0ull;
    for (int i = t; i <= n_tables; i += 1024) cnt[i] = 0ull;
    __syncthreads();
    const bool mine = p < n_frames * 256;
    SgPair r{};
    int nb = 0;
    if (mine) {
        if ((p & 255) < n_las) {                      // table descriptor of (frame, channel)
            const int64_t i = (int64_t)(p >> 8) * n_las + (p & 255);
            const int id = table_ids[i];
            SgTable d{};
            if (id >= 0 && id < n_tables) d = tables[id];
            resolved[i] = d;
        }
        r = sg_pair(p, frame_off, tile_base, max_tiles, table_ids, n_las, n_tables);
        nb = (r.rows + blk - 1) / blk;
        if (r.rows > 0) atomicAdd(&cnt[r.key], (1ull << 32) | (unsigned long long)nb);
    }
    __syncthreads();
    // exclusive scan of the packed per-table counts (segments << 32 | blocks)
    const int n = n_tables + 1, per = (n + 1023) / 1024, b0 = t * per, b1 = b0 + per < n ? b0 + per : n;
    unsigned long long sum = 0;
    for (int k = b0; k < b1; ++k) sum += cnt[k];
    sc[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const unsigned long long add = t >= d ? sc[t - d] : 0; __syncthreads(); sc[t] += add; __syncthreads(); }
    unsigned long long run = sc[t] - sum;
    for (int k = b0; k < b1; ++k) { const unsigned long long c = cnt[k]; cnt[k] = run; run += c; }     // cnt: now the table's base, bumped below as its cursor
    if (t == 1023) {
        seg_n[0] = (int32_t)(sc[1023] >> 32); seg_n[1] = (int32_t)(sc[1023] & 0xffffffffull);
        one_chunk_blk[0] = 0; one_chunk_blk[1] = (int32_t)(sc[1023] & 0xffffffffull);
    }
    __syncthreads();
    if (mine && r.rows > 0) {
        const unsigned long long c = atomicAdd(&cnt[r.key], (1ull << 32) | (unsigned long long)nb);
        const int slot = (int)(c >> 32), bb = (int)(c & 0xffffffffull);
        seg_start[slot] = r.start; seg_cnt[slot] = r.rows; seg_frame[slot] = (p >> 8) | ((p & 255) << 22); seg_blk[slot] = bb;
        for (int q = 0; q < nb; ++q) {
            int32_t *br = seg_of_blk + (int64_t)(bb + q) * SG_BLKREC;
            br[0] = slot; br[1] = (int32_t)r.start; br[2] = r.rows; br[3] = (p >> 8) | ((p & 255) << 22); br[4] = bb;
        }
    }
}

// get_fov_flag(calib.lidar_to_rect(xyz), (h, w), calib) (simulation.py:39-47, :535-536) in float64, fixed operation order.
// The projection is OpenPCDet's (pcdet/utils/calibration_kitti.py, the reference's un-vendored submodule lib/OpenPCDet:
// parity unpinned, SURVEY 8 c): rect_to_img divides the image coordinates by the RECTIFIED point's z, not by the third
// homogeneous coordinate -- the two differ by P2[2][3], which real KITTI files carry (~ 3e-3) --, and the depth is that
// coordinate minus P2[2][3].
__device__ __forceinline__ bool sg_in_fov(const SgFov &v, double x, double y, double z)
{
    double r[3];
    for (int j = 0; j < 3; ++j) r[j] = ((x * v.m[j] + y * v.m[3 + j]) + z * v.m[6 + j]) + v.m[9 + j];
    double h[3];
    for (int j = 0; j < 3; ++j) h[j] = ((r[0] * v.p[4 * j] + r[1] * v.p[4 * j + 1]) + r[2] * v.p[4 * j + 2]) + v.p[4 * j + 3];
    const double u = h[0] / r[2], w = h[1] / r[2];
    const double depth = h[2] - v.p[11];
    return u >= 0 && u < v.img_w && w >= 0 && w < v.img_h && depth >= 0;
}

// per frame: tile offsets of the kept rows and the statistics (simulation.py:522-530).  diff2 (per frame: twice the intensity-
// difference sum of the attenuated beams, final once the per-beam kernels are through) may be null: the pre-augment crop has none.
// One WAVE (all 64 lanes call it): lane l takes tiles l, l + 64, .. -- the counts come in one round of loads per 64 tiles and are summed by
// shuffles (one thread walking the tiles waited for every load in turn: 21 us for a sweep's 47 tiles, a tenth of a single sweep's chain).
// tile_cnt / tile_mv may have been written by other blocks of the running launch (k_compact_count's last block): read past the L1.
__device__ __forceinline__ void sg_compact_scan_frame(int f, int64_t n, const int32_t *tile_cnt, int32_t *__restrict__ tile_base, int64_t *__restrict__ out_counts,
                                                      int64_t *__restrict__ out_stats, const unsigned long long *diff2, int64_t max_tiles,
                                                      const int32_t *tile_mv, int32_t *__restrict__ tile_mv_base, int64_t *__restrict__ out_mv_counts)
{
    const int lane = threadIdx.x & 63;
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    const volatile int32_t *vc = tile_cnt + (int64_t)f * max_tiles, *vm = tile_mv ? tile_mv + (int64_t)f * max_tiles : nullptr;
    int run = 0, mrun = 0;
    int64_t att = 0;
    for (int64_t t0 = 0; t0 < tiles; t0 += 64) {
        const int64_t t = t0 + lane;
        const int c = t < tiles ? vc[t] : 0;
        const int m = (vm && t < tiles) ? vm[t] : 0;
        const int kc = c & 0xffff;
        int ik = kc, im = m, ia = c >> 16;       // inclusive prefix of the kept / moved counts, total of the attenuated
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(ik, o), b = __shfl_up(im, o);
            if (lane >= o) { ik += a; im += b; }
            ia += __shfl_xor(ia, o);
        }
        if (t < tiles) {
            tile_base[(int64_t)f * max_tiles + t] = run + ik - kc;
            if (vm) tile_mv_base[(int64_t)f * max_tiles + t] = mrun + im - m;
        }
        run += __shfl(ik, 63); mrun += __shfl(im, 63); att += ia;
    }
    if (lane != 0) return;
    if (out_mv_counts) out_mv_counts[f] = mrun;
    out_counts[f] = run;
    out_stats[f * 3 + 0] = att;              // num_attenuated (:525)
    out_stats[f * 3 + 1] = n - run;          // num_removed (simulation.py:522, + the camera crop :538)
    const double diff_sum = diff2 ? (double)(long long)((const volatile unsigned long long *)diff2)[f] / 2.0 : 0.0;
    out_stats[f * 3 + 2] = att > 0 ? (int64_t)(diff_sum / (double)att) : 0;   // :527-530 int()
}

__global__ __launch_bounds__(64) void k_compact_scan(const int64_t *__restrict__ frame_off,
                                                     const int32_t *__restrict__ tile_cnt,
                                                     int32_t *__restrict__ tile_base, int64_t *__restrict__ out_counts,
                                                     int64_t *__restrict__ out_stats, const unsigned long long *__restrict__ diff2, int64_t max_tiles,
                                                     const int32_t *__restrict__ tile_mv, int32_t *__restrict__ tile_mv_base, int64_t *__restrict__ out_mv_counts)
{
    const int f = blockIdx.x;
    sg_compact_scan_frame(f, frame_off[f + 1] - frame_off[f], tile_cnt, tile_base, out_counts, out_stats, diff2, max_tiles, tile_mv, tile_mv_base, out_mv_counts);
}

// Stable compaction of kept rows, per frame.  keep byte: bit 0 = row is in the output, bit 1 = row passed the noise filter
// (num_attenuated counts those, before the camera crop: simulation.py:525 precedes :532-540).
template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_compact_count(const T *__restrict__ rows_in, const T *__restrict__ srows, const int32_t *__restrict__ frame_unsorted,
                                                            const uint32_t *__restrict__ rec,
                                                            const uint32_t *__restrict__ rec_q, const T *__restrict__ rng, const double *__restrict__ thr_poly,
                                                            uint8_t *__restrict__ keep, const int64_t *__restrict__ frame_off,
                                                            int32_t *__restrict__ tile_cnt, int64_t max_tiles, SgFov fov, int32_t *__restrict__ tile_mv,
                                                            unsigned long long *__restrict__ tiles_done, int32_t *__restrict__ tile_base, int64_t *__restrict__ out_counts,
                                                            int64_t *__restrict__ out_stats, const unsigned long long *__restrict__ diff2,
                                                            int32_t *__restrict__ tile_mv_base, int64_t *__restrict__ out_mv_counts)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) {
        if (tiles_done && blockIdx.x == 0 && threadIdx.x == 0) {       // an empty frame has no tile to complete it: its counts here
            out_counts[f] = 0; out_stats[f * 3 + 0] = 0; out_stats[f * 3 + 1] = 0; out_stats[f * 3 + 2] = 0;
            if (out_mv_counts) out_mv_counts[f] = 0;
        }
        return;
    }
    const T *rows = frame_unsorted[f] ? srows : rows_in;            // sorted position g = row g (see k_sort_scatter)
    const double p0 = thr_poly[(int64_t)f * 3], p1 = thr_poly[(int64_t)f * 3 + 1], p2 = thr_poly[(int64_t)f * 3 + 2];
    int c = 0, mv = 0;                               // mv: kept rows with label 2 (packed result transfer: their coordinates travel apart)
    // the four rows of a thread side by side: records and ranges first, then the records behind queue slots (a dependent gather for a third
    // of the rows), then the decisions -- row after row the kernel was a chain of up to twelve latencies per thread (1.9 TB/s)
    uint32_t rcs[4];
    T dds[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * SG_BLOCK + threadIdx.x;
        rcs[q] = r < n ? rec[base + r] : 0u;
        dds[q] = (r < n && rng != nullptr) ? rng[base + r] : (T)0;      // (unused for rows the pass over all rows did not simulate)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (rcs[q] & SG_REC_SLOT) rcs[q] = rec_q[rcs[q] & ~SG_REC_SLOT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * SG_BLOCK + threadIdx.x;
        if (r >= n) continue;
        // keep = (label == 2) | (intensity > p0 d^2 + p1 d + p2), d the ORIGINAL range, d^2 in the row dtype
        // (simulation.py:465, :469, :518-520)
        const uint32_t rc = rcs[q];
        const int lab_i = (int)((rc >> SG_REC_LABEL_SHIFT) & 3u);
        // Without the camera crop the decision needs the label, the (new or original) intensity and the original range only: a
        // beam the pass over all rows simulated left its range in rng, and the record holds the intensity unless the beam came
        // back unchanged from a later kernel -- those, and rows without a laser, read the row as before.
        const bool from_rec = rng != nullptr && !fov.enabled && !(rc & SG_REC_COPY) && (lab_i != 0 || (rc & SG_REC_HAS_I));
        bool noise_ok, is_att;
        bool k;
        if (from_rec) {
            const T dd = dds[q];
            const T dd2 = dd * dd;
            const double thr = (p0 * (double)dd2 + p1 * (double)dd) + p2;
            noise_ok = (lab_i == 2) || ((double)(T)(int)(rc & 255u) > thr);
            is_att = lab_i == 1;
            k = noise_ok;
        } else {
            const SgRow<T> o = sg_rebuild_row<T>(rows + (base + r) * 5, rc);
            const T dd2 = o.dd * o.dd;
            const double thr = (p0 * (double)dd2 + p1 * (double)o.dd) + p2;
            noise_ok = (o.lab == (T)2) || ((double)o.i > thr);
            is_att = o.lab == (T)1;
            k = noise_ok;
            if (fov.enabled && k) k = sg_in_fov(fov, (double)o.x, (double)o.y, (double)o.z);   // :532-540
        }
        keep[base + r] = (uint8_t)((k ? 1 : 0) | (noise_ok ? 2 : 0));
        c += k;
        mv += (k && lab_i == 2) ? 1 : 0;
        c += (noise_ok && is_att) ? (1 << 16) : 0;                          // high half: rows that count in num_attenuated (:525, before the crop)
    }
    __shared__ int s[4], s2[4], s_last;
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_down(c, o); mv += __shfl_down(mv, o); }
    if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = c; s2[threadIdx.x >> 6] = mv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_cnt[(int64_t)f * max_tiles + blockIdx.x] = s[0] + s[1] + s[2] + s[3];   // kept | attenuated << 16 (a tile has 1024 rows)
        if (tile_mv) tile_mv[(int64_t)f * max_tiles + blockIdx.x] = s2[0] + s2[1] + s2[2] + s2[3];
        // Small batches (tiles_done != null): the block that completes a frame scans its tiles -- what k_compact_scan does as a launch of its
        // own: one link less on the chain.  Not for large batches: the device-scope fence this needs writes back the L2 of the block's XCD
        // (eight XCDs, eight L2s), and 32 768 of them made this kernel 1.37 ms long on 256 sweeps instead of 0.15.
        if (tiles_done) {
            __threadfence();
            const unsigned long long tiles = (unsigned long long)((n + SG_TILE - 1) / SG_TILE);
            s_last = atomicAdd(&tiles_done[f], 1ull) == tiles - 1;
            if (s_last) __threadfence();
        }
    }
    if (tiles_done) {                                 // (kernel argument: uniform)
        __syncthreads();
        if (s_last && threadIdx.x < 64)
            sg_compact_scan_frame(f, n, tile_cnt, tile_base, out_counts, out_stats, diff2, max_tiles, tile_mv, tile_mv_base, out_mv_counts);
    }
}

// PACK (packed result transfer, snowgpu_set_result_transfer): instead of the 5-column output row, per kept row a 4-byte word -- source row
// (30 bits) | label 0 / 1 / 2, or 3 = "column 4 keeps the input's channel value" (Q5) -- and its output intensity (row dtype); the moved
// coordinates of the label-2 rows (simulation.py:176-180), a small minority, go to a list of their own in output order.  The host side of
// the library copies x, y, z (and the channel of code-3 rows) from the caller's INPUT rows: 8 instead of 24 bytes per point cross the link.
struct SgPack { uint32_t *meta; void *inten; void *mv; const int32_t *tile_mv_base; const int64_t *mv_counts; };

template <typename T, bool PACK>
__global__ __launch_bounds__(SG_BLOCK) void k_compact_scatter(const T *__restrict__ rows_in, const T *__restrict__ srows, const int32_t *__restrict__ frame_unsorted,
                                                              const uint32_t *__restrict__ rec,
                                                              const uint32_t *__restrict__ rec_q, const uint8_t *__restrict__ keep, const int32_t *__restrict__ perm,
                                                              const int64_t *__restrict__ frame_off,
                                                              const int32_t *__restrict__ tile_base, T *__restrict__ out_rows,
                                                              int32_t *__restrict__ out_src, int64_t *__restrict__ out_stats,
                                                              int64_t max_tiles, SgPack pk)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const bool uns = frame_unsorted[f] != 0;
    const T *rows = uns ? srows : rows_in;
    __shared__ int wave_cnt[4][4];               // [round][wave]
    [[maybe_unused]] __shared__ int wave_mv[4][4];
    const int tid = threadIdx.x, w = tid >> 6;
    bool k[4];
    int pre[4];
    uint32_t rcs[4];
    [[maybe_unused]] int pre_mv[4];
    // keep flags, then the kept rows' records, then the records behind queue slots: each kind for the thread's four rows at once (row after
    // row they were a chain of dependent loads; see k_compact_count)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * SG_BLOCK + tid;
        k[q] = r < n && (keep[base + r] & 1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) rcs[q] = k[q] ? rec[base + tile0 + q * SG_BLOCK + tid] : 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (rcs[q] & SG_REC_SLOT) rcs[q] = rec_q[rcs[q] & ~SG_REC_SLOT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long m = __ballot(k[q]);
        pre[q] = __popcll(m & sg_lanemask_lt());
        if ((tid & 63) == 0) wave_cnt[q][w] = __popcll(m);
        if constexpr (PACK) {
            const unsigned long long mm = __ballot(k[q] && ((rcs[q] >> SG_REC_LABEL_SHIFT) & 3u) == 2u);
            pre_mv[q] = __popcll(mm & sg_lanemask_lt());
            if ((tid & 63) == 0) wave_mv[q][w] = __popcll(mm);
        }
    }
    __syncthreads();
    int run = tile_base[(int64_t)f * max_tiles + blockIdx.x];
    [[maybe_unused]] int64_t mrun = 0;
    if constexpr (PACK) {
        // the moved coordinates of a batch form ONE list in output order (one copy down the link): this frame's part starts behind
        // those of the frames before it (a handful of frames per call: summed here rather than scanned by another launch)
        for (int g = 0; g < f; ++g) mrun += pk.mv_counts[g];
        mrun += pk.tile_mv_base[(int64_t)f * max_tiles + blockIdx.x];
    }
    for (int q = 0; q < 4; ++q) {
        int off = run;
        for (int ww = 0; ww < w; ++ww) off += wave_cnt[q][ww];
        if (k[q]) {
            const int64_t r = base + tile0 + q * SG_BLOCK + tid;
            const int64_t dst = base + off + pre[q];
            const int32_t src = uns ? perm[r] : (int32_t)(r - base);
            if constexpr (PACK) {
                const uint32_t rc = rcs[q];
                const SgRow<T> o = sg_rebuild_row<T>(rows + r * 5, rc);
                const uint32_t label = (rc >> SG_REC_LABEL_SHIFT) & 3u;
                const uint32_t code = (label == 0 && (rc & SG_REC_COPY)) ? 3u : label;
                pk.meta[dst] = (uint32_t)src | (code << 30);
                ((T *)pk.inten)[dst] = o.i;
                if (label == 2) {
                    int64_t moff = mrun + pre_mv[q];
                    for (int ww = 0; ww < w; ++ww) moff += wave_mv[q][ww];
                    T *d = (T *)pk.mv + moff * 3;
                    d[0] = o.x; d[1] = o.y; d[2] = o.z;
                }
            } else {
                const SgRow<T> o = sg_rebuild_row<T>(rows + r * 5, rcs[q]);
                T *d = out_rows + dst * 5;
                d[0] = o.x; d[1] = o.y; d[2] = o.z; d[3] = o.i; d[4] = o.lab;
                out_src[dst] = src;
            }
        }
        run += wave_cnt[q][0] + wave_cnt[q][1] + wave_cnt[q][2] + wave_cnt[q][3];
        if constexpr (PACK) mrun += wave_mv[q][0] + wave_mv[q][1] + wave_mv[q][2] + wave_mv[q][3];
    }
}

// ---- pre-augment camera crop (precompute.py:96-99): pc = pc[get_fov_flag(lidar_to_rect(pc[:, 0:3]), (1024, 1920))] -----
// Stable compaction of the INPUT rows of every frame by the FOV test on their original coordinates: flag + tile counts,
// per-frame scan (k_compact_scan), then scatter to the frame's new offset.
template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_crop_flag(const T *__restrict__ rows, const int64_t *__restrict__ frame_off,
                                                        uint8_t *__restrict__ keep, int32_t *__restrict__ tile_cnt, int64_t max_tiles, SgFov fov)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    int c = 0;
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * SG_BLOCK + threadIdx.x;
        if (r >= n) continue;
        const T *row = rows + (base + r) * 5;
        const bool k = sg_in_fov(fov, (double)row[0], (double)row[1], (double)row[2]);
        keep[base + r] = k ? 1 : 0;
        c += k;
    }
    __shared__ int s[4];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[(int64_t)f * max_tiles + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <typename T>
__global__ __launch_bounds__(SG_BLOCK) void k_crop_scatter(const T *__restrict__ rows, const uint8_t *__restrict__ keep,
                                                           const int64_t *__restrict__ frame_off, const int64_t *__restrict__ new_off,
                                                           const int32_t *__restrict__ tile_base, T *__restrict__ out_rows,
                                                           int32_t *__restrict__ crop_src, int64_t max_tiles)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = frame_off[f + 1] - base;
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    __shared__ int wave_cnt[4][4];
    const int tid = threadIdx.x, w = tid >> 6;
    bool k[4];
    int pre[4];
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * SG_BLOCK + tid;
        k[q] = r < n && keep[base + r];
        const unsigned long long m = __ballot(k[q]);
        pre[q] = __popcll(m & sg_lanemask_lt());
        if ((tid & 63) == 0) wave_cnt[q][w] = __popcll(m);
    }
    __syncthreads();
    int run = tile_base[(int64_t)f * max_tiles + blockIdx.x];
    const int64_t nbase = new_off[f];
    for (int q = 0; q < 4; ++q) {
        int off = run;
        for (int ww = 0; ww < w; ++ww) off += wave_cnt[q][ww];
        if (k[q]) {
            const int64_t r = tile0 + q * SG_BLOCK + tid;
            const T *sr = rows + (base + r) * 5;
            T *d = out_rows + (nbase + off + pre[q]) * 5;
            d[0] = sr[0]; d[1] = sr[1]; d[2] = sr[2]; d[3] = sr[3]; d[4] = sr[4];
            crop_src[nbase + off + pre[q]] = (int32_t)r;
        }
        run += wave_cnt[q][0] + wave_cnt[q][1] + wave_cnt[q][2] + wave_cnt[q][3];
    }
}
// ------------------------------------------------------------------------------------------------
// Compact input (snowgpu_augment_batch_compact): rows that crossed the link as (x, y, z, intensity) float32 + one channel BYTE -- 17 bytes
// per point instead of the STF row's 20 (precompute.py:78 keeps the channel as a fifth float32) -- become the (x, y, z, intensity,
// channel) rows every kernel reads.  One thread per row; the batch's only pass that exists for the link's sake (0.67 GB written per
// 256 sweeps, spread over the chunks of the pipeline).
__global__ __launch_bounds__(256) void k_expand_rows(const float4 *__restrict__ xyzi, const uint8_t *__restrict__ ch, float *__restrict__ rows, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 v = xyzi[i];
    float *r = rows + i * 5;
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; r[4] = (float)ch[i];
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

// ------------------------------------------------------------------------------------------------
// launch wrappers (C linkage, called from snowgpu_api.cpp)

#define SG_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return (int)e__;            \
    } while (0)

extern "C" int sg_launch_expand_rows(const void *xyzi, const uint8_t *ch, void *rows, int64_t n, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_expand_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)xyzi, ch, (float *)rows, n);
    SG_CHECK_LAUNCH();
    return 0;
}

extern "C" int sg_launch_resolve_tables(const SgTable *tables, int n_tables, const int32_t *table_ids, int64_t n, SgTable *out,
                                        void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_resolve_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tables, n_tables,
                       table_ids, n, out);
    SG_CHECK_LAUNCH();
    return 0;
}

// lean_plane / lean_part: optional -- the ground planes (n_frames x 4) and the tile-partials buffer of the noise-threshold prepass
// (sg_prepass_reserve_tiles): the first kernel then leaves the prepass' per-tile statistics on its way over the rows
extern "C" int sg_launch_sort(const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t n_total,
                              int32_t *tile_hist, int32_t *tile_base, uint16_t *rank, uint8_t *ch8, int32_t *perm, int32_t *status,
                              int64_t max_tiles, const double *lean_plane, double *lean_part, int32_t *tile_unsorted, int32_t *frame_unsorted,
                              void *srows, int identity_perm, int phase /* 1: histogram + scan; 2: scatter; 3: both */, void *stream)
{
    (void)n_total;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    SgLeanTile lt{};
    lt.plane = lean_plane; lt.delta = 0.5; lt.part = lean_part; lt.max_tiles = max_tiles;
    if (!(phase & 1)) goto scatter;
    if (lean_plane && lean_part) {
        if (dtype == 0) hipLaunchKernelGGL((k_sort_hist<float, true>), grid, dim3(SG_BLOCK), 0, st, (const float *)rows, frame_off, tile_hist, rank, ch8, status, max_tiles, lt, tile_unsorted);
        else hipLaunchKernelGGL((k_sort_hist<double, true>), grid, dim3(SG_BLOCK), 0, st, (const double *)rows, frame_off, tile_hist, rank, ch8, status, max_tiles, lt, tile_unsorted);
    } else {
        if (dtype == 0) hipLaunchKernelGGL((k_sort_hist<float, false>), grid, dim3(SG_BLOCK), 0, st, (const float *)rows, frame_off, tile_hist, rank, ch8, status, max_tiles, lt, tile_unsorted);
        else hipLaunchKernelGGL((k_sort_hist<double, false>), grid, dim3(SG_BLOCK), 0, st, (const double *)rows, frame_off, tile_hist, rank, ch8, status, max_tiles, lt, tile_unsorted);
    }
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sort_scan, dim3(n_frames), dim3(SG_BLOCK), 0, st, frame_off, tile_hist, tile_base, max_tiles, tile_unsorted, frame_unsorted);
    SG_CHECK_LAUNCH();
scatter:
    if (!(phase & 2)) return 0;
    if (dtype == 0) hipLaunchKernelGGL(k_sort_scatter<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)rows, ch8, frame_off, tile_hist, tile_base, rank, perm, (float *)srows, frame_unsorted, identity_perm, max_tiles);
    else hipLaunchKernelGGL(k_sort_scatter<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)rows, ch8, frame_off, tile_hist, tile_base, rank, perm, (double *)srows, frame_unsorted, identity_perm, max_tiles);
    SG_CHECK_LAUNCH();
    return 0;
}

extern "C" int sg_launch_gather_rows(const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t n_total, int64_t max_frame,
                                     const int32_t *perm, void *srows, int32_t *frame_unsorted, int32_t *status, void *stream)
{
    (void)n_total;
    if (n_frames <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((max_frame + SG_BLOCK - 1) / SG_BLOCK, 256)), (unsigned)n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_gather_rows<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)rows, frame_off, perm, (float *)srows, frame_unsorted, status);
    else hipLaunchKernelGGL(k_gather_rows<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)rows, frame_off, perm, (double *)srows, frame_unsorted, status);
    SG_CHECK_LAUNCH();
    return 0;
}

static int sg_cu_count(int dev_id)
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    return cus > 0 ? cus : 256;
}

template <typename K>
static int sg_set_lds(K kernel, size_t lds, bool *attr_set)
{
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {          // per device: several contexts may live in one process
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
    }
    return 0;
}

template <typename T, int LMAX, int BLOCK, bool LIST, int DICT>
static int launch_beams_t(const SgBeamArgs *a, hipStream_t st)
{
    // (the pass over all rows: ranges + one word per listed flake; a list-mode scan: three double columns -- see k_beams)
    const size_t lds = DICT ? (LIST ? sizeof(double) * (size_t)BLOCK * 3 * (size_t)LMAX : (sizeof(double) + sizeof(uint32_t)) * (size_t)BLOCK * (size_t)LMAX)
                                  + sizeof(int) * (4 * (size_t)(BLOCK < 64 ? 64 : BLOCK) + (size_t)LMAX * BLOCK)
                            : sizeof(double) * (size_t)BLOCK * 4 * ((size_t)LMAX + 1);
    static bool attr_set[64] = {};
    if (int e = sg_set_lds(k_beams<T, LMAX, BLOCK, LIST, DICT>, lds, attr_set)) return e;
    constexpr int THREADS = BLOCK < 64 ? 64 : BLOCK;
    unsigned blocks;
    if (LIST) {                                          // list mode: at most what the chip can hold at once
        const int64_t n = (int64_t)a->work_hi - a->work_lo;
        if (n <= 0) return 0;
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        const unsigned per_cu = (unsigned)std::min<size_t>(32 * 64 / THREADS, std::max<size_t>(1, (size_t)(160 * 1024) / lds));
        blocks = (unsigned)std::min<int64_t>((n + BLOCK - 1) / BLOCK, (int64_t)sg_cu_count(dev_id) * per_cu);
        // the in-place pass only sees the entries beyond the hand-over buffer, i.e. normally none: a small grid (its LDS-heavy
        // blocks would otherwise queue for CU space just to find that out)
        if (!DICT) blocks = std::min(blocks, 128u);
    } else {
        blocks = (unsigned)a->grid_blocks;               // blocks [blk_lo, blk_hi) or chunk a->chunk of the segment order
    }
    if (blocks == 0) return 0;
    hipLaunchKernelGGL((k_beams<T, LMAX, BLOCK, LIST, DICT>), dim3(blocks), dim3(THREADS), lds, st, *a);
    SG_CHECK_LAUNCH();
    return 0;
}

template <typename T, int LMAX, int BLOCK>
static int launch_beams_m(const SgBeamArgs *a, int direct, int dict_only, hipStream_t st)
{
    if (direct) {
        if (a->exact_math) return launch_beams_t<T, LMAX, BLOCK, false, 2>(a, st);   // (registers to spare for tangent, root and quotient in its loop)
        return launch_beams_t<T, LMAX, BLOCK, false, 1>(a, st);
    }
    return dict_only ? launch_beams_t<T, LMAX, BLOCK, true, 1>(a, st) : launch_beams_t<T, LMAX, BLOCK, true, 0>(a, st);
}

template <typename T, int LMAX, int BLOCK, bool LISTQ>
static int launch_power_t(const SgBeamArgs *a, hipStream_t st, bool plan_only = false, hipEvent_t ev_few = nullptr, int which = 3)
{
    const size_t lds = sizeof(double) * ((size_t)BLOCK * ((SG_KP_THREE_MAX > 0 && LMAX <= SG_KP_THREE_MAX && LMAX >= SG_KP_THREE_MIN) ? 3 : 4) * (LMAX + 1));
    static bool attr_set[64] = {};
    if (int e = sg_set_lds(k_power<T, LMAX, BLOCK, LISTQ>, lds, attr_set)) return e;
    constexpr int THREADS = BLOCK < 64 ? 64 : BLOCK, LANES = BLOCK < 64 ? BLOCK : 64;
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    // persistent waves: what the chip holds at once (LDS and the 32-waves-per-CU limit), fewer if the queue cannot be longer
    unsigned per_cu = (unsigned)std::min<size_t>(32 * 64 / THREADS, std::max<size_t>(1, (size_t)(160 * 1024) / lds));
    // the queue of the pass over all rows: its persistent blocks would hold every CU's LDS until they are done, while the
    // later tiers and the prepass run beside it -- it takes a share (in quarters) of what fits
    if (!LISTQ && a->kp_lds_quarters > 0 && a->kp_lds_quarters < 4) per_cu = std::max(1u, per_cu * (unsigned)a->kp_lds_quarters / 4u);
    int64_t blocks = (int64_t)sg_cu_count(dev_id) * per_cu;
    int64_t items_ub = LISTQ ? ((int64_t)a->work_hi + LANES - 1) / LANES : (a->n_total + LANES - 1) / LANES + 2 * a->n_regions_ub;
    if (LISTQ && a->work_hint > 0) items_ub = std::min<int64_t>(items_ub, (4 * (int64_t)a->work_hint + LANES - 1) / LANES + 8);   // (see launch_tier_scan_t)
    blocks = std::min<int64_t>(blocks, (items_ub + THREADS / 64 - 1) / (THREADS / 64));
    if (blocks <= 0) return 0;
    if (!LISTQ) {
        const unsigned pg = (unsigned)((a->n_regions_ub + 255) / 256);
        if (pg == 0) return 0;
        // back runs (beams with several flakes, ~7 % of the rows of a sweep) as windows of SG_KP_WIN waves' worth of slots, taken
        // in order of flake count -- unless the resident waves take them all in about two rounds of single-wave items anyway
        // (small batches: a wave walking a window would serialise what the idle rest of the chip could do at once)
        const int64_t back_est = a->n_total / 10;
        const int lanes_back = (BLOCK < 64 || back_est <= (int64_t)blocks * (THREADS / 64) * LANES * 5 / 2) ? LANES : LANES * SG_KP_WIN;
        if (plan_only) {                              // (its own call: the plan runs on the scan's stream, the kernels it feeds beside the tiers)
            hipLaunchKernelGGL(k_power_plan, dim3(pg), dim3(256), 0, st, *a, LANES, lanes_back, (int)a->n_regions_ub);
            SG_CHECK_LAUNCH();
            return 0;
        }
        if (a->pw_items1 && (which & 1)) {            // the beams with few flakes: ahead of k_power on its stream (beside it, on the
            hipStream_t s1 = st;                      // tiers' stream, was measured: 4.67 instead of 4.59 ms)
            const unsigned g1 = (unsigned)std::min<int64_t>((int64_t)sg_cu_count(dev_id) * 4, (a->n_total / LANES + a->n_regions_ub + 3) / 4);
            if (g1 > 0) {
                if (a->front_max == 1) hipLaunchKernelGGL((k_power_few<T, 1>), dim3(g1), dim3(256), 0, s1, *a, SG_QPLANES(LMAX));
                else if (a->front_max == 2) hipLaunchKernelGGL((k_power_few<T, 2>), dim3(g1), dim3(256), 0, s1, *a, SG_QPLANES(LMAX));
                else hipLaunchKernelGGL((k_power_few<T, 3>), dim3(g1), dim3(256), 0, s1, *a, SG_QPLANES(LMAX));
            }
            SG_CHECK_LAUNCH();
            if (ev_few && hipEventRecord(ev_few, st) != hipSuccess) return (int)hipGetLastError();
        }
    }
    if (!(which & 2)) return 0;
    hipLaunchKernelGGL((k_power<T, LMAX, BLOCK, LISTQ>), dim3((unsigned)blocks), dim3(THREADS), lds, st, *a);
    SG_CHECK_LAUNCH();
    return 0;
}

// the LDS-free scan of a later tier (class a->cls, capacity lmax = 8, 16 or 63): fills the tier's hand-over buffer, flakes in scan order
template <typename T>
static int launch_tier_scan_t(const SgBeamArgs *a, int lmax, hipStream_t st)
{
    const int64_t n = (int64_t)a->work_hi - a->work_lo;
    if (n <= 0) return 0;
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    int64_t want = (n + 255) / 256;
    // (a class that held a handful of beams in the batches before gets a grid for four times that, not one for its buffer: 8192 waves queued
    // for CUs beside the persistent kernels of the phase to find 23 beams -- 0.36 ms; the fixed grid strides, so a low guess only costs time)
    if (a->work_hint > 0) want = std::min<int64_t>(want, (4 * (int64_t)a->work_hint + 255) / 256 + 8);
    const unsigned blocks = (unsigned)std::min<int64_t>(want, (int64_t)sg_cu_count(dev_id) * 8);
    if (lmax == 8) hipLaunchKernelGGL((k_tier_scan_direct<T, 8>), dim3(blocks), dim3(256), 0, st, *a);
    else if (lmax == 16) hipLaunchKernelGGL((k_tier_scan_direct<T, 16>), dim3(blocks), dim3(256), 0, st, *a);
    else hipLaunchKernelGGL((k_tier_scan_direct<T, SG_LCAP>), dim3(blocks), dim3(256), 0, st, *a);
    SG_CHECK_LAUNCH();
    return 0;
}

extern "C" int sg_launch_tier_scan(const SgBeamArgs *a, int dtype, int lmax, void *stream)
{
    return dtype == 0 ? launch_tier_scan_t<float>(a, lmax, (hipStream_t)stream) : launch_tier_scan_t<double>(a, lmax, (hipStream_t)stream);
}

// threads per block of the pass with list capacity lmax (the segment builder counts blocks of this size)
extern "C" int sg_beams_block(int lmax) { return lmax == 4 ? 256 : (lmax == 8 ? SG_LANES_8 : (lmax == 16 ? SG_LANES_16 : SG_LANES_63)); }

// lmax = per-beam list capacity of this pass: 4 (160 B of LDS per beam: 16 waves per CU), 8, 16 or 63 (the largest
// LDS list).  direct: the pass over all rows, dict hand-over to sg_launch_power; else list mode over class a->cls.
extern "C" int sg_launch_beams(const SgBeamArgs *a, int dtype, int lmax, int direct, int dict_only, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        if (lmax == 4) return launch_beams_m<float, 4, 256>(a, direct, dict_only, st);
        if (lmax == 8) return launch_beams_m<float, 8, SG_LANES_8>(a, direct, dict_only, st);
        if (lmax == 16) return launch_beams_m<float, 16, SG_LANES_16>(a, direct, dict_only, st);
        return launch_beams_m<float, SG_LCAP, SG_LANES_63>(a, direct, dict_only, st);
    }
    if (lmax == 4) return launch_beams_m<double, 4, 256>(a, direct, dict_only, st);
    if (lmax == 8) return launch_beams_m<double, 8, SG_LANES_8>(a, direct, dict_only, st);
    if (lmax == 16) return launch_beams_m<double, 16, SG_LANES_16>(a, direct, dict_only, st);
    return launch_beams_m<double, SG_LCAP, SG_LANES_63>(a, direct, dict_only, st);
}

// the received-power kernel for the queue a direct-mode pass of capacity lmax filled
// plan_only = 1: k_power_plan alone (work items of k_power / k_power_few, places of the tier lists' slices); 0: k_power_few and k_power
extern "C" int sg_launch_power(const SgBeamArgs *a, int dtype, int lmax, void *stream, int plan_only, void *ev_few, int which)
{
    hipStream_t st = (hipStream_t)stream;
    const bool po = plan_only != 0;
    hipEvent_t ef = (hipEvent_t)ev_few;
    if (dtype == 0) {
        if (lmax == 4) return launch_power_t<float, 4, 256, false>(a, st, po, ef, which);
        if (lmax == 8) return launch_power_t<float, 8, SG_LANES_8, false>(a, st, po, ef, which);
        if (lmax == 16) return launch_power_t<float, 16, SG_LANES_16, false>(a, st, po, ef, which);
        return launch_power_t<float, SG_LCAP, SG_LANES_63, false>(a, st, po, ef, which);
    }
    if (lmax == 4) return launch_power_t<double, 4, 256, false>(a, st, po, ef, which);
    if (lmax == 8) return launch_power_t<double, 8, SG_LANES_8, false>(a, st, po, ef, which);
    if (lmax == 16) return launch_power_t<double, 16, SG_LANES_16, false>(a, st, po, ef, which);
    return launch_power_t<double, SG_LCAP, SG_LANES_63, false>(a, st, po, ef, which);
}

template <typename T>
static int launch_power_all_t(const SgBeamArgs *a, int cls8, int cls16, int waves_per_cu, int ticket, hipStream_t st)
{
    constexpr int L16 = 48;
    const size_t lds = sizeof(double) * std::max<size_t>(std::max<size_t>((size_t)64 * 4 * 5, (size_t)64 * 3 * 9), (size_t)L16 * 3 * 17);
    static bool attr_set[64] = {};
    if (int e = sg_set_lds(k_power_all<T, L16>, lds, attr_set)) return e;
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const int per_cu = std::max(1, std::min(waves_per_cu, (int)((size_t)(160 * 1024) / lds)));
    const int64_t blocks = std::min<int64_t>((int64_t)sg_cu_count(dev_id) * per_cu, a->n_total / 48 + 3);
    if (blocks <= 0) return 0;
    SgKpAll u{cls8, cls16, ticket};
    hipLaunchKernelGGL((k_power_all<T, L16>), dim3((unsigned)blocks), dim3(64), lds, st, *a, u);
    SG_CHECK_LAUNCH();
    return 0;
}

extern "C" int sg_launch_power_all(const SgBeamArgs *a, int dtype, int cls8, int cls16, int waves_per_cu, int ticket, void *stream)
{
    return dtype == 0 ? launch_power_all_t<float>(a, cls8, cls16, waves_per_cu, ticket, (hipStream_t)stream)
                      : launch_power_all_t<double>(a, cls8, cls16, waves_per_cu, ticket, (hipStream_t)stream);
}

extern "C" int sg_launch_tier_gather(const SgBeamArgs *a, void *stream)
{
    const unsigned g = (unsigned)((a->n_regions_ub + 3) / 4);
    if (g == 0) return 0;
    hipLaunchKernelGGL(k_tier_gather, dim3(g), dim3(256), 0, (hipStream_t)stream, *a, (int)a->n_regions_ub);
    SG_CHECK_LAUNCH();
    return 0;
}

// ... and for the hand-over buffer of a list-mode pass
extern "C" int sg_launch_power_list(const SgBeamArgs *a, int dtype, int lmax, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        if (lmax == 8) return launch_power_t<float, 8, SG_LANES_8, true>(a, st);
        if (lmax == 16) return launch_power_t<float, 16, SG_LANES_16, true>(a, st);
        return launch_power_t<float, SG_LCAP, SG_LANES_63, true>(a, st);
    }
    if (lmax == 8) return launch_power_t<double, 8, SG_LANES_8, true>(a, st);
    if (lmax == 16) return launch_power_t<double, 16, SG_LANES_16, true>(a, st);
    return launch_power_t<double, SG_LCAP, SG_LANES_63, true>(a, st);
}

extern "C" int sg_launch_huge(const SgBeamArgs *a, int dtype, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(a->h_lanes / 64);
    if (blocks == 0) return 0;
    if (dtype == 0) hipLaunchKernelGGL(k_beams_huge<float>, dim3(blocks), dim3(64), 0, st, *a);
    else hipLaunchKernelGGL(k_beams_huge<double>, dim3(blocks), dim3(64), 0, st, *a);
    SG_CHECK_LAUNCH();
    return 0;
}

// the same for a small batch (see k_seg_small); returns -1 if the batch is not small (nothing launched)
extern "C" int sg_launch_segments_small(const int64_t *frame_off, int n_frames, const int32_t *tile_base, int64_t max_tiles, const int32_t *table_ids,
                                        int n_las, int n_tables, int block, int32_t *seg_blk, int64_t *seg_start, int32_t *seg_cnt, int32_t *seg_frame,
                                        int32_t *seg_n, int32_t *seg_of_blk, int32_t *chunk_blk, const SgTable *tables, SgTable *resolved,
                                        unsigned long long *zero, int64_t n_zero, void *stream)
{
    if (n_frames * 256 > 1024 || n_tables + 1 > SG_SEG_SMALL_TABLES || n_zero > (1 << 16)) return -1;
    hipLaunchKernelGGL(k_seg_small, dim3(1), dim3(1024), 0, (hipStream_t)stream, frame_off, n_frames, tile_base, max_tiles, table_ids, n_las, n_tables, block,
                       seg_start, seg_cnt, seg_frame, seg_blk, seg_of_blk, seg_n, chunk_blk, tables, resolved, zero, n_zero);
    SG_CHECK_LAUNCH();
    return 0;
}

extern "C" int sg_launch_segments(const int64_t *frame_off, int n_frames, const int32_t *tile_base, int64_t max_tiles, const int32_t *table_ids,
                                  int n_las, int n_tables, int block, unsigned long long *tbl_cnt, unsigned long long *tbl_base, int32_t *seg_blk,
                                  int64_t *seg_start, int32_t *seg_cnt, int32_t *seg_frame, int32_t *seg_n, int32_t *seg_of_blk,
                                  int32_t *chunk_blk, const SgTable *tables, SgTable *resolved, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)n_frames;         // 256 pairs per frame, one thread each
    if (hipMemsetAsync(tbl_cnt, 0, sizeof(unsigned long long) * ((size_t)n_tables + 1) * SG_TBL_STRIDE, st) != hipSuccess) return (int)hipGetLastError();
    hipLaunchKernelGGL(k_seg_count, dim3(grid), dim3(256), 0, st, frame_off, n_frames, tile_base, max_tiles, table_ids, n_las, n_tables, block, tbl_cnt,
                       tables, resolved);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1024), 0, st, tbl_cnt, tbl_base, n_tables + 1, seg_n, chunk_blk);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_seg_place, dim3(grid), dim3(256), 0, st, frame_off, n_frames, tile_base, max_tiles, table_ids, n_las, n_tables, block,
                       tbl_base, tbl_cnt, seg_start, seg_cnt, seg_frame, seg_blk, seg_of_blk);
    SG_CHECK_LAUNCH();
    return 0;
}

extern "C" int sg_launch_compact(const void *rows, const void *srows, const int32_t *frame_unsorted, int dtype, const uint32_t *rec, const uint32_t *rec_q, const void *rng, const double *thr_poly, uint8_t *keep, const int32_t *perm,
                                 const int64_t *frame_off, int n_frames, int64_t n_total, int32_t *tile_cnt,
                                 int32_t *tile_base, void *out_rows, int32_t *out_src, int64_t *out_counts,
                                 int64_t *out_stats, const unsigned long long *diff2, const SgFov *fov, int64_t max_tiles, const SgPackOut *pack,
                                 unsigned long long *tiles_done /* n_frames words, zero */, void *stream)
{
    (void)n_total;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    SgFov fv{};
    if (fov) fv = *fov;
    int32_t *tile_mv = pack ? pack->tile_mv : nullptr;
    int32_t *tmb = pack ? pack->tile_mv_base : nullptr;
    int64_t *mvc = pack ? pack->mv_counts : nullptr;
    if (dtype == 0) hipLaunchKernelGGL(k_compact_count<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)rows, (const float *)srows, frame_unsorted, rec, rec_q, (const float *)rng, thr_poly, keep, frame_off, tile_cnt, max_tiles, fv, tile_mv,
                                       tiles_done, tile_base, out_counts, out_stats, diff2, tmb, mvc);
    else hipLaunchKernelGGL(k_compact_count<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)rows, (const double *)srows, frame_unsorted, rec, rec_q, (const double *)rng, thr_poly, keep, frame_off, tile_cnt, max_tiles, fv, tile_mv,
                            tiles_done, tile_base, out_counts, out_stats, diff2, tmb, mvc);
    SG_CHECK_LAUNCH();
    if (!tiles_done) {
        hipLaunchKernelGGL(k_compact_scan, dim3(n_frames), dim3(64), 0, st, frame_off, tile_cnt, tile_base, out_counts, out_stats, diff2, max_tiles,
                           (const int32_t *)tile_mv, tmb, mvc);
        SG_CHECK_LAUNCH();
    }
    SgPack pk{};
    if (pack) { pk.meta = pack->meta; pk.inten = pack->inten; pk.mv = pack->mv; pk.tile_mv_base = pack->tile_mv_base; pk.mv_counts = pack->mv_counts; }
    if (dtype == 0) {
        if (pack) hipLaunchKernelGGL((k_compact_scatter<float, true>), grid, dim3(SG_BLOCK), 0, st, (const float *)rows, (const float *)srows, frame_unsorted, rec, rec_q, keep, perm, frame_off, tile_base, (float *)out_rows, out_src, out_stats, max_tiles, pk);
        else hipLaunchKernelGGL((k_compact_scatter<float, false>), grid, dim3(SG_BLOCK), 0, st, (const float *)rows, (const float *)srows, frame_unsorted, rec, rec_q, keep, perm, frame_off, tile_base, (float *)out_rows, out_src, out_stats, max_tiles, pk);
    } else {
        if (pack) hipLaunchKernelGGL((k_compact_scatter<double, true>), grid, dim3(SG_BLOCK), 0, st, (const double *)rows, (const double *)srows, frame_unsorted, rec, rec_q, keep, perm, frame_off, tile_base, (double *)out_rows, out_src, out_stats, max_tiles, pk);
        else hipLaunchKernelGGL((k_compact_scatter<double, false>), grid, dim3(SG_BLOCK), 0, st, (const double *)rows, (const double *)srows, frame_unsorted, rec, rec_q, keep, perm, frame_off, tile_base, (double *)out_rows, out_src, out_stats, max_tiles, pk);
    }
    SG_CHECK_LAUNCH();
    return 0;
}

// pre-augment crop, stage 1: flags + per-frame counts (out_counts[f] = rows of frame f inside the camera's view)
extern "C" int sg_launch_crop_count(const void *rows, int dtype, const int64_t *frame_off, int n_frames, uint8_t *keep, int32_t *tile_cnt,
                                    int32_t *tile_base, int64_t *out_counts, int64_t *stats_scratch, const SgFov *fov, int64_t max_tiles,
                                    void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_crop_flag<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)rows, frame_off, keep, tile_cnt, max_tiles, *fov);
    else hipLaunchKernelGGL(k_crop_flag<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)rows, frame_off, keep, tile_cnt, max_tiles, *fov);
    SG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_compact_scan, dim3(n_frames), dim3(64), 0, st, frame_off, tile_cnt, tile_base, out_counts, stats_scratch, (const unsigned long long *)nullptr, max_tiles,
                       (const int32_t *)nullptr, (int32_t *)nullptr, (int64_t *)nullptr);
    SG_CHECK_LAUNCH();
    return 0;
}

// stage 2: rows of frame f to new_off[f] .. (stable), crop_src = their rows in the original frame
extern "C" int sg_launch_crop_scatter(const void *rows, int dtype, const uint8_t *keep, const int64_t *frame_off, const int64_t *new_off,
                                      int n_frames, const int32_t *tile_base, void *out_rows, int32_t *crop_src, int64_t max_tiles, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_crop_scatter<float>, grid, dim3(SG_BLOCK), 0, st, (const float *)rows, keep, frame_off, new_off, tile_base, (float *)out_rows, crop_src, max_tiles);
    else hipLaunchKernelGGL(k_crop_scatter<double>, grid, dim3(SG_BLOCK), 0, st, (const double *)rows, keep, frame_off, new_off, tile_base, (double *)out_rows, crop_src, max_tiles);
    SG_CHECK_LAUNCH();
    return 0;
}
