// sg_common.h -- types shared by the host API (snowgpu_api.cpp) and the gfx950 kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define SG_PI 3.141592653589793 /* np.pi (simulation.py:26) */
#define SG_TWO_PI (2 * SG_PI)
#define SG_RBINS 1230           /* M_extended (simulation.py:113) */
#define SG_NBINS 2048           /* azimuth bins per table: 3.07 mrad, one beam width at 3 mrad */
#define SG_BIN_MARGIN 1e-6      /* rad; flakes are filed under every bin their angular interval +- margin touches */
#define SG_BEAM_MARGIN 1e-9     /* rad; a beam scans every bin its wedge +- margin touches */
#define SG_TBL_STRIDE 16     /* 64-bit words between the per-table counters of the segment builder: one 128-byte line each -- the 16 384 atomics of a
                               256-sweep batch fall on 64 tables, and sixteen counters to a line queued them behind one another (k_seg_count 30 us) */
#define SG_BLKREC 8          /* int32 words per block record of the segment order (five used) */
#define SG_HITS_UNDECIDED 0x40000000   /* in a beam's flake count: a distance test of the pass over all rows was too close to call (sg_beam.h: sg_near_ray) */
#define SG_MAX_LASERS 256
#define SG_LCAP 63              /* hard cap on flakes intersecting one beam (slow path capacity) */
#define SG_TILE 1024            /* rows per sort / compaction tile */

// One filed flake, 64 bytes = one half cache line, beam-independent quantities hoisted out of
// get_occlusions' beam loop (simulation.py:351-354, :405; geometry.py:138-190, :32-80).
struct SgEntry {      // 64 bytes; the first half decides whether the flake meets a beam, the second is read of those that do (sg_beam.h: sg_flake_test)
    double phi;      // atan2(y, x) in [0, 2 pi]                simulation.py:351-352
    double x, y, r;  // table row                               simulation.py:329-330
    double t0, t1;   // tangent angles (right, left)            geometry.py:32-80
    double rho;      // sqrt(x^2 + y^2)                         simulation.py:332
    uint32_t flags;  // bit 0: this bin is the first bin of the flake's (circular) bin range
    uint32_t src;    // row in the uploaded table
};

#define SG_OV_CAP 16      /* flakes per overflow slot: a beam of the pass over all rows that meets more flakes than its LDS list holds, up to
                             this many, leaves ALL of them in the slot of its sorted position and its tier runs no second scan */
#define SG_OV_STRIDE (2 + 3 * SG_OV_CAP)   /* doubles per overflow slot: range, azimuth, then (a1, a2, rho) per flake -- the plane order of the
                                              hand-over queues, stride 1 */
#ifndef SG_QSTEPS
#define SG_QSTEPS 16     /* coarse range index of a bin: counts below 0, 8, .. 120 m (at most 64: k_table_index is one wave per bin) */
#define SG_QSTEP_M 8.0
#endif

struct SgTable {
    const SgEntry *entries;     // bins concatenated, each bin sorted by rho ascending
    const uint32_t *bin_start;  // n_bins + 1 offsets into entries
    const uint32_t *bin_q;      // n_bins x SG_QSTEPS: records of the bin nearer than SG_QSTEP_M * k metres (coarse range index)
    uint32_t n_bins;
    uint32_t n_entries;
    double inv_bin_w;           // n_bins / (2 pi)
    uint32_t n_flakes;
    uint32_t max_bin;           // longest bin (entries)
};

struct SgLasers {
    double focal_slope[SG_MAX_LASERS];   // simulation.py:75
    double focal_offset[SG_MAX_LASERS];  // simulation.py:76
    int32_t min_i[SG_MAX_LASERS];        // simulation.py:72
    int32_t max_i[SG_MAX_LASERS];        // simulation.py:123-126
    int32_t n;
};

// Result record of one beam, one uint32 per channel-sorted position (the per-beam kernels write 4 bytes per row; the
// compaction kernels rebuild the output row from the ORIGINAL row and this record -- no un-compacted copy of the cloud):
//   bits 0..7   new intensity (simulation.py:188, an integer in [min_i, max_i] <= 255), labels 1 / 2 only
//   bits 8..9   label: 0 unchanged, 1 attenuated, 2 scattered (simulation.py:160, :174)
//   bit  10     copy-through row: its channel has no laser (Q5), column 4 keeps the channel value
//   bit  11     label 0: bits 0..7 are the row's original intensity (SG_REC_HAS_I)
//   bits 12..22 argmax bin of the power profile (simulation.py:151), label 2: the point moves to k / 10 - c tau / 2
//   bit  31     reference: the record is rec_q[bits 0..30] (written by k_power, densely, in queue order)
#define SG_REC_LABEL_SHIFT 8
#define SG_REC_COPY (1u << 10)
#define SG_REC_HAS_I (1u << 11) /* label 0 only: bits 0..7 hold the row's ORIGINAL intensity (an integer in [0, 255]), so the
                                   noise-floor pass need not read the row; set by the pass over all rows for beams it finishes itself */
#define SG_REC_SLOT (1u << 31)  /* the record proper is rec_q[low 31 bits]: the row's slot in the hand-over queue */
#define SG_REC_K_SHIFT 12
#define SG_MAX_CLASSES 4        /* later capacity tiers incl. the global-list tier */

#if defined(__HIPCC__) || defined(__HIP__)
typedef int2 int2_t;
#else
struct int2_t { int32_t x, y; };
#endif

struct SgBeamArgs {
    const void *rows;            // AoS rows, all frames
    const int64_t *frame_off;    // n_frames + 1
    int32_t n_frames;
    int64_t n_total;
    int64_t uniform_rows;        // > 0: every frame has this many rows (frame = position / uniform_rows)
    float inv_uniform_rows;
    const int32_t *perm;         // channel-sorted position (global) -> frame-local source row (written for UNSORTED frames only)
    // Rows in channel-sorted order without a gather: a frame whose input rows already are channel-sorted (frame_unsorted[f] == 0:
    // channel-major sweeps) is read in place -- sorted position g IS row g of `rows` --, any other frame (firing order: an STF .bin
    // interleaves the channels, precompute.py:78) from the sorted copy the channel sort's scatter pass makes (row g of `srows`).
    const void *srows;           // n_total rows, valid where frame_unsorted[f] != 0
    const int32_t *frame_unsorted;   // n_frames
    const SgTable *frame_tables; // n_frames x n_lasers resolved descriptors (entries == nullptr: unknown table id)
    const SgLasers *las;
    const double *rgrid;         // SG_RBINS
    double beam_div_deg;
    uint32_t *rec;               // per sorted position: result record (SG_REC_*)
    uint32_t *rec_q;             // per queue slot of the direct-mode pass: result record of the beam queued there
    void *rng;                   // per sorted position: the beam's range in the row dtype (simulation.py:89), written by the pass over
                                 // all rows for every simulated beam: the noise-floor pass reads 4 bytes instead of gathering the row
    int32_t *status;             // [0] error code, [1] first offending sorted row
    int32_t *tier_hint;          // page-locked host words the device can write (or null): beams per later tier of THIS batch, read by the host when it schedules a later one
    unsigned long long *diff2;   // per frame: sum over attenuated rows of 2 * (0.9 * max_intensity - new_i)
    int32_t exact_math;          // 1: libm sin / tan + true division (validation mode)
    int32_t per_lane_scan;       // candidate scan: >= 0: the wave flattens it in the pass over all rows, one beam per lane in the tiers;
                                 // -1 = flattened everywhere (SNOWGPU_PER_LANE_SCAN)
    // tier classes: a beam of the first pass that met more flakes than its list holds is flagged with the first later
    // tier whose capacity takes all of them (the scan keeps COUNTING after the list is full, so the count is exact)
    int32_t n_cls;               // later tiers (the last one is the global-list tier, capacity = table size)
    int32_t cls_cap[SG_MAX_CLASSES];
    // direct mode: blocks walk (table, frame, channel) segments of the sorted rows (seg_blk != null) or plain chunks of
    // q_chunk sorted positions.  Either way a block lies in ONE region, and the region is also its slice of the dict queue.
    const int32_t *seg_blk;      // n_seg: first block of segment i
    const int64_t *seg_start;    // n_seg: global sorted position of the segment's first row
    const int32_t *seg_cnt;      // n_seg: rows
    const int32_t *seg_frame;    // n_seg: frame | channel << 22
    const int32_t *seg_n;        // [0] = n_seg, [1] = blocks in all segments
    const int32_t *seg_of_blk;   // grid_blocks x SG_BLKREC: per block of the segment order {segment, its first sorted position, its rows, frame | channel << 22, its first block}
    int64_t grid_blocks;         // host: blocks of this launch (upper bound; surplus blocks leave at once)
    int32_t q_chunk;             // linear mode: sorted positions per region (a multiple of the block size)
    // The pass runs as a few launches over consecutive block ranges, so that k_power of one range runs next to the scan
    // of the next: blocks [blk_lo, blk_hi) (linear order), or chunk `chunk` of chunk_blk (segment order: device-built
    // boundaries, cut at segment starts).
    int64_t blk_lo, blk_hi;
    const int32_t *chunk_blk;
    int32_t chunk;
    // Hand-over queue of the direct-mode pass: a beam that met flakes hands its range, azimuth and flake list to k_power,
    // which builds the occlusion dict and everything after it.  Blocked SoA (groups of 64 slots, snowgpu_kernels.hip).
    // A region's beams with ONE flake fill its slice from the front, the others from the back (waves of uniform loop
    // length in k_power); qn[region] = front count | back count << 32, bumped once per wave.
    double *dq;
    int32_t *dq_g;               // per slot: sorted position of the beam
    uint16_t *dq_sc;             // per slot: flakes in the list | channel << 8
    unsigned long long *qn;      // per region
    int2_t *pw_items;            // work items of k_power (k_power_plan): {first slot, count | (frame + 1) << 10}
    int32_t *pw_count;           // [0] items planned, [1] items of k_power_few, [2] beams on back_list, [3] k_power_all's item ticket (reset per chunk)
    // One work list for the received-power phase (k_power_all): the multi-flake beams of ALL regions closed up -- back_list[i] = queue slot,
    // region r's back run at bbase[r] (k_power_plan) copied by k_tier_gather -- so that a wave's 64 lanes are 64 beams whatever region they
    // came from (per region the back run is ~80 slots at C2: a round of 64 and a round of 16).  null: the regions' runs are the items.
    int32_t *back_list;
    int32_t *bbase;
    int2_t *pw_items1;           // work items of k_power_few (the front of every region's slice), or null: k_power takes them too
    int32_t front_max;           // beams with up to this many flakes fill a region's slice from the front (1 .. 3 with pw_items1; else 1)
    int64_t n_regions_ub;        // host: upper bound of the regions (segments / linear chunks)
    int32_t blk_rows;            // rows per block of the direct-mode pass
    int32_t kp_lds_quarters;     // host: share of a CU's capacity k_power takes for the main queue (1..4 quarters; 0 = all)
    int64_t dq_n;                // plane stride (= n_total)
    // list mode: this launch handles entries [work_lo, min(work_hi, count)) of class `cls` of the tier lists
    // The pass over all rows puts an over-full beam on the list of the tier that holds all its flakes: into its region's slice of
    // tier_sparse (class k at k * tier_stride; tn[region][k] entries from the region's first sorted position on), which k_power_plan
    // (bases) and k_tier_gather close up into tier_list.  Any order: a tier's results do not depend on it.
    int32_t *tier_list;          // the class lists, tier_stride (= n_total) entries apart
    int32_t *tier_info;          // [0..3] entries per class
    int64_t tier_stride;
    int32_t *tier_sparse;        // the same before closing up
    int32_t *tn, *tbase;         // per region x SG_MAX_CLASSES: entries, start in the closed-up list
    int32_t cls;
    int32_t work_lo, work_hi;
    int32_t work_hint;           // host: beams this class is expected to hold (from the batches before; 0: unknown) -- sizes the grids of its kernels
    // row kernels (snowgpu_rows.hip): beams whose dict needs NumPy's pairwise sum (an owner with >= 8 slots) are deferred to a
    // second instantiation through this list (laid out like tier_list: class k at k * tier_stride) and its per-class counters
    int32_t *redo_list;
    int32_t *redo_cnt;
    // dict hand-over of a list-mode pass: entry i of the class -> slot i (planes of tq_cap entries)
    double *tq;
    uint16_t *tq_sc;
    int32_t tq_cap;
    int32_t tq_unsorted;
    // Overflow slots of the pass over all rows, one per sorted position (SG_OV_STRIDE doubles, touched only by beams that over-fill
    // their LDS list): range, azimuth, (a1, a2, rho) of every flake met, in arrival order; ov_sc[g] = flakes | channel << 8.
    // k_power<.., LISTQ> of a class reads them instead of a hand-over buffer when ov_list is set (and sorts by range as it loads).
    double *ov;
    uint16_t *ov_sc;
    int32_t ov_cap;              // 0: no overflow slots (every over-full beam is scanned again by its tier)
    int32_t ov_list;             // k_power<.., LISTQ>: this class's flake lists are the overflow slots of its rows         // the slots hold the flakes in scan order (k_tier_scan_direct): k_power sorts them by range as it loads them
    // global-list tier: per-lane lists in global memory, h_cap entries each, h_lanes lanes
    double *h_lists;
    int32_t h_cap, h_lanes;
    // optional occlusion-dict tap (snowgpu_debug_occlusions)
    int32_t *dbg_count;
    double *dbg_rj;
    double *dbg_ratio;
    int32_t dbg_cap;
};

// Camera-FOV crop of augment() (simulation.py:39-47, :532-540) as the compaction applies it: M = V2C^T R0^T (4 x 3,
// lidar_to_rect) and P2 (3 x 4) in float64, image h x w.  enabled = 0: no crop.
struct SgFov {
    int32_t enabled;
    int32_t reserved_;           // (the pre-augment crop of precompute.py:96-99 is a context switch: snowgpu_set_fov_precrop)
    double m[12];                // lidar_to_rect: rect = [x y z 1] . m   (row-major 4 x 3)
    double p[12];                // P2 (row-major 3 x 4)
    double img_h, img_w;
};

// Launch wrappers implemented in snowgpu_kernels.hip (hipStream_t passed as void*).
#ifdef __cplusplus
extern "C" {
#endif
// ONE persistent kernel for everything k_power_few left: the 16-entry class (overflow slots), the 8-entry class (overflow slots) and the
// multi-flake beams of the main queue (a->back_list), in that order -- longest lists first -- from one item space; waves_per_cu persistent
// one-wave blocks per CU; ticket: items by an atomic cursor (a->pw_count[3]) instead of striding.  cls8 / cls16: the classes' indices in
// the tier lists, or -1.
int sg_launch_power_all(const SgBeamArgs *a, int dtype, int cls8, int cls16, int waves_per_cu, int ticket, void *stream);
// compact input: (x, y, z, intensity) float32 rows + channel bytes -> (x, y, z, intensity, channel) float32 rows
int sg_launch_expand_rows(const void *xyzi, const uint8_t *ch, void *rows, int64_t n, void *stream);
int sg_launch_sort(const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t n_total,
                   int32_t *tile_hist, int32_t *tile_base, uint16_t *rank, uint8_t *ch8, int32_t *perm, int32_t *status,
                   int64_t max_tiles_per_frame, const double *lean_plane, double *lean_part, int32_t *tile_unsorted,
                   int32_t *frame_unsorted, void *srows, int identity_perm /* 1: perm also for sorted frames (debug tap) */,
                   int phase /* 1: histogram + per-frame scan; 2: second pass (sorted copy + perm of unsorted frames); 3: both */, void *stream);
// a caller-supplied permutation: the sorted copy by a plain gather, every frame flagged unsorted
int sg_launch_gather_rows(const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t n_total, int64_t max_frame,
                          const int32_t *perm, void *srows, int32_t *frame_unsorted, int32_t *status, void *stream);
// direct == 1: the pass over all rows (dict hand-over to sg_launch_power); else list mode over class a->cls,
// dict_only == 1: hand the dicts to sg_launch_power_list, 0: received power in place
int sg_launch_beams(const SgBeamArgs *args, int dtype, int lmax, int direct, int dict_only, void *stream);
int sg_launch_power(const SgBeamArgs *args, int dtype, int lmax, void *stream, int plan_only /* 1: k_power_plan alone; 0: the kernels it feeds */,
                    void *ev_few /* hipEvent_t recorded behind k_power_few, or null */, int which /* 1: k_power_few alone, 2: k_power alone, 3: both */);
int sg_launch_tier_gather(const SgBeamArgs *args, void *stream);
int sg_launch_power_list(const SgBeamArgs *args, int dtype, int lmax, void *stream);
int sg_launch_huge(const SgBeamArgs *args, int dtype, void *stream);
// the scan of a later tier without LDS lists: hits go to the tier's hand-over buffer in scan order (args->tq_unsorted must be 1 for its k_power)
int sg_launch_tier_scan(const SgBeamArgs *args, int dtype, int lmax, void *stream);
// class args->cls of the tier lists (capacity lmax) as a row kernel: G lanes per beam, scan + dict + received power in one pass
int sg_launch_rows(const SgBeamArgs *args, int dtype, int lmax, void *stream);
int sg_launch_segments(const int64_t *frame_off, int n_frames, const int32_t *tile_base, int64_t max_tiles, const int32_t *table_ids,
                       int n_las, int n_tables, int block, unsigned long long *tbl_cnt, unsigned long long *tbl_base, int32_t *seg_blk,
                       int64_t *seg_start, int32_t *seg_cnt, int32_t *seg_frame, int32_t *seg_n, int32_t *seg_of_blk,
                       int32_t *chunk_blk, const SgTable *tables, SgTable *resolved /* n_frames x n_las, or null */, void *stream);
int sg_launch_segments_small(const int64_t *frame_off, int n_frames, const int32_t *tile_base, int64_t max_tiles, const int32_t *table_ids,
                             int n_las, int n_tables, int block, int32_t *seg_blk, int64_t *seg_start, int32_t *seg_cnt, int32_t *seg_frame,
                             int32_t *seg_n, int32_t *seg_of_blk, int32_t *chunk_blk, const SgTable *tables, SgTable *resolved,
                             unsigned long long *zero, int64_t n_zero, void *stream);
int sg_beams_block(int lmax);
int sg_launch_resolve_tables(const SgTable *tables, int n_tables, const int32_t *table_ids, int64_t n, SgTable *out, void *stream);
// Packed result transfer (snowgpu_set_result_transfer): what the compaction writes instead of 5-column rows -- per kept row of frame f, at
// frame_off[f] + j: meta = source row (30 bits) | code << 30 (0 / 1 / 2 = label, 3 = label 0 whose column 4 keeps the input's channel value)
// and the output intensity (row dtype); the moved coordinates of the batch's label-2 rows form one list in output order (frame after
// frame: frame f's start = the sum of mv_counts of the frames before it), three values each, at mv.
struct SgPackOut {
    uint32_t *meta;
    void *inten;
    void *mv;
    int64_t *mv_counts;          // n_frames: kept label-2 rows
    int32_t *tile_mv, *tile_mv_base;   // scratch: n_frames x max_tiles
};
int sg_launch_compact(const void *rows, const void *srows, const int32_t *frame_unsorted, int dtype, const uint32_t *rec, const uint32_t *rec_q, const void *rng, const double *thr_poly, uint8_t *keep, const int32_t *perm,
                      const int64_t *frame_off, int n_frames, int64_t n_total, int32_t *tile_cnt,
                      int32_t *tile_base, void *out_rows, int32_t *out_src, int64_t *out_counts,
                      int64_t *out_stats, const unsigned long long *diff2, const SgFov *fov, int64_t max_tiles_per_frame,
                      const SgPackOut *pack /* or null: rows + out_src */, unsigned long long *tiles_done /* n_frames words, zero */, void *stream);
int sg_launch_crop_count(const void *rows, int dtype, const int64_t *frame_off, int n_frames, uint8_t *keep, int32_t *tile_cnt,
                         int32_t *tile_base, int64_t *out_counts, int64_t *stats_scratch, const SgFov *fov, int64_t max_tiles, void *stream);
int sg_launch_crop_scatter(const void *rows, int dtype, const uint8_t *keep, const int64_t *frame_off, const int64_t *new_off,
                           int n_frames, const int32_t *tile_base, void *out_rows, int32_t *crop_src, int64_t max_tiles, void *stream);
#ifdef __cplusplus
}
#endif
