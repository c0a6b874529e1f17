// sg_common.h -- types shared by the host API (snowgpu_api.cpp) and the gfx950 kernels.
#pragma once
#include <stdint.h>

#define SG_PI 3.141592653589793 /* np.pi (simulation.py:26) */
#define SG_TWO_PI (2 * SG_PI)
#define SG_RBINS 1230           /* M_extended (simulation.py:113) */
#define SG_NBINS 2048           /* azimuth bins per table: 3.07 mrad, one beam width at 3 mrad */
#define SG_BIN_MARGIN 1e-6      /* rad; flakes are filed under every bin their angular interval +- margin touches */
#define SG_BEAM_MARGIN 1e-9     /* rad; a beam scans every bin its wedge +- margin touches */
#define SG_MAX_LASERS 256
#define SG_LCAP 63              /* hard cap on flakes intersecting one beam (slow path capacity) */
#define SG_TILE 1024            /* rows per sort / compaction tile */

// One filed flake, 64 bytes = one half cache line, beam-independent quantities hoisted out of
// get_occlusions' beam loop (simulation.py:351-354, :405; geometry.py:138-190, :32-80).
struct SgEntry {
    double rho;      // sqrt(x^2 + y^2)                         simulation.py:332
    double phi;      // atan2(y, x) in [0, 2 pi]                simulation.py:351-352
    double t0, t1;   // tangent angles (right, left)            geometry.py:32-80
    double x, y, r;  // table row                               simulation.py:329-330
    uint32_t flags;  // bit 0: this bin is the first bin of the flake's (circular) bin range
    uint32_t src;    // row in the uploaded table
};

struct SgTable {
    const SgEntry *entries;     // bins concatenated, each bin sorted by rho ascending
    const uint32_t *bin_start;  // n_bins + 1 offsets into entries
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

struct SgBeamArgs {
    const void *rows;            // AoS rows, all frames
    const int64_t *frame_off;    // n_frames + 1
    int32_t n_frames;
    int64_t n_total;
    int64_t uniform_rows;        // > 0: every frame has this many rows (frame = position / uniform_rows)
    float inv_uniform_rows;
    const int32_t *perm;         // channel-sorted position (global) -> frame-local source row
    const SgTable *tables;
    int32_t n_tables;
    const int32_t *table_ids;    // n_frames x n_lasers
    const SgTable *frame_tables; // n_frames x n_lasers resolved descriptors (entries == nullptr: unknown table id)
    const SgLasers *las;
    const double *rgrid;         // SG_RBINS
    double beam_div_deg;
    void *tmp_rows;              // channel-sorted, un-compacted result rows
    uint8_t *keep;               // per sorted row
    int32_t *status;             // [0] error code, [1] first offending sorted row, [2] overflow beams
    int32_t *ovf_list;           // sorted positions that overflowed this pass's list capacity
    int32_t *ovf_count;          // counter to bump for them (status[2] or status[3])
    int32_t ovf_cap;
    const int32_t *work_list;    // non-null: process these sorted positions (overflow pass)
    const int32_t *work_count;
    int32_t work_cap;            // entries of work_list that may be read
    unsigned long long *diff2;   // per frame: sum over attenuated rows of 2 * (0.9 * max_intensity - new_i)
    int32_t *dbg_count;          // optional occlusion-dict tap
    double *dbg_rj;
    double *dbg_ratio;
    int32_t dbg_cap;
    // Segment-ordered direct mode (optional, first pass): blocks walk (table, frame, channel) segments of the sorted
    // rows, so that chip-wide one or two flake tables are in use at a time and stay in L2.  null = linear order.
    const int32_t *seg_blk;      // n_seg: first block of segment i
    const int64_t *seg_start;    // n_seg: global sorted position of the segment's first row
    const int32_t *seg_cnt;      // n_seg: rows
    const int32_t *seg_frame;    // n_seg
    const int32_t *seg_n;        // [0] = n_seg, [1] = blocks in all segments
    const int32_t *seg_of_blk;   // grid_blocks: segment of block b (valid below seg_blk[n_seg])
    int64_t grid_blocks;         // host: blocks to launch in that mode (upper bound; surplus blocks leave at once)
    // First-pass split: the direct-mode pass stops at the occlusion dict and queues the beams that met a flake;
    // k_power runs the received-power phase over the queue with every lane busy.
    const int32_t *pq_list;      // sorted positions of the queued beams, ascending (built from the keep flags 16 + n_flakes)
    double *pq_dict;             // pq_stride doubles per sorted position: (range, ratio) x (n_flakes + 1), the hard target last
    int32_t *pq_count;
    int32_t pq_cap;
    int32_t pq_stride;           // 2 * (first-pass capacity + 1)
    int32_t exact_math;          // 1: libm sin + true division in the power term (validation mode)
    unsigned long long *phase_cycles;   // optional [8]: per-wave cycle totals per phase (profiling builds of the call)
};

// Launch wrappers implemented in snowgpu_kernels.hip (hipStream_t passed as void*).
#ifdef __cplusplus
extern "C" {
#endif
int sg_launch_sort(const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t n_total,
                   int32_t *tile_hist, int32_t *tile_base, uint16_t *rank, int32_t *perm, int32_t *status,
                   int64_t max_tiles_per_frame, void *stream);
int sg_launch_beams(const SgBeamArgs *args, int dtype, int lmax, void *stream);
int sg_launch_power(const SgBeamArgs *args, int dtype, int lmax, void *stream);
int sg_launch_segments(const int64_t *frame_off, int n_frames, const int32_t *tile_base, int64_t max_tiles, const int32_t *table_ids,
                       int n_las, int n_tables, int block, unsigned long long *tbl_cnt, unsigned long long *tbl_base, int32_t *seg_blk,
                       int64_t *seg_start, int32_t *seg_cnt, int32_t *seg_frame, int32_t *seg_n, int32_t *seg_of_blk, void *stream);
int sg_beams_block(int lmax);
int sg_launch_list(const uint8_t *keep, int64_t n_total, int32_t *tile_cnt, int32_t *tile_base, int32_t *list, int32_t *count, int32_t cap,
                   int lo1, int hi1, int lo2, int hi2, void *stream);
int sg_launch_resolve_tables(const SgTable *tables, int n_tables, const int32_t *table_ids, int64_t n, SgTable *out, void *stream);
int sg_launch_compact(const void *tmp_rows, int dtype, const double *thr_poly, uint8_t *keep, const int32_t *perm,
                      const int64_t *frame_off, int n_frames, int64_t n_total, int32_t *tile_cnt,
                      int32_t *tile_base, void *out_rows, int32_t *out_src, int64_t *out_counts,
                      int64_t *out_stats, const unsigned long long *diff2, int64_t max_tiles_per_frame,
                      void *stream);
#ifdef __cplusplus
}
#endif
