// sg_prepass.h -- device noise-threshold prepass (simulation.py:449-467; wet_ground/augmentation.py:195-266)
// and the wet-ground model (wet_ground/augmentation.py:25-161).  Implemented in snowgpu_prepass.hip.
#pragma once
#include <stdint.h>

struct SgPrepassScratch {
    void *buf[16];
    size_t cap[16];
};

struct SgWetParams {
    double water_height, pavement_depth, noise_floor, power_factor, delta;
    int flat_earth, replace;
    const double *lines;        // optional DEVICE array n_frames x 4: the two fitted lines supplied by the caller (snowgpu_set_wet_lines)
    int estimation;             // 0: 'linear' (augmentation.py:215-221, :247-253), 1: 'poly' (:223-229, :243-246; seeded RANSAC)
    uint64_t seed;              // 'poly': seed of the RANSAC draws
    double *fit_out;            // optional DEVICE array n_frames x 8: the fitted curves (k_pre_export_fit)
    const int32_t *src_first;   // optional DEVICE array, indexed like the rows: out_src gets src_first[row] instead of the row's index in its
                                // frame (a chained call: the rows are an earlier stage's output, src_first its source rows)
};

#define SG_PRE_REC 18      /* doubles per frame of sg_prepass_stats_run's record */

#ifdef __cplusplus
extern "C" {
#endif
int sg_prepass_stats_run(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t n_total,
                         int64_t max_frame, const double *plane, int32_t *d_hist, double *d_rec, int32_t *status, void *stream);
// Returns 0, a positive hipError_t, or -1 on allocation failure.  plane: n_frames x 4 (wx, wy, wz, h).
// tiles_done: the per-tile statistics were already left in sg_prepass_reserve_tiles()'s buffer by the channel sort (sg_launch_sort)
// srows / frame_unsorted: optional -- the channel sort's sorted copy and the per-frame flags that say where it is valid
int sg_prepass_run(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off, int n_frames,
                   int64_t n_total, int64_t max_frame, const double *plane, double noise_floor, double *thr_poly, int32_t *status,
                   void *stream, int tiles_done, const void *srows, const int32_t *frame_unsorted, int hist_cleared);
// clears the prepass' histogram on `stream` ahead of time (independent of the batch's data): sg_prepass_run(.., hist_cleared = 1) then skips its fill
int sg_prepass_clear_hist(SgPrepassScratch *s, int n_frames, void *stream);
int sg_prepass_stats_early(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t max_frame,
                           const double *plane, void *stream);
double *sg_prepass_reserve_tiles(SgPrepassScratch *s, int n_frames, int64_t max_frame);
int sg_wet_run(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off,
               const int64_t *frame_cnt, int n_frames, int64_t n_total, int64_t max_frame, const double *plane, const SgWetParams *wp, double *out_rows, int32_t *out_src,
               int64_t *out_counts, int32_t *out_flags, int32_t *status, void *stream);
// out[off[f] + i] = first[off[f] + second[off[f] + i]] for i < counts[f] (device arrays)
int sg_launch_compose_src(const int64_t *frame_off, const int64_t *counts, int n_frames, int64_t max_frame,
                          const int32_t *second, const int32_t *first, int32_t *out, void *stream);
// ransac_polyfit (augmentation.py:171-192, order 2) on m <= 50 caller-supplied points in device memory: d_out[0..2] = coefficients, [3] = trial kept
int sg_debug_ransac_quad(const double *d_x, const double *d_y, int m, uint64_t seed, uint64_t frame, double *d_out, void *stream);
void sg_prepass_release(SgPrepassScratch *s);
#ifdef __cplusplus
}
#endif
