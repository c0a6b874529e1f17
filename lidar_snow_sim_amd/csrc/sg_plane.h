// sg_plane.h -- device ground-plane estimate (tools/wet_ground/planes.py:12-50).  Implemented in snowgpu_plane.hip.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define SG_PLANE_REFERENCE 0   /* what the reference returns today (scikit-learn >= 1.2): ([0, 0, 1], -1.55), whatever the cloud */
#define SG_PLANE_LSQ 1         /* least squares over the crop of planes.py:21-27 */
#define SG_PLANE_RANSAC 2      /* Philox-seeded RANSAC over the crop, refit on the consensus set */

struct SgPlaneScratch {
    void *buf[4];
    size_t cap[4];
};

struct SgPlaneParams {
    int method;
    int trials;          // RANSAC trials (planes.py:35 max_trials=1000)
    int min_rows;        // crop rows <= min_rows -> flat earth (planes.py:29: the reference compares with the column count, 5)
    uint64_t seed;
    double std_height;   // planes.py:12 standart_height
};

#ifdef __cplusplus
extern "C" {
#endif
// plane: n_frames x 4 doubles (wx, wy, wz, h); info: optional n_frames x 4 int32 (crop rows or -1, model 0 / 1 / 2, rows the model was
// fitted on, valid RANSAC trials).  frame_cnt: optional rows actually present per frame.  Returns 0, a hipError_t, or -1 (allocation).
int sg_plane_run(SgPlaneScratch *s, const SgPlaneParams *p, const void *rows, int dtype, const int64_t *frame_off,
                 const int64_t *frame_cnt, int n_frames, int64_t n_total, int64_t max_frame, double *plane, int32_t *info, void *stream);
void sg_plane_release(SgPlaneScratch *s);
#ifdef __cplusplus
}
#endif
