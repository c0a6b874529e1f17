/*
 * snowgpu.h -- C ABI of libsnowgpu.so, the MI355X (gfx950) snowfall-augmentation engine.
 *
 * This is the drop-in boundary for ONE hot path of SysCV/LiDAR_snow_sim: the per-beam snow
 * scattering simulation behind
 *     tools/snowfall/simulation.py::augment                 (simulation.py:427-544)
 *       -> process_single_channel                           (simulation.py:50-194)
 *       -> get_occlusions / compute_occlusion_dict          (simulation.py:298-424, :231-295)
 *       -> tools/snowfall/geometry.py                       (geometry.py:14-223)
 *       -> received_power / xsi                             (simulation.py:547-569)
 * plus the noise-threshold prepass it shares with the wet-ground model
 *     tools/wet_ground/augmentation.py::estimate_laser_parameters (augmentation.py:195-266)
 * and the wet-ground intensity model itself
 *     tools/wet_ground/augmentation.py::ground_water_augmentation (augmentation.py:25-161).
 *
 * The reference is pure Python/NumPy and has no FFI; what a maintainer binds instead of its
 * Python loops is shown in INTEGRATION.md (a ctypes stub).  All pointers are caller-owned,
 * little-endian, C-order.  No C++ exception crosses this boundary: every entry point returns
 * an int status (0 = ok) and snowgpu_last_error() returns a message for the last failure.
 *
 * Conventions shared by every entry point
 *   rows      : (x, y, z, intensity, channel) per point, float32 (dtype 0) or float64 (dtype 1)
 *               -- the STF .bin row the reference reads (precompute.py:78) and returns
 *   frames    : a batch is n_frames frames concatenated; frame f owns rows
 *               [frame_offsets[f], frame_offsets[f+1])
 *   labels    : output column 4 is the reference's label: 0 unchanged, 1 attenuated,
 *               2 scattered (simulation.py:160, :174, :192); rows whose channel is not an
 *               integer in [0, n_lasers) are never simulated and keep their channel value there
 *               (simulation.py:480-483, quirk Q5)
 *   order     : output rows are in the reference's order -- sorted by channel
 *               (simulation.py:447), STABLY here (the reference's argsort is unstable, so its
 *               within-channel order is implementation-defined); out_src returns, per output
 *               row, the index of the input row it came from (frame-local)
 */
#ifndef SNOWGPU_H
#define SNOWGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct snowgpu_ctx snowgpu_ctx;

enum {
    SNOWGPU_OK = 0,
    SNOWGPU_E_INVALID = 1,      /* bad argument (null pointer, negative size, unknown table id) */
    SNOWGPU_E_HIP = 2,          /* a HIP runtime call failed; see snowgpu_last_error */
    SNOWGPU_E_TABLE = 3,        /* particle table the reference could not process either
                                   (disk containing the origin, non-finite row) */
    SNOWGPU_E_RANGE = 4,        /* a simulated point has range >= ~120 m: the reference raises
                                   IndexError there (simulation.py:146-149, quirk Q6) */
    SNOWGPU_E_CHANNELS = 5,     /* channel column holds values other than integers in [0, 255]
                                   and no explicit permutation was supplied */
    SNOWGPU_E_OVERFLOW = 6,     /* more than SNOWGPU_MAX_FLAKES_GLOBAL flakes intersect one beam */
    SNOWGPU_E_GROUND = 7,       /* fewer than 3 ground points: the reference raises TypeError
                                   (simulation.py:462 on None, quirk Q7) */
    SNOWGPU_E_NO_DEVICE = 8     /* no HIP device / device index out of range */
};

#define SNOWGPU_MAX_FLAKES_PER_BEAM 63   /* largest per-beam list kept in LDS; beams beyond it take the global-list tier */
#define SNOWGPU_MAX_FLAKES_GLOBAL 8192  /* capacity of that tier (or the largest uploaded table, if smaller); the
                                           reference's lists are unbounded (simulation.py:413-419) */
#define SNOWGPU_MAX_LASERS 256
#define SNOWGPU_RANGE_BINS 1230  /* M_extended, simulation.py:113 */

/* ---- context ------------------------------------------------------------------------------- */

/* One context per (device, host thread).  Owns a HIP stream, the uploaded tables and all scratch. */
int snowgpu_create(int device, snowgpu_ctx **out);
void snowgpu_destroy(snowgpu_ctx *ctx);
const char *snowgpu_last_error(const snowgpu_ctx *ctx);
/* "snowgpu <version> gfx950 ..." -- also usable without a device to check the library loads. */
const char *snowgpu_version(void);

/* Replaces yaml.safe_load(calib/20171102_64E_S3.yaml) + the per-channel constants of
 * simulation.py:72-76, :123-126.  focal_offset[c] = (1 - focal_distance[c]*100/13100)**2 is
 * computed by the caller exactly as the reference does (Python float arithmetic). */
int snowgpu_set_lasers(snowgpu_ctx *ctx, int n_lasers, const double *focal_slope,
                       const double *focal_offset, const int32_t *min_intensity,
                       const int32_t *max_intensity);

/* Replaces np.load(<prefix>_<line>.npy) (simulation.py:324-330) and hoists the per-flake,
 * beam-independent work of get_occlusions (simulation.py:351-354; geometry.py:138-190, :32-80)
 * out of the frame loop: rows are (x, y, disk radius) float64.  table_id is a small non-negative
 * integer chosen by the caller; uploading again under the same id replaces the table. */
int snowgpu_upload_table(snowgpu_ctx *ctx, int table_id, const double *xyr, int64_t n_flakes);
/* snowgpu_upload_table for rows that are already in DEVICE memory (K x 3 float64): derived, binned and sorted by kernels.
 * Per-flake quantities come from the device math library (atan / atan2 / asin may differ from glibc's in the last bit);
 * for the reference's .npy tables, whose results are pinned bit for bit, use snowgpu_upload_table. */
int snowgpu_file_table_device(snowgpu_ctx *ctx, int table_id, const double *d_xyr, int64_t n_flakes);

/* Drops a table and its device memory; the id may be uploaded again.  Batches that name a dropped id fail with
 * SNOWGPU_E_INVALID. */
int snowgpu_free_table(snowgpu_ctx *ctx, int table_id);
int snowgpu_table_count(const snowgpu_ctx *ctx);

/* dart_throwing (tools/snowfall/sampling.py:90-194) on the device: same process (uniform-area centres, Exp(scale)
 * sphere diameters truncated at 20 mm, disk = slice at uniform height, no disk over the origin, no overlaps in dart
 * order, stop when the occupied area reaches occupancy_ratio * pi * r_0^2), NOT the same random stream (Philox4x32-10
 * keyed by (seed, dart index) instead of a sequential NumPy Generator) -- statistical parity, see DESIGN.md.
 *   diameter_scale_mm  = 10 / rate_parameter with rate_parameter = gunn_marshall(rate) or sekhon_srivastava(rate)
 *                        (sampling.py:108-115, :154)
 *   table_id >= 0      file the table under that id (as snowgpu_upload_table does); -1: only return the rows
 *   xyr_out / cap      optional host buffer for the K x 3 rows; *n_out = K */
/*   The rows never visit the host unless xyr_out asks for them: with table_id >= 0 the table is filed by kernels on the
 *   sampler's output (same per-flake quantities and bin order as snowgpu_upload_table, with the device math library's
 *   atan / atan2 / asin, which may differ from glibc's in the last bit). */
int snowgpu_sample_table(snowgpu_ctx *ctx, int table_id, double occupancy_ratio, double diameter_scale_mm,
                         double r_0, uint64_t seed, double *xyr_out, int64_t cap, int64_t *n_out);

/*
 * How the results of a pipelined snowgpu_augment_batch cross the link (the reference returns a fresh N' x 5 array, simulation.py:523,
 * :540-544; here the caller's out_rows / out_src receive the same bytes either way):
 *   mode 0 (default)  output rows + source indices: 24 bytes per point down the link (20 with out_src = NULL)
 *   mode 1 "packed"   per kept row its source row | label and its intensity (8 bytes; 12 for float64 rows), the moved coordinates of
 *                     scattered rows (label 2, simulation.py:176-180) apart, every copy sized by the per-frame counts; `threads` host
 *                     threads of the library (0: the CPUs this process may use minus two, at most 8; they run on the NUMA node the caller's row
 *                     buffers live on, the device's node if that cannot be told) assemble the caller's rows --
 *                     x, y, z and the channel of rows without a laser are COPIED from the caller's input rows, nothing is computed
 *                     on the host.  A third of the download; for callers bound by the link with cores to spare.
 * rows == NULL (resident rows) and single-chunk batches always use mode 0.  Mode 1 reads `rows` while it writes `out_rows`: the two must
 * not overlap (SNOWGPU_E_INVALID; mode 0 tolerates rows == out_rows), and a frame holds fewer than 2^30 rows (30-bit source rows).
 */
int snowgpu_set_result_transfer(snowgpu_ctx *ctx, int mode, int threads);
/* NUMA node HIP device `device` hangs on (sysfs, by PCI bus id), or -1 if it cannot be told: a launcher that runs one process per GPU should
 * start it on that node's CPUs (page-locked buffers touched from the other socket cost the packed transfer a quarter of its rate). */
int snowgpu_device_numa_node(int device);
/* timeline of the last packed call, ms since its start: everything enqueued, every download landed, every row assembled; threads used */
int snowgpu_debug_transfer_times(snowgpu_ctx *ctx, double *out4);

/* Rows per chunk of the host-pointer entry's upload / compute / download pipeline (default 3 * 2^19, i.e. 12 sweeps of
 * 64 x 2048; chunks alternate between SNOWGPU_PIPE_LANES = 2 compute lanes); 0 = no pipeline: one upload, one launch sequence, one download.  The
 * reference has no counterpart (its arrays never leave the host; precompute.py:78 / :106 are its I/O boundary). */
int snowgpu_set_pipeline(snowgpu_ctx *ctx, int64_t chunk_rows);

/* Validation switch for the received-power term A * sin^2(pi (R - r) / (c tau_h)) (simulation.py:549).
 * 0 (default): the engine's own sine (one reduction step + odd polynomial, < 1 ULP) and a multiplication by
 * 1 / (c tau_h); 1: the device math library's sin and a true division, operation for operation what NumPy
 * evaluates.  Both modes give the same labels / intensities (tests/test_gpu_parity.py); mode 1 is ~3x slower.
 * The switch also covers the beam-limit distance test (geometry.py:94-106, :131-135): 0 decides it as |y cos - x sin| < r and falls
 * back on the reference's tangent / root / quotient only within 1e-12 (|x| + |y|) of equality; 1 evaluates the reference's expression,
 * with the math library's tan, for every flake. */
int snowgpu_set_exact_math(snowgpu_ctx *ctx, int on);

/* on = 1: every kernel of a device-pointer batch on the caller's stream -- no side streams, no events (the environment switch SNOWGPU_SERIAL
 * as a call).  One batch alone is slower that way (4.7 against 4.0 ms per 256 sweeps: the received-power kernels and the prepass no longer
 * run side by side), but SEVERAL batches in flight on several contexts, one stream each, overlap across batches -- the memory-bound sort and
 * compaction of one beside the latency-bound per-beam kernels of another: 3.66 ms per batch with three in flight (round 6),
 * PROVIDED every stream has a hardware queue of its own: the HIP runtime serves a process' streams of one priority from GPU_MAX_HW_QUEUES
 * (default 4) queues and hands a new stream the least-used one, so two contexts' streams may share a queue and then run one after the other.
 * Either give the contexts streams of different priorities (snowgpu_lane_stream below: two lanes 3.8 ms in an unchanged environment) or set
 * GPU_MAX_HW_QUEUES=32 in the environment before the process first touches the GPU.  (With more than four queues ONE batch on its four
 * streams is slower -- 4.85 ms: hops between streams wait for queues to be switched in --, so the variable belongs to the
 * several-batches-in-flight deployment only.)  The Python tensor boundary does all this for its compute lanes (augment_batch(..., lane=k)). */
int snowgpu_set_serial(snowgpu_ctx *ctx, int on);

/* A stream of the context's for a caller that keeps several batches in flight: level 0 / 1 / 2 = the device's highest / normal / lowest stream
 * priority (made on the first call, destroyed with the context; *stream is a hipStream_t).  The HIP runtime serves each priority from a
 * queue pool of its own, so three serial contexts on streams of three DIFFERENT levels run side by side under the runtime's default of four
 * hardware queues -- streams of one level may be handed the same queue and then run one after the other (scripts/probe/queue_map_probe.hip).
 * The level only decides whose waves the dispatcher places first when two streams have work ready. */
int snowgpu_lane_stream(snowgpu_ctx *ctx, int level, void **stream);

/* Debug / parity tap: per-flake quantities of a filed table, by table row: range (simulation.py:332), azimuth in
 * [0, 2 pi] (:351-352) and the two tangent angles ordered (right, left) (geometry.py:138-190, :32-80).  out: K x 4 doubles. */
int snowgpu_debug_table(snowgpu_ctx *ctx, int table_id, double *out, int64_t cap_rows);

/* Status words (int32[8], layout under snowgpu_augment_batch_device) of the last batch that went through a host-pointer
 * entry of this context: e.g. out8[2..5] = beams each later list capacity took (summed over the chunks of a pipelined batch). */
int snowgpu_last_status(snowgpu_ctx *ctx, int32_t *out8);

/* What the status words of a DEVICE-pointer entry mean: the caller of snowgpu_augment_batch_device / snowgpu_augment_wet_batch_device
 * downloads its int32[8] `d_status` once the stream has caught up and hands the words to this function, which returns the SNOWGPU_E_*
 * code the host-pointer entries would have returned for them and leaves the message in snowgpu_last_error (0 and no message for
 * status8[0] == 0).  Same exception mapping as the host entries in the Python mirror: SNOWGPU_E_RANGE -> IndexError
 * (simulation.py:149), SNOWGPU_E_GROUND -> TypeError (simulation.py:462). */
int snowgpu_status_error(snowgpu_ctx *ctx, const int32_t *status8);

/* The 1230-entry range grid of simulation.py:106-116 as the library computes it (for tests). */
int snowgpu_range_grid(double *out /* SNOWGPU_RANGE_BINS */);

/* ---- the hot path -------------------------------------------------------------------------- */

/*
 * snowgpu_augment_batch -- augment() (simulation.py:427-544, only_camera_fov=False) for a batch of
 * frames whose rows live in HOST memory.
 *
 *   rows       the frames, concatenated -- or NULL right after snowgpu_prepass_stats on the same frames (their rows are still on the device)
 *   table_ids  n_frames x n_lasers: the table feeding channel c of frame f, i.e. the id the caller
 *              uploaded `<prefix>_<order[c]+1>.npy` under (simulation.py:70, :78, :482-486)
 *   beam_divergence_deg  as passed to augment() (degrees; simulation.py:96-97)
 *   thr_poly   n_frames x 3 quadratic (p0, p1, p2) of the per-point noise threshold over range
 *              (simulation.py:467-469), or NULL to run the prepass (simulation.py:449-467) on the
 *              device with plane (plane_w[3], plane_h) per frame: n_frames x 4 doubles (wx, wy, wz, h)
 *   plane      or NULL as well: calculate_plane (simulation.py:449, planes.py:12-50) runs on the device too, by the
 *              context's method (snowgpu_set_plane_method; default: the plane the reference returns today)
 *   noise_floor  augment()'s noise_floor (only used by the device prepass)
 *   perm       optional n_total int32 permutation (frame-local source row of each channel-sorted
 *              position); NULL = stable counting sort by channel on the device
 *   out_rows   capacity n_total rows (worst case: nothing removed); compacted per frame, frame f
 *              starts at row frame_offsets[f]
 *   out_src    n_total int32: frame-local input row of each output row; may be NULL (not downloaded: 20 instead of
 *              24 bytes per point on the way back)
 *   out_counts n_frames: rows kept per frame
 *   out_stats  n_frames x 3: num_attenuated, num_removed, avg_intensity_diff (simulation.py:525-542)
 *   out_thr_poly optional n_frames x 3: the threshold polynomial actually used
 *
 * A batch larger than ~1.5 chunks runs as a PIPELINE of chunks of whole frames (snowgpu_set_pipeline): all uploads stream
 * through one DMA queue, the chunks compute on two alternating lanes, and each chunk's results leave on a third stream while
 * the next chunks compute -- one host thread and one context keep both directions of the link and the CUs busy.
 * Page-locked rows / out_rows / out_src (snowgpu_host_alloc) make the upload asynchronous and let the download be a small
 * kernel that writes host memory directly; pageable memory works and is slower.  The call returns when everything has landed.
 */
int snowgpu_augment_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets,
                          const void *rows, int dtype, const int32_t *table_ids,
                          double beam_divergence_deg, const double *thr_poly,
                          const double *plane, double noise_floor, const int32_t *perm,
                          void *out_rows, int32_t *out_src, int64_t *out_counts,
                          int64_t *out_stats, double *out_thr_poly);

/*
 * snowgpu_augment_batch for callers bound by the link (round 6): the frames cross it as (x, y, z, intensity) float32 rows plus ONE BYTE per
 * row for the channel -- 17 bytes per point instead of the 20 of the STF row, which keeps the channel as a fifth float32
 * (precompute.py:78) -- and a kernel makes the rows on the device (k_expand_rows).  float32, integer channels 0 .. 255, no caller
 * permutation, no pre-augment crop; everything else -- table ids, plane / polynomial, results, statistics, the pipeline, the packed result
 * transfer (whose host threads then copy x, y, z from `xyzi`), the threshold callback -- as snowgpu_augment_batch.  out_rows are
 * (x, y, z, intensity, label) float32 rows: byte for byte what snowgpu_augment_batch returns for the same frames.
 */
int snowgpu_augment_batch_compact(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const float *xyzi,
                                  const uint8_t *channels, const int32_t *table_ids, double beam_divergence_deg,
                                  const double *thr_poly, const double *plane, double noise_floor, float *out_rows,
                                  int32_t *out_src, int64_t *out_counts, int64_t *out_stats, double *out_thr_poly);

/*
 * Same computation with every array already in DEVICE memory (hipMalloc'ed by the caller, e.g. a
 * torch tensor's data_ptr) and launched on the caller's stream (hipStream_t passed as void*; NULL =
 * the context's stream).  Nothing is copied to the host and the call does not synchronise: this is
 * the entry point bench.py times.  max_frame_rows = rows of the largest frame (sizes the per-frame grids;
 * 0 = unknown, n_total is used; when max_frame_rows * n_frames == n_total all frames are taken to be that size).
 * d_status (device int32[8]) receives {[0] error code, [1] first offending sorted row or -1, [2] [3] [4] [5] beams handed
 * to the 2nd / 3rd / 4th / 5th list capacity (the last one in use is the global-list tier), [6] [7] unused}; check it
 * after synchronising the stream.  The call forks work onto the context's side streams and joins
 * them back before the compaction kernels, so everything is ordered after earlier work and before later work on
 * `stream`; one batch at a time per context (its scratch buffers are reused).
 */
int snowgpu_augment_batch_device(snowgpu_ctx *ctx, int n_frames, int64_t n_total,
                                 int64_t max_frame_rows, const int64_t *d_frame_offsets, const void *d_rows, int dtype,
                                 const int32_t *d_table_ids, double beam_divergence_deg,
                                 const double *d_thr_poly, const double *d_plane, double noise_floor,
                                 const int32_t *d_perm, void *d_out_rows, int32_t *d_out_src,
                                 int64_t *d_out_counts, int64_t *d_out_stats, double *d_out_thr_poly,
                                 int32_t *d_status, void *stream);

/*
 * Debug/parity tap: the occlusion dicts of get_occlusions (simulation.py:298-424) for the rows of
 * ONE frame, in the channel-sorted order, flattened: count[i] entries for sorted row i stored at
 * [i*cap, i*cap + count[i]) of rj / ratio, near -> far, hard target last.  Host pointers.
 */
int snowgpu_debug_occlusions(snowgpu_ctx *ctx, int64_t n_rows, const void *rows, int dtype,
                             const int32_t *table_ids, double beam_divergence_deg, int cap,
                             int32_t *count, double *rj, double *ratio, int32_t *sorted_src);

/*
 * Camera-FOV crop of augment(only_camera_fov=True) (simulation.py:39-47, :532-540; get_fov_flag(lidar_to_rect(xyz),
 * (img_h, img_w))) inside the compaction kernels of every later batch of this context: v2c = Tr_velo_to_cam (3 x 4,
 * row-major), r0 = R0_rect (3 x 3), p2 = P2 (3 x 4); (1024, 1920) is the reference's image.  Cropped rows count in
 * num_removed (:538); num_attenuated / avg_intensity_diff are taken before the crop (:525-530).  enabled = 0 switches it
 * off (the matrices may then be NULL).  The reference's projection lives in an un-vendored module: textbook KITTI,
 * float64 -- parity unpinned.
 */
int snowgpu_set_fov(snowgpu_ctx *ctx, int enabled, const double *v2c, const double *r0, const double *p2, int img_h, int img_w);
/* precompute.py:96-99 crops every frame to the camera's view BEFORE augment() is called.  on = 1 (with a crop set by
 * snowgpu_set_fov): snowgpu_augment_batch compacts the uploaded frames on the device first -- num_removed etc. then
 * count against the cropped frame, as in the reference's loop, and out_src still indexes the ORIGINAL frame's rows.
 * Host-pointer entry only (the per-frame counts of the cropped batch are read back to lay the batch out). */
int snowgpu_set_fov_precrop(snowgpu_ctx *ctx, int on);

/* ---- ground plane ------------------------------------------------------------------------------ */

/*
 * calculate_plane (tools/wet_ground/planes.py:12-50; called at simulation.py:449 and wet_ground/augmentation.py:41) on the
 * device.  The reference crops the cloud to a strip in front of the car (planes.py:21-27), returns the flat-earth plane
 * ([0, 0, 1], -1.55) for a crop of no more rows than the array has columns (:29-32) and otherwise fits z = c0 x + c1 y + b with
 * scikit-learn's RANSACRegressor (:35) -- unseeded, and with scikit-learn >= 1.2 the call raises, so the except branch (:43-48)
 * returns the flat-earth plane for every cloud.  Methods of this library (parity unpinned by construction):
 *   SNOWGPU_PLANE_REFERENCE  (default) what the reference returns today: ([0, 0, 1], standard_height), no row is read
 *   SNOWGPU_PLANE_LSQ        least squares over the crop (float64, fixed summation order); w = [c0, c1, -1] / |.|, h = b (:36-41)
 *   SNOWGPU_PLANE_RANSAC     RANSAC as scikit-learn < 1.2 ran it for the reference (3-point samples, threshold = MAD of z,
 *                            most inliers, refit on them), samples from Philox4x32-10 keyed by (seed, frame, trial)
 * Both estimators return the flat-earth plane for a crop of <= min_rows rows (the reference: 5, the column count) or when no
 * model can be fitted.  The method applies to every later batch of this context that brings neither plane nor thr_poly
 * (plane == NULL), to snowgpu_wet_ground_batch / snowgpu_augment_wet_batch with a NULL plane, and to snowgpu_estimate_planes.
 */
enum { SNOWGPU_PLANE_REFERENCE = 0, SNOWGPU_PLANE_LSQ = 1, SNOWGPU_PLANE_RANSAC = 2 };
int snowgpu_set_plane_method(snowgpu_ctx *ctx, int method, uint64_t seed, int max_trials /* 0 = 1024 */, int min_rows,
                             double standard_height);
/* The planes themselves, for frames in host memory: out_planes n_frames x 4 (wx, wy, wz, h); out_info (optional) n_frames x 4
 * int32: rows in the crop (-1: not counted, reference method), model used (0 flat earth, 1 least squares, 2 ransac), rows the
 * final fit used, valid RANSAC trials. */
int snowgpu_estimate_planes(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                            double *out_planes, int32_t *out_info);
/* ... and with every array in device memory, asynchronous on the caller's stream (semantics of snowgpu_augment_batch_device). */
int snowgpu_estimate_planes_device(snowgpu_ctx *ctx, int n_frames, int64_t n_total, int64_t max_frame_rows,
                                   const int64_t *d_frame_offsets, const void *d_rows, int dtype, double *d_out_planes,
                                   int32_t *d_out_info, void *stream);

/*
 * First half of the noise-threshold prepass (simulation.py:449-461; wet_ground/augmentation.py:195-235) for a caller that wants the
 * reference's answer on ITS OWN machine: the reference takes the sparsest bin of every histogram row with np.argpartition(hist, 2)
 * [:, 0] (:236), whose result depends on the NumPy build (quirk Q8).  The device makes the expensive part -- ground rows, I / cos,
 * the regression line p, the 50 x 2555 histogram, the sums of the quadratic fit --, the caller takes the row minima with its own
 * NumPy, fits the noise line and the quadratic from the sums, and passes the polynomials to snowgpu_augment_batch (thr_poly).
 *   plane     n_frames x 4, or NULL (estimated on the device, snowgpu_set_plane_method)
 *   out_hist  n_frames x 50 x 2555 int32 (histogram2d of (range, I / cos) over (10, 70) x (5, max), augmentation.py:232-233)
 *   out_rec   n_frames x SNOWGPU_PREPASS_REC doubles: ground rows, mean range, np.mean of the float32 range column as NumPy
 *             computes it, mean I / cos, max I / cos, p slope, p intercept (linregress, :216-219), then the sums over the ground
 *             rows of a2 a2, a2 a1, a2, a1 a1, a1, a2 d c, a2 c, a1 d c, a1 c, d c, c  (a1 = d = range, a2 = range^2, c = cos of the
 *             incident angle): the normal equations of np.polyfit(range, nf (m0 d + m1) c, 2) are linear in the noise line (m0, m1)
 * Fewer than 3 ground rows in a frame: SNOWGPU_E_GROUND, as the batch entries report it.  Empty bins of out_hist already hold
 * the frame's ground-row count (hist[hist == 0] = len(pointcloud_planes), :234-235).  The uploaded rows stay in the context: the
 * next snowgpu_augment_batch of the SAME frames may pass rows = NULL and computes on them instead of uploading them again.
 */
#define SNOWGPU_PREPASS_REC 18
int snowgpu_prepass_stats(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows, int dtype,
                          const double *plane, int32_t *out_hist, double *out_rec);

/*
 * The same division of labour INSIDE one batch call (round 6): with a callback set, snowgpu_augment_batch calls that bring neither thr_poly
 * nor a caller permutation compute the device half above per group of frames (a chunk of the pipelined entry), start the per-beam kernels
 * of the group, and call
 *     fn(user, first_frame, n_frames, hist, rec, thr_poly_out)
 * on the CALLING thread as soon as the group's histograms (n_frames x 50 x 2555 int32) and records (n_frames x SNOWGPU_PREPASS_REC) have
 * landed in page-locked memory -- while those kernels, and the uploads / kernels / downloads of the other groups, run.  fn writes the
 * groups' polynomials (n_frames x 3, highest power first, np.polyfit's order) and returns 0 (anything else fails the call with
 * SNOWGPU_E_INVALID); the library uploads them and runs the group's compaction (simulation.py:516-540).  The rows cross the link once
 * and the host's np.argpartition (wet_ground/augmentation.py:236) hides behind the device's work; the two-call form above costs a second
 * crossing and runs the three stages in sequence.  fn = NULL switches back to the device's own fit.  Python: augment_batch(q8='numpy').
 */
typedef int (*snowgpu_threshold_fn)(void *user, int first_frame, int n_frames, const int32_t *hist, const double *rec, double *thr_poly_out);
int snowgpu_set_threshold_callback(snowgpu_ctx *ctx, snowgpu_threshold_fn fn, void *user);

/* ---- measurement hooks ------------------------------------------------------------------------ */

/* Record a HIP event pair around every launch of the per-beam kernel (the dominant kernel) on the stream
 * it is launched on, for up to max_launches launches.  snowgpu_profile_end synchronises that stream and
 * returns the summed kernel time in milliseconds and the number of launches timed. */
int snowgpu_profile_begin(snowgpu_ctx *ctx, int max_launches);
int snowgpu_profile_end(snowgpu_ctx *ctx, double *beam_kernel_ms, int *n_launches);

/* ---- page-locked host buffers ------------------------------------------------------------------ */

/*
 * The host-pointer entries (snowgpu_augment_batch, snowgpu_wet_ground_batch, snowgpu_augment_wet_batch) accept any
 * host memory, but only page-locked memory moves at PCIe speed and lets the copies of one context overlap the
 * kernels of another.  snowgpu_host_alloc returns such a buffer (hipHostMalloc); read the frames into it and hand
 * it in as `rows`, pass another one as `out_rows`.  The reference has no counterpart: its arrays never leave the host
 * (precompute.py:78 np.fromfile -> :106 tofile).
 */
int snowgpu_host_alloc(snowgpu_ctx *ctx, size_t bytes, void **ptr);
int snowgpu_host_free(snowgpu_ctx *ctx, void *ptr);

/* ---- wet ground ---------------------------------------------------------------------------- */

/*
 * ground_water_augmentation() (tools/wet_ground/augmentation.py:25-161, estimation_method
 * 'linear', debug off) for a batch of frames in host memory.  Output rows are float64 whatever
 * the input dtype (augmentation.py:150), ordered [non-ground rows ; kept ground rows]
 * (augmentation.py:151-152).  A frame with fewer than 1000 ground rows is returned unchanged
 * (augmentation.py:51-52) with out_flags[f] = 1.
 *   plane  n_frames x 4 (wx, wy, wz, h), or NULL: calculate_plane (augmentation.py:41) on the device (snowgpu_set_plane_method)
 */
int snowgpu_wet_ground_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets,
                             const void *rows, int dtype, const double *plane,
                             double water_height, double pavement_depth, double noise_floor,
                             double power_factor, int flat_earth, double delta, int replace,
                             double *out_rows, int32_t *out_src, int64_t *out_counts,
                             int32_t *out_flags);

/* The two lines estimate_laser_parameters fits (augmentation.py:216-219 p = linregress(range, I / cos); :248-251 the noise
 * line through the sparsest histogram bins, quirk Q8) for the NEXT snowgpu_wet_ground_batch of this context, supplied by a
 * caller that fits them itself -- with its own NumPy, whose argpartition build decides Q8 -- instead of the device's
 * first-minimum fit: n_frames x 4 doubles (p slope, p intercept, noise-line slope, noise-line intercept).  One use; NULL clears. */
int snowgpu_set_wet_lines(snowgpu_ctx *ctx, int n_frames, const double *lines);

/* estimation_method of ground_water_augmentation (reference: tools/wet_ground/augmentation.py:25, passed through by
 * pointcloud_viewer.py:2820, :2851 as self.estimation) for every later wet-ground call of this context
 * (snowgpu_wet_ground_batch, snowgpu_augment_wet_batch[_device]):
 *   method 0  'linear' -- linregress for the laser power (:215-221) and for the noise level (:247-253); the default
 *   method 1  'poly'   -- np.polyfit of degree 2 for the laser power (:223-229) and ransac_polyfit (:171-192: n = 15, k = 100,
 *                         t = 0.1, d = 15, f = 0.8) for the noise level (:243-246).  The reference draws the RANSAC samples from
 *                         NumPy's process-global UNSEEDED generator (np.random.randint, :183), so it differs from run to run;
 *                         here trial t of frame f draws from Philox4x32-10 keyed by (seed; f, t): same cloud + same seed = same
 *                         curves on every run and GPU.  Parity unpinned by construction.
 * A frame in which NO range row of the 50 x 2555 histogram has its sparsest bin above 5 returns SNOWGPU_E_GROUND under 'poly'
 * (np.polyfit raises TypeError on an empty vector); with one or two such rows the noise curve is np.polyfit's answer to the
 * under-determined system -- the minimum-norm solution of its column-scaled Vandermonde system -- as in the reference, whose RANSAC
 * cannot replace it there (a consensus set needs more than d = 15 points).  snowgpu_set_wet_lines cannot be combined with 'poly'. */
int snowgpu_set_wet_estimation(snowgpu_ctx *ctx, int method, uint64_t seed);

/* The curves the last wet-ground call of this context fitted: per frame 8 doubles -- laser power c2, c1, c0
 * (relative_output_intensity = power_factor * polyval(c, range), :221 / :228), noise level c2, c1, c0 (adaptive_noise_threshold =
 * noise_floor * polyval(c, range), :245 / :252), ground rows, RANSAC trial whose consensus refit was kept (-1: the fit over all
 * points).  'linear' frames report their two lines with c2 = 0. */
int snowgpu_wet_last_fit(snowgpu_ctx *ctx, int n_frames, double *out);

/* Parity tap of the 'poly' noise fit: the device's ransac_polyfit(x, y, order=2) (augmentation.py:171-192) on m (3 .. 50) host points
 * with the Philox draws of (seed; frame): out4 = c2, c1, c0, trial kept (-1: the fit over all points). */
int snowgpu_debug_ransac_polyfit(snowgpu_ctx *ctx, int m, const double *x, const double *y, uint64_t seed, uint64_t frame, double *out4);

/*
 * augment() followed by ground_water_augmentation() on its output, as pointcloud_viewer.py:2807-2821 chains them
 * (snow first, then wet with replace=False), as ONE launch sequence: the intermediate cloud never leaves the device.
 * Arguments are those of snowgpu_augment_batch followed by those of snowgpu_wet_ground_batch (wet_plane: n_frames x 4, or
 * NULL: estimated on the device from the snowfall stage's result, as the chained reference calls do).
 * out_stats are the snowfall statistics; out_rows (float64) / out_counts / out_flags the wet-ground result;
 * out_src maps every final row to its row in the original input frame.
 */
int snowgpu_augment_wet_batch(snowgpu_ctx *ctx, int n_frames, const int64_t *frame_offsets, const void *rows,
                              int dtype, const int32_t *table_ids, double beam_divergence_deg,
                              const double *thr_poly, const double *plane, double noise_floor,
                              const int32_t *perm, const double *wet_plane, double water_height,
                              double pavement_depth, double wet_noise_floor, double power_factor,
                              int flat_earth, double delta, int replace, double *out_rows, int32_t *out_src,
                              int64_t *out_counts, int64_t *out_stats, int32_t *out_flags);

/*
 * The same chain with every array in DEVICE memory, asynchronous on the caller's stream (semantics of
 * snowgpu_augment_batch_device: no host copy, no synchronisation, no allocation after the first call of a given size --
 * capturable into a HIP graph).  d_out_rows: n_total x 5 float64; the snowfall stage's rows stay in context scratch.
 */
int snowgpu_augment_wet_batch_device(snowgpu_ctx *ctx, int n_frames, int64_t n_total, int64_t max_frame_rows,
                                     const int64_t *d_frame_offsets, const void *d_rows, int dtype,
                                     const int32_t *d_table_ids, double beam_divergence_deg, const double *d_thr_poly,
                                     const double *d_plane, double noise_floor, const int32_t *d_perm,
                                     const double *d_wet_plane, double water_height, double pavement_depth,
                                     double wet_noise_floor, double power_factor, int flat_earth, double delta, int replace,
                                     double *d_out_rows, int32_t *d_out_src, int64_t *d_out_counts, int64_t *d_out_stats,
                                     int32_t *d_out_flags, int32_t *d_status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SNOWGPU_H */
