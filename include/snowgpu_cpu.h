/*
 * snowgpu_cpu.h -- libsnowcpu.so: the CPU twin of libsnowgpu.so's augment path (SURVEY.md section 8 b, last row of the proposed C ABI:
 * "snowgpu_cpu_* twins running the C++ CPU restatement for the baseline"; section 8 d: "the build's C++ CPU restatement ... on all host
 * cores -- state the count").
 *
 * What it is: the SAME per-beam arithmetic as the HIP kernels -- csrc/sg_beam.h (beam geometry, candidate scan over the binned table,
 * compute_occlusion_dict, amplitudes, the exactly pruned received-power profile, the attenuate-or-scatter decision), csrc/sg_table_host.h
 * (table filing) and csrc/sg_row.h (output rows), i.e. the kernels' own device functions compiled for the host -- driven by host threads,
 * one beam at a time; plus the frame steps around it (stable channel sort, simulation.py:447; np.round, noise-floor filter and statistics,
 * :516-530).  Replaces the same reference code as snowgpu_augment_batch: tools/snowfall/simulation.py::augment (:427-544) with
 * only_camera_fov=False.
 *
 * What it is NOT: a fallback.  Nothing in lidar_snow_sim_amd/ loads this library; libsnowgpu.so has no CPU path and fails with
 * SNOWGPU_E_NO_DEVICE without a GPU.  It exists so that bench.py can time the build's own algorithm on the host cores next to the GPU
 * (cpu_twin in the bench line; the reference-shaped restatement, oracle/, stays the cpu_baseline) and so that tests can hold the two
 * builds of one source to the same bytes.
 */
#ifndef SNOWGPU_CPU_H
#define SNOWGPU_CPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* "snowcpu <version> host ..." */
const char *snowgpu_cpu_version(void);

/*
 * augment() for a batch of frames on the host.  Arguments as snowgpu_augment_batch (include/snowgpu.h) where they have the same name:
 *   rows, dtype, frame_offsets   input frames, (x, y, z, intensity, channel) float32 (0) / float64 (1)
 *   tables_xyr / tables_k        n_tables particle tables (K x 3 float64: x, y, disk radius), filed here as snowgpu_upload_table files them
 *   table_ids                    n_frames x n_lasers: index into tables_xyr of the table channel c of frame f reads (simulation.py:78)
 *   focal_slope .. max_intensity the per-laser constants of snowgpu_set_lasers
 *   thr_poly                     n_frames x 3: the noise-threshold polynomial of every frame (simulation.py:462-469) -- REQUIRED: the prepass
 *                                (plane, estimate_laser_parameters, polyfit) is not restated here; callers pass what the device prepass or the
 *                                host mirror fitted
 *   threads                      host threads (<= 0: the CPUs this process may run on)
 *   out_rows / out_src / out_counts / out_stats   as snowgpu_augment_batch: frame f's output rows lie at frame_offsets[f] .. + out_counts[f]
 *   status                       int32[2]: [0] 0 or SNOWGPU_E_RANGE (4) / SNOWGPU_E_TABLE (3) / SNOWGPU_E_OVERFLOW (6), [1] offending row or table
 * Returns 0, or the status code.  Bit for bit the rows of snowgpu_augment_batch in its default arithmetic (tests/test_gpu_parity.py).
 */
int snowgpu_cpu_augment_batch(int n_frames, const int64_t *frame_offsets, const void *rows, int dtype, int n_tables,
                              const double *const *tables_xyr, const int64_t *tables_k, const int32_t *table_ids, int n_lasers,
                              const double *focal_slope, const double *focal_offset, const int32_t *min_intensity,
                              const int32_t *max_intensity, double beam_divergence_deg, const double *thr_poly, double noise_floor,
                              int threads, void *out_rows, int32_t *out_src, int64_t *out_counts, int64_t *out_stats, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif
