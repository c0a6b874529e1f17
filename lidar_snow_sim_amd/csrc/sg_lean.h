// sg_lean.h -- the per-tile statistics of the lean snowfall prepass (snowgpu_prepass.hip), shared with the channel sort's first
// kernel (snowgpu_kernels.hip: k_sort_hist<T, true>), which streams the same rows and can take them on the way.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "sg_common.h"

// q: sums over the ground rows with a1 = range, a2 = range^2 (a float32 product for float32 rows, as np.polyfit has it), c = cos(angle)
enum { LQ_A2A2 = 0, LQ_A2A1, LQ_A2, LQ_A1A1, LQ_A1, LQ_A2GC, LQ_A2C, LQ_A1GC, LQ_A1C, LQ_GC, LQ_C };
#define LP_COLS 20         /* doubles per tile of the lean path's partials */
enum { LP_N = 0, LP_SX, LP_SY, LP_YMAX, LP_MXX, LP_MXY, LP_PREFIX, LP_Q0 = 8 /* .. LP_Q0 + 10 */ };

struct SgLeanTile {
    const double *plane;    // n_frames x 4 (wx, wy, wz, h)
    double delta;           // ground band half width (simulation.py:450-451: 0.5)
    double *part;           // [frame][tile][LP_COLS]
    int64_t max_tiles;
};

// v + (v of the lane N places up in the same row of 16 lanes), N = 8, 4, 2, 1: a DPP move per half of the double instead of a
// cross-lane read through the LDS pipeline.  What the lanes whose partner lies outside the row get is not used by the callers.
template <int N>
__device__ __forceinline__ double lean_row_shl(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x100 | N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x100 | N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// wave reduction, the sum in lane 0: the tree of `v += __shfl_down(v, o)`, o = 32 .. 1, bit for bit (lane 0 adds the same pairs in
// the same order); the four steps inside a row of 16 lanes as DPP moves
__device__ __forceinline__ double lean_wave_sum(double v)
{
    v += __shfl_down(v, 32);
    v += __shfl_down(v, 16);
    v += lean_row_shl<8>(v);
    v += lean_row_shl<4>(v);
    v += lean_row_shl<2>(v);
    v += lean_row_shl<1>(v);
    return v;
}

// block reduction of K values in a fixed order: lanes -> waves -> thread 0 (256 threads)
template <int K>
__device__ __forceinline__ void lean_block_sum(double (&v)[K], double *smem /* 4 * K */)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = lean_wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) for (int k = 0; k < K; ++k) smem[w * K + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < K; ++k) v[k] = ((smem[k] + smem[K + k]) + smem[2 * K + k]) + smem[3 * K + k];
}

// ground test, range, I / cos and cos of one row against the plane (w, h): simulation.py:450-455, augmentation.py:207-208
template <typename T>
__device__ __forceinline__ bool lean_row(double delta, double w0, double w1, double w2, double h, double wn, T x, T y, T z, T inten,
                                         double &gd, double &gn, double &gc)
{
    const double dot = ((double)x * w0 + (double)y * w1) + (double)z * w2;   // np.matmul(pc[:, :3], w)
    const double hog = dot + h;
    if (!(hog < delta && hog > -delta)) return false;                        // simulation.py:450-451
    double nrm;
    if (sizeof(T) == 4) nrm = (double)sqrtf((float)((x * x + y * y) + z * z));   // float32 norm (simulation.py:455)
    else { const double xd = (double)x, yd = (double)y, zd = (double)z; nrm = sqrt((xd * xd + yd * yd) + zd * zd); }
    const double c = dot / (nrm * wn);                                       // simulation.py:454-455; cos(arccos(c)) taken as c
    gc = fabs(c) <= 1.0 ? c : NAN;                                           // arccos outside [-1, 1] is NaN in the reference too
    gn = (double)inten / gc;                                                 // augmentation.py:207
    gd = nrm;                                                                // augmentation.py:208
    return true;
}

// The statistics of one 1024-row tile from the four rows every thread of a 256-thread block holds (any assignment of rows to
// threads: the sums do not depend on it beyond rounding, and the reductions run in a fixed order either way); thread 0 writes
// the tile's LP_COLS partials.  sm: 4 * 13 + 6 doubles of LDS.
template <typename T>
__device__ __forceinline__ void lean_tile_stats(const SgLeanTile &a, int f, int64_t tile, const T (&rx)[4], const T (&ry)[4], const T (&rz)[4],
                                                const T (&ri)[4], const bool (&valid)[4], double *sm)
{
    const double *pl = a.plane + 4 * f;
    const double w0 = pl[0], w1 = pl[1], w2 = pl[2], h = pl[3];
    const double wn = sqrt((w0 * w0 + w1 * w1) + w2 * w2);              // np.linalg.norm(w)
    bool g[4];
    double gd[4], gn[4], gc[4];
    double v[3] = {0.0, 0.0, 0.0};
    double ymax = -INFINITY;
    for (int q = 0; q < 4; ++q) {
        g[q] = valid[q] && lean_row<T>(a.delta, w0, w1, w2, h, wn, rx[q], ry[q], rz[q], ri[q], gd[q], gn[q], gc[q]);
        if (g[q]) { v[0] += 1.0; v[1] += gd[q]; v[2] += gn[q]; ymax = fmax(ymax, gn[q]); }
    }
    double *smax = sm + 52, *s_mean = sm + 56;
    lean_block_sum<3>(v, sm);
    ymax = fmax(ymax, __shfl_down(ymax, 32)); ymax = fmax(ymax, __shfl_down(ymax, 16));
    ymax = fmax(ymax, lean_row_shl<8>(ymax)); ymax = fmax(ymax, lean_row_shl<4>(ymax)); ymax = fmax(ymax, lean_row_shl<2>(ymax)); ymax = fmax(ymax, lean_row_shl<1>(ymax));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = ymax;
    if (threadIdx.x == 0) { s_mean[0] = v[0] > 0 ? v[1] / v[0] : 0.0; s_mean[1] = v[0] > 0 ? v[2] / v[0] : 0.0; }
    __syncthreads();
    const double mx = s_mean[0], my = s_mean[1];
    const double cnt = v[0], sx = v[1], sy = v[2];      // (valid in thread 0 only)
    double u[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4; ++q) {
        if (!g[q] || gn[q] != gn[q]) continue;
        const double dx = gd[q] - mx, dy = gn[q] - my;
        u[0] += dx * dx; u[1] += dx * dy;
        // np.polyfit keeps the float32 dtype of x for the Vandermonde columns: x^2 is a float32 product
        double a2;
        if constexpr (sizeof(T) == 4) { const float xf = (float)gd[q]; a2 = (double)(xf * xf); }
        else a2 = gd[q] * gd[q];
        const double a1 = gd[q], c = gc[q], dc = gd[q] * c;
        u[2 + LQ_A2A2] += a2 * a2; u[2 + LQ_A2A1] += a2 * a1; u[2 + LQ_A2] += a2; u[2 + LQ_A1A1] += a1 * a1; u[2 + LQ_A1] += a1;
        u[2 + LQ_A2GC] += a2 * dc; u[2 + LQ_A2C] += a2 * c; u[2 + LQ_A1GC] += a1 * dc; u[2 + LQ_A1C] += a1 * c;
        u[2 + LQ_GC] += dc; u[2 + LQ_C] += c;
    }
    lean_block_sum<13>(u, sm);
    if (threadIdx.x == 0) {
        double *o = a.part + ((int64_t)f * a.max_tiles + tile) * LP_COLS;
        o[LP_N] = cnt; o[LP_SX] = sx; o[LP_SY] = sy;
        o[LP_YMAX] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
        o[LP_MXX] = u[0]; o[LP_MXY] = u[1];
        for (int k = 0; k < 11; ++k) o[LP_Q0 + k] = u[2 + k];
    }
}
