// snowgpu_plane.hip -- ground-plane estimate on gfx950: the engine's own, deterministic counterpart of
// tools/wet_ground/planes.py::calculate_plane (planes.py:12-50; called at simulation.py:449 and wet_ground/augmentation.py:41).
//
// What the reference does: crop the cloud to a strip in front of the car (planes.py:21-27); if the crop holds no more rows
// than the array has columns, return the flat-earth plane ([0, 0, 1], -1.55) (:29-32); otherwise fit z = c0 x + c1 y + b with
// scikit-learn's RANSACRegressor (:35: unseeded, and with scikit-learn >= 1.2 the spelling loss='squared_loss' raises, so the
// except branch :43-48 returns the flat-earth plane every time) and return w = [c0, c1, -1] / |.|, h = b (:36-41).
//
// Here ("needs a deterministic plane estimator (seeded RANSAC or least squares)"):
//   method 0  the plane the reference returns today, whatever the cloud: ([0, 0, 1], -1.55) -- no row is read
//   method 1  least squares on the cropped rows: centred 2 x 2 normal equations in float64, fixed summation order
//   method 2  RANSAC in the shape scikit-learn < 1.2 ran it for the reference (3-point samples, residual threshold = median
//             absolute deviation of z, squared residuals compared with it, most inliers wins, refit on the inliers), with the
//             samples drawn from Philox4x32-10 keyed by (seed, frame, trial): the same cloud and seed give the same plane on
//             every run and every GPU.  Ties: smaller mean squared residual, then the earlier trial.
// Both fall back to the flat-earth plane where the reference's code does (crop of <= min_rows rows) or would (no valid model).
// Parity with the reference is unpinned here by construction (SURVEY 8 c: its RANSAC is unseeded); the tests hold method 1 to
// NumPy's lstsq on the same rows and method 2 to its own invariants.
//
// Two kernels per batch, on whatever stream the prepass runs on:
//   k_plane_crop  streams the rows once (x, y, z), evaluates the crop in the ROW dtype as NumPy does (float32 comparisons
//                 for float32 clouds) and compacts the kept row indices per 1024-row tile in row order (wave ballots);
//   k_plane_fit   one block per frame: dense gather of the cropped points (float64), then the estimator.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "sg_common.h"
#include "sg_philox.h"
#include "sg_plane.h"

#define PLB 256
#define PL_CHUNK 1536          /* points staged in LDS at a time by the RANSAC consensus loop (36 KB) */

struct PlaneArgs {
    const void *rows;
    const int64_t *frame_off;
    const int64_t *frame_cnt;   // optional: rows present in frame f (compacted input)
    int n_frames;
    int64_t max_tiles;
    int method;
    uint64_t seed;
    int trials;
    int min_rows;               // planes.py:29: crop rows <= this -> flat earth (the reference compares with the column count)
    double std_height;          // planes.py:12 standart_height
    int32_t *tile_cnt;          // [frame][tile]
    int32_t *tile_idx;          // n_total: kept rows of a tile, compacted at its first position, frame-local indices
    double *pts;                // 3 doubles per cropped row, dense per frame from 3 * frame_off[f]
    double *plane;              // n_frames x 4
    int32_t *info;              // n_frames x 4: crop rows, model (0 flat earth, 1 least squares, 2 ransac), inliers, valid trials
};

__device__ __forceinline__ int64_t pl_rows(const PlaneArgs &a, int f)
{
    return a.frame_cnt ? a.frame_cnt[f] : a.frame_off[f + 1] - a.frame_off[f];
}

// planes.py:21-26 in the row dtype: NumPy compares a float32 column with Python floats in float32 (NEP 50), and
// -1.86 - 0.01 * x is a float32 product and a float32 difference there
template <typename T>
__device__ __forceinline__ bool pl_in_crop(T x, T y, T z)
{
    const T lim = (T)-1.86 - (T)0.01 * x;
    return z < (T)-1.55 && z > lim && x > (T)10 && x < (T)70 && y > (T)-3 && y < (T)3;
}

template <typename T>
__global__ __launch_bounds__(PLB) void k_plane_crop(PlaneArgs a)
{
    const int f = blockIdx.y;
    const int64_t base = a.frame_off[f], n = pl_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const T *rows = (const T *)a.rows;
    __shared__ int wc[4][4];
    const int tid = threadIdx.x, wv = tid >> 6;
    const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
    bool k[4];
    int pre[4];
    T rx[4], ry[4], rz[4];
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PLB + tid;
        const T *p = rows + (base + (r < n ? r : 0)) * 5;
        rx[q] = p[0]; ry[q] = p[1]; rz[q] = p[2];
    }
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PLB + tid;
        k[q] = r < n && pl_in_crop<T>(rx[q], ry[q], rz[q]);
        const unsigned long long m = __ballot(k[q]);
        pre[q] = __popcll(m & lt);
        if ((tid & 63) == 0) wc[q][wv] = __popcll(m);
    }
    __syncthreads();
    int run = 0;
    for (int q = 0; q < 4; ++q) {
        int off = run;
        for (int ww = 0; ww < wv; ++ww) off += wc[q][ww];
        if (k[q]) a.tile_idx[base + tile0 + off + pre[q]] = (int32_t)(tile0 + q * PLB + tid);
        run += wc[q][0] + wc[q][1] + wc[q][2] + wc[q][3];
    }
    if (tid == 0) a.tile_cnt[(int64_t)f * a.max_tiles + blockIdx.x] = run;
}

// ---- block helpers: fixed-order reductions whose result every thread gets ------------------------------------------
template <int K>
__device__ __forceinline__ void pl_block_sum(double (&v)[K], double *sm /* 4 * K */)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < K; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    __syncthreads();
    if (lane == 0) for (int k = 0; k < K; ++k) sm[w * K + k] = v[k];
    __syncthreads();
    for (int k = 0; k < K; ++k) v[k] = ((sm[k] + sm[K + k]) + sm[2 * K + k]) + sm[3 * K + k];
}

__device__ __forceinline__ long long pl_block_count(long long c, long long *sm /* 4 */)
{
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// order-preserving map double -> uint64 (total order of the finite values)
__device__ __forceinline__ unsigned long long pl_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double pl_unkey(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// np.median of val(i), i < m (m >= 1): the k-th smallest by bisection over the ordered keys (64 block-wide counts, exact),
// and for even m the mean with its upper neighbour, as NumPy takes it.
template <typename F>
__device__ __forceinline__ double pl_median(int m, F val, long long *smc)
{
    const int k = (m - 1) / 2;                       // lower middle (0-based)
    unsigned long long lo = 0, hi = ~0ull;
    while (lo < hi) {                                // smallest key with count(key' <= key) >= k + 1
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        long long c = 0;
        for (int i = threadIdx.x; i < m; i += PLB) c += pl_key(val(i)) <= mid;
        c = pl_block_count(c, smc);
        if (c >= k + 1) hi = mid; else lo = mid + 1;
    }
    const double lower = pl_unkey(lo);
    if (m & 1) return lower;
    // upper middle: the same value if it occurs often enough, else the smallest value above it
    long long c = 0;
    unsigned long long nxt = ~0ull;
    for (int i = threadIdx.x; i < m; i += PLB) {
        const unsigned long long kv = pl_key(val(i));
        c += kv <= lo;
        if (kv > lo && kv < nxt) nxt = kv;
    }
    c = pl_block_count(c, smc);
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(nxt, o); nxt = v < nxt ? v : nxt; }
    __shared__ unsigned long long smn[4];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smn[threadIdx.x >> 6] = nxt;
    __syncthreads();
    nxt = smn[0];
    for (int w = 1; w < 4; ++w) nxt = smn[w] < nxt ? smn[w] : nxt;
    const double upper = c >= k + 2 ? lower : pl_unkey(nxt);
    return (lower + upper) / 2;                      // np.median: mean of the two middle values
}

// z = c0 x + c1 y + b by least squares over the points that `take` selects: centred sums in two passes, every reduction in
// a fixed order.  Returns the number of points used (< 3 or a singular system: ok = false).
template <typename F>
__device__ __forceinline__ int pl_lsq(const double *pts, int m, F take, double *smd, long long *smc, double &c0, double &c1, double &b, bool &ok)
{
    double s[3] = {0.0, 0.0, 0.0};
    long long cnt = 0;
    for (int i = threadIdx.x; i < m; i += PLB) {
        const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (take(x, y, z)) { s[0] += x; s[1] += y; s[2] += z; ++cnt; }
    }
    pl_block_sum<3>(s, smd);
    cnt = pl_block_count(cnt, smc);
    ok = false;
    c0 = c1 = b = 0.0;
    if (cnt < 3) return (int)cnt;
    const double xm = s[0] / (double)cnt, ym = s[1] / (double)cnt, zm = s[2] / (double)cnt;
    double q[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < m; i += PLB) {
        const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (take(x, y, z)) {
            const double dx = x - xm, dy = y - ym, dz = z - zm;
            q[0] += dx * dx; q[1] += dx * dy; q[2] += dy * dy; q[3] += dx * dz; q[4] += dy * dz;
        }
    }
    pl_block_sum<5>(q, smd);
    const double det = q[0] * q[2] - q[1] * q[1];
    if (!(det > 1e-14 * (q[0] * q[2])) || !(q[0] > 0.0) || !(q[2] > 0.0)) return (int)cnt;      // points on one line in (x, y)
    c0 = (q[3] * q[2] - q[4] * q[1]) / det;
    c1 = (q[4] * q[0] - q[3] * q[1]) / det;
    b = zm - (c0 * xm + c1 * ym);
    ok = isfinite(c0) && isfinite(c1) && isfinite(b);
    return (int)cnt;
}

template <typename T>
__global__ __launch_bounds__(PLB) void k_plane_fit(PlaneArgs a)
{
    const int f = blockIdx.x;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int64_t base = a.frame_off[f], n = pl_rows(a, f);
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    const int32_t *tc = a.tile_cnt + (int64_t)f * a.max_tiles;
    const T *rows = (const T *)a.rows;
    double *pts = a.pts + 3 * base;
    __shared__ int s_cnt[PLB], s_pre[PLB];
    __shared__ double smd[4 * 5];
    __shared__ long long smc[4];
    __shared__ int s_run;
    __shared__ __attribute__((aligned(16))) double s_pts[3 * PL_CHUNK];
    // ---- dense gather, in row order: tiles in chunks of 256 (exclusive scan of their counts), wave w copies tiles w, w + 4, ..
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < tiles; t0 += PLB) {
        const int64_t t = t0 + tid;
        const int c = t < tiles ? tc[t] : 0;
        s_cnt[tid] = c;
        int inc = c;                                   // inclusive scan: wave, then the four wave totals
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        __shared__ int wtot[4];
        if (lane == 63) wtot[wv] = inc;
        __syncthreads();
        int off = s_run;
        for (int w = 0; w < wv; ++w) off += wtot[w];
        s_pre[tid] = off + inc - c;
        __syncthreads();
        const int nt = (int)(tiles - t0 < PLB ? tiles - t0 : PLB);
        for (int k = wv; k < nt; k += 4) {
            const int ck = s_cnt[k], pk = s_pre[k];
            const int32_t *src = a.tile_idx + base + (t0 + k) * SG_TILE;
            for (int j = lane; j < ck; j += 64) {
                const T *p = rows + (base + src[j]) * 5;
                double *d = pts + 3 * (int64_t)(pk + j);
                d[0] = (double)p[0]; d[1] = (double)p[1]; d[2] = (double)p[2];
            }
        }
        __syncthreads();
        if (tid == 0) s_run += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    const int m = s_run;
    __threadfence_block();
    __syncthreads();
    double *out = a.plane + 4 * f;
    int32_t *info = a.info ? a.info + 4 * f : nullptr;
    auto flat_earth = [&](int inl, int valid) {          // planes.py:29-32 / :43-48
        if (tid == 0) {
            out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; out[3] = a.std_height;
            if (info) { info[0] = m; info[1] = 0; info[2] = inl; info[3] = valid; }
        }
    };
    auto emit = [&](double c0, double c1, double b, int model, int inl, int valid) {
        if (tid == 0) {
            const double nrm = sqrt((c0 * c0 + c1 * c1) + 1.0);        // np.linalg.norm([c0, c1, -1]), planes.py:41
            out[0] = c0 / nrm; out[1] = c1 / nrm; out[2] = -1.0 / nrm; out[3] = b;   // h stays un-normalised (:40, quirk Q12)
            if (info) { info[0] = m; info[1] = model; info[2] = inl; info[3] = valid; }
        }
    };
    // (a plane needs three rows: a direct C-ABI caller may set min_rows below 2, and the samples below index rows m - 1 and m - 2)
    if (m <= a.min_rows || m < 3) { flat_earth(0, 0); return; }
    double c0, c1, b;
    bool ok;
    if (a.method == SG_PLANE_LSQ) {
        const int used = pl_lsq(pts, m, [](double, double, double) { return true; }, smd, smc, c0, c1, b, ok);
        if (ok) emit(c0, c1, b, 1, used, 0); else flat_earth(used, 0);
        return;
    }
    // ---- RANSAC -----------------------------------------------------------------------------------------------------
    // residual threshold = MAD of z (scikit-learn's default: np.median(np.abs(y - np.median(y))))
    const double med = pl_median(m, [&](int i) { return pts[3 * i + 2]; }, smc);
    const double thr = pl_median(m, [&](int i) { return fabs(pts[3 * i + 2] - med); }, smc);
    const int rounds = (a.trials + PLB - 1) / PLB;
    int best_cnt = -1, best_trial = 0x7fffffff, n_valid = 0;
    double best_ss = INFINITY, bc0 = 0.0, bc1 = 0.0, bb = 0.0;
    const int n_chunks = (m + PL_CHUNK - 1) / PL_CHUNK;
    for (int r = 0; r < rounds; ++r) {
        const int trial = r * PLB + tid;
        // three distinct rows of the crop from Philox(seed; frame, trial)
        bool valid = trial < a.trials;
        double m0 = 0.0, m1 = 0.0, mb = 0.0;
        if (valid) {
            uint32_t u[4];
            philox_u32x4(a.seed, (uint64_t)f, (uint32_t)trial, 0x504C414Eu /* "PLAN" */, u);
            int i0 = (int)(((uint64_t)u[0] * (uint64_t)m) >> 32);
            int i1 = (int)(((uint64_t)u[1] * (uint64_t)(m - 1)) >> 32);
            int i2 = (int)(((uint64_t)u[2] * (uint64_t)(m - 2)) >> 32);
            if (i1 >= i0) ++i1;                                        // sampling without replacement
            const int lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
            if (i2 >= lo) ++i2;
            if (i2 >= hi) ++i2;
            const double x0 = pts[3 * i0], y0 = pts[3 * i0 + 1], z0 = pts[3 * i0 + 2];
            const double x1 = pts[3 * i1], y1 = pts[3 * i1 + 1], z1 = pts[3 * i1 + 2];
            const double x2 = pts[3 * i2], y2 = pts[3 * i2 + 1], z2 = pts[3 * i2 + 2];
            const double xm = ((x0 + x1) + x2) / 3.0, ym = ((y0 + y1) + y2) / 3.0, zm = ((z0 + z1) + z2) / 3.0;
            double sxx = 0, sxy = 0, syy = 0, sxz = 0, syz = 0;
            const double dx[3] = {x0 - xm, x1 - xm, x2 - xm}, dy[3] = {y0 - ym, y1 - ym, y2 - ym}, dz[3] = {z0 - zm, z1 - zm, z2 - zm};
            for (int k = 0; k < 3; ++k) { sxx += dx[k] * dx[k]; sxy += dx[k] * dy[k]; syy += dy[k] * dy[k]; sxz += dx[k] * dz[k]; syz += dy[k] * dz[k]; }
            const double det = sxx * syy - sxy * sxy;
            if (!(det > 1e-12 * (sxx * syy)) || !(sxx > 0.0) || !(syy > 0.0)) valid = false;       // collinear sample: no model
            else {
                m0 = (sxz * syy - syz * sxy) / det;
                m1 = (syz * sxx - sxz * sxy) / det;
                mb = zm - (m0 * xm + m1 * ym);
                valid = isfinite(m0) && isfinite(m1) && isfinite(mb);
            }
        }
        // consensus: every thread tests its model against all points, staged through LDS (all lanes read the same address)
        int cnt = 0;
        double ss = 0.0;
        for (int c = 0; c < n_chunks; ++c) {
            const int p0 = c * PL_CHUNK, pn = m - p0 < PL_CHUNK ? m - p0 : PL_CHUNK;
            if (n_chunks > 1 || r == 0) {
                __syncthreads();
                for (int i = tid; i < 3 * pn; i += PLB) s_pts[i] = pts[3 * p0 + i];
                __syncthreads();
            }
            if (valid) {
                for (int i = 0; i < pn; ++i) {
                    const double res = s_pts[3 * i + 2] - ((m0 * s_pts[3 * i] + m1 * s_pts[3 * i + 1]) + mb);
                    const double r2 = res * res;                       // loss='squared_loss' ...
                    if (r2 <= thr) { ++cnt; ss += r2; }                // ... compared with the (unsquared) MAD, as scikit-learn does
                }
            }
        }
        n_valid += valid ? 1 : 0;
        if (valid && cnt >= 3) {
            const double mean_ss = ss / (double)cnt;
            if (cnt > best_cnt || (cnt == best_cnt && (mean_ss < best_ss || (mean_ss == best_ss && trial < best_trial)))) {
                best_cnt = cnt; best_ss = mean_ss; best_trial = trial; bc0 = m0; bc1 = m1; bb = mb;
            }
        }
    }
    // block arg-best: (count desc, mean squared residual asc, trial asc)
    __shared__ int r_cnt[PLB], r_trial[PLB];
    __shared__ double r_ss[PLB], r_m[3 * PLB];
    __syncthreads();
    r_cnt[tid] = best_cnt; r_trial[tid] = best_trial; r_ss[tid] = best_ss;
    r_m[3 * tid] = bc0; r_m[3 * tid + 1] = bc1; r_m[3 * tid + 2] = bb;
    __syncthreads();
    for (int s = PLB / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const int o = tid + s;
            const bool better = r_cnt[o] > r_cnt[tid] ||
                                (r_cnt[o] == r_cnt[tid] && (r_ss[o] < r_ss[tid] || (r_ss[o] == r_ss[tid] && r_trial[o] < r_trial[tid])));
            if (better) {
                r_cnt[tid] = r_cnt[o]; r_trial[tid] = r_trial[o]; r_ss[tid] = r_ss[o];
                r_m[3 * tid] = r_m[3 * o]; r_m[3 * tid + 1] = r_m[3 * o + 1]; r_m[3 * tid + 2] = r_m[3 * o + 2];
            }
        }
        __syncthreads();
    }
    long long nv = pl_block_count((long long)n_valid, smc);
    const int win = r_cnt[0];
    const double w0 = r_m[0], w1 = r_m[1], wb = r_m[2];
    if (win < 3) { flat_earth(0, (int)nv); return; }                     // scikit-learn raises: planes.py:43-48
    // refit on the consensus set of the winning model
    const int used = pl_lsq(pts, m, [&](double x, double y, double z) { const double res = z - ((w0 * x + w1 * y) + wb); return res * res <= thr; },
                            smd, smc, c0, c1, b, ok);
    if (ok) emit(c0, c1, b, 2, used, (int)nv); else flat_earth(used, (int)nv);
}

__global__ void k_plane_const(double *plane, int32_t *info, int n_frames, double std_height)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    plane[4 * f] = 0.0; plane[4 * f + 1] = 0.0; plane[4 * f + 2] = 1.0; plane[4 * f + 3] = std_height;
    if (info) { info[4 * f] = -1; info[4 * f + 1] = 0; info[4 * f + 2] = 0; info[4 * f + 3] = 0; }
}

// ---- host side --------------------------------------------------------------------------------------------------------
enum { PB_TCNT = 0, PB_TIDX, PB_PTS, PB_N };

static int pl_ensure(SgPlaneScratch *s, int i, size_t bytes)
{
    if (bytes <= s->cap[i]) return 0;
    if (s->buf[i]) (void)hipFree(s->buf[i]);
    s->buf[i] = nullptr; s->cap[i] = 0;
    const size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&s->buf[i], want) != hipSuccess) return -1;
    s->cap[i] = want;
    return 0;
}

extern "C" void sg_plane_release(SgPlaneScratch *s)
{
    for (int i = 0; i < 4; ++i) { if (s->buf[i]) (void)hipFree(s->buf[i]); s->buf[i] = nullptr; s->cap[i] = 0; }
}

#define PL_LCHK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

extern "C" int sg_plane_run(SgPlaneScratch *s, const SgPlaneParams *p, const void *rows, int dtype, const int64_t *frame_off,
                            const int64_t *frame_cnt, int n_frames, int64_t n_total, int64_t max_frame, double *plane, int32_t *info,
                            void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n_frames <= 0) return 0;
    const double std_height = p->std_height;
    if (p->method == SG_PLANE_REFERENCE || n_total <= 0) {
        hipLaunchKernelGGL(k_plane_const, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0, st, plane, info, n_frames, std_height);
        PL_LCHK();
        return 0;
    }
    const int64_t max_tiles = (max_frame + SG_TILE - 1) / SG_TILE > 0 ? (max_frame + SG_TILE - 1) / SG_TILE : 1;
    const size_t n = (size_t)n_total, nf = (size_t)n_frames;
    if (pl_ensure(s, PB_TCNT, nf * (size_t)max_tiles * 4) || pl_ensure(s, PB_TIDX, (n + SG_TILE) * 4) || pl_ensure(s, PB_PTS, n * 3 * 8)) return -1;
    PlaneArgs a{};
    a.rows = rows; a.frame_off = frame_off; a.frame_cnt = frame_cnt; a.n_frames = n_frames; a.max_tiles = max_tiles;
    a.method = p->method; a.seed = p->seed; a.trials = p->trials > 0 ? p->trials : 1024; a.min_rows = p->min_rows; a.std_height = std_height;
    a.tile_cnt = (int32_t *)s->buf[PB_TCNT]; a.tile_idx = (int32_t *)s->buf[PB_TIDX]; a.pts = (double *)s->buf[PB_PTS];
    a.plane = plane; a.info = info;
    // tiles past the end of a short frame are never written by k_plane_crop: their counts must read 0
    if (hipMemsetAsync(a.tile_cnt, 0, nf * (size_t)max_tiles * 4, st) != hipSuccess) return (int)hipGetLastError();
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_plane_crop<float>, grid, dim3(PLB), 0, st, a);
    else hipLaunchKernelGGL(k_plane_crop<double>, grid, dim3(PLB), 0, st, a);
    PL_LCHK();
    if (dtype == 0) hipLaunchKernelGGL(k_plane_fit<float>, dim3((unsigned)n_frames), dim3(PLB), 0, st, a);
    else hipLaunchKernelGGL(k_plane_fit<double>, dim3((unsigned)n_frames), dim3(PLB), 0, st, a);
    PL_LCHK();
    return 0;
}
