// snowgpu_prepass.hip -- the frame-level prepass and the wet-ground model on gfx950.
//
//   noise-threshold prepass   simulation.py:449-467 + wet_ground/augmentation.py:195-266 ('linear')
//   wet-ground augmentation   wet_ground/augmentation.py:25-161 + wet_ground/phy_equations.py:35-108
//
// Both start from the same per-frame estimate (estimate_laser_parameters): ground rows by plane distance,
// incident angle, I / cos(angle) against range, a least-squares line, a 50 x 2555 (range x normalised
// intensity) histogram whose per-range-row sparsest occupied bin gives the noise line.  The reference does
// this with NumPy/SciPy calls on the host; here it is a chain of small kernels, one grid row per frame,
// whose floating-point reductions run in a fixed order (tile partials -> ordered final sum) so that a
// batch is reproducible run to run.  The reductions are float64; they agree with NumPy's to ~1e-12, not
// bit for bit (NumPy's own summation order depends on its build) -- see DESIGN.md "prepass tolerance".
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <cstdlib>
#include "sg_common.h"
#include "sg_prepass.h"
#include "sg_lean.h"
#include "sg_philox.h"
#include "sg_math.h"

#define PB 256
#define HX 50     /* range rows of the histogram (augmentation.py:232) */
#define HY 2555   /* normalised-intensity bins */

struct PreFrame {          // per-frame state shared by the kernels
    double n_ground;       // ground rows
    double xmean, ymean;   // mean range / mean normalised intensity
    double xmean32;        // np.mean of the float32 range column as NumPy computes it (float32 pairwise sum)
    double ymax;           // max normalised intensity (histogram range, augmentation.py:233)
    double p0, p1;         // linregress(dist, normalised)            augmentation.py:216-219
    double pmin0, pmin1;   // noise line                              augmentation.py:248-251
    double poly[3];        // simulation.py:467
    // wet model, estimation_method = 'poly' (augmentation.py:223-229, :243-246): quadratics in range instead of the two lines
    double pq[3];          // np.polyfit(dist, normalised, 2)
    double mq[3];          // ransac_polyfit(x, min_vals, order=2)
    int32_t rows_done;     // lean chain: histogram rows whose minimum has been taken (k_lean_rowmin_solve: the block that completes the frame fits its lines)
    int32_t quad;          // 1: k_wet_apply evaluates pq / mq
    int32_t ransac_trial;  // the trial whose consensus refit was kept (-1: the fit over all points)
    int32_t unchanged;     // wet path: < 1000 ground rows (augmentation.py:51-52)
    int32_t need_mean32;   // float32 rows and the noise line falls back to p (augmentation.py:250-251)
    // lean snowfall prepass (k_lean_*): centred second moments of (range, I / cos) and the sums of the quadratic fit
    double sxx, sxy;
    double q[11];          // LQ_* below
};
// (LQ_* : the sums of the quadratic fit, LP_* : the per-tile partials -- sg_lean.h)

struct PreArgs {
    const void *rows;
    const void *srows;          // optional (snowfall prepass): the channel sort's sorted copy of the frames that came unsorted ...
    const int32_t *frame_unsorted;   // ... and which frames those are (sg_common.h: SgBeamArgs)
    const int64_t *frame_off;
    const int64_t *frame_cnt;   // optional: rows actually present in frame f (compacted input); else off[f+1] - off[f]
    int n_frames;
    int64_t max_tiles;
    const double *plane;   // n_frames x 4
    double delta;          // ground band half width (0.5 in the snowfall path)
    int flat_earth;        // wet: incident angle from -z (augmentation.py:61-63)
    int cos_only;          // snowfall prepass: only cos(incident angle) is ever used -> g_ang holds the cosine itself and
                           // cos(arccos(c)) is taken as c (a relative difference of ~1e-16, far inside the prepass tolerance);
                           // saves an acos and two cos per ground row
    int rows_as_f64;       // wet: np.hstack with the float64 height column promotes the ground rows to float64
                           // (augmentation.py:50), so range / mean are float64 whatever the input dtype
    double noise_floor, power_factor;
    const double *lines_override;   // optional n_frames x 4 (p slope, p intercept, noise-line slope, intercept): replaces the two fitted lines
    double *qpart;         // estimation_method = 'poly': per tile the 8 sums of the quadratic fit of (range, I / cos)
    uint64_t seed;         // ... and the seed of its RANSAC draws
    // per-row scratch (n_total)
    double *g_dist, *g_norm, *g_ang;   // range, I / cos(angle), incident angle (or its cosine: cos_only); g_norm = NaN for non-ground rows
    // per-tile partials: [frame][tile][k]
    double *part;          // 12 doubles per tile
    int32_t *hist;         // [frame][HX][HY]
    double *rowmin;        // [frame][HX]  yedges[argmin] or -1
    float *cdist;          // ground ranges compacted in row order (float32 rows only)
    PreFrame *fr;
    int32_t *status;
};

__device__ __forceinline__ int64_t pre_rows(const PreArgs &a, int f)
{
    return a.frame_cnt ? a.frame_cnt[f] : a.frame_off[f + 1] - a.frame_off[f];
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
    return v;
}
// block reduction of K values in a fixed order: lanes -> waves -> wave 0
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *smem /* 4*K */)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) for (int k = 0; k < K; ++k) smem[w * K + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < K; ++k) v[k] = ((smem[k] + smem[K + k]) + smem[2 * K + k]) + smem[3 * K + k];
}

// ---- P1: ground rows, incident angle, normalised intensity; tile partials (count, sum x, sum y, max y) ------
template <typename T>
__global__ __launch_bounds__(PB) void k_pre_ground(PreArgs a)
{
    const int f = blockIdx.y;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const double *pl = a.plane + 4 * f;
    const double w0 = pl[0], w1 = pl[1], w2 = pl[2], h = pl[3];
    const double wn = sqrt((w0 * w0 + w1 * w1) + w2 * w2);              // np.linalg.norm(w)
    const T *rows = (const T *)a.rows;
    double v[3] = {0.0, 0.0, 0.0};
    double ymax = -INFINITY;
    T rx[4], ry[4], rz[4], ri[4];                // all loads of the tile in flight before the first use
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        const T *p = rows + (base + (r < n ? r : 0)) * 5;
        rx[q] = p[0]; ry[q] = p[1]; rz[q] = p[2]; ri[q] = p[3];
    }
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        if (r >= n) continue;
        const T x = rx[q], y = ry[q], z = rz[q], inten = ri[q];
        const double dot = ((double)x * w0 + (double)y * w1) + (double)z * w2;   // np.matmul(pc[:, :3], w)
        const double hog = dot + h;
        double gn = NAN, gd = 0.0, ga = 0.0;
        if (hog < a.delta && hog > -a.delta) {                           // simulation.py:450-451 / augmentation.py:46-47
            double nrm;
            if (sizeof(T) == 4 && !a.rows_as_f64) nrm = (double)sqrtf((float)((x * x + y * y) + z * z));   // float32 norm (simulation.py:455)
            else { const double xd = (double)x, yd = (double)y, zd = (double)z; nrm = sqrt((xd * xd + yd * yd) + zd * zd); }
            double c;
            if (a.flat_earth) c = -((double)z / (nrm * 1.0));            // augmentation.py:61-63
            else c = dot / (nrm * wn);                                   // simulation.py:454-455
            if (a.cos_only) {
                ga = fabs(c) <= 1.0 ? c : NAN;                           // arccos outside [-1, 1] is NaN in the reference too
                gn = (double)inten / ga;
            } else {
                ga = acos(c);
                gn = (double)inten / cos(ga);                            // augmentation.py:207
            }
            gd = nrm;                                                    // augmentation.py:208
            v[0] += 1.0; v[1] += gd; v[2] += gn;
            ymax = fmax(ymax, gn);
        }
        a.g_norm[base + r] = gn;                                         // NaN marks a non-ground row: range / angle are then never read
        if (gn == gn) { a.g_dist[base + r] = gd; a.g_ang[base + r] = ga; }
    }
    __shared__ double sm[12];
    __shared__ double smax[4];
    block_sum<3>(v, sm);
    ymax = wave_max(ymax);
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = ymax;
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = a.part + ((int64_t)f * a.max_tiles + blockIdx.x) * 12;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
        o[3] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    }
}

// One wave per frame: lane l sums tiles l, l + 64, ... in order, then the 64 lane totals are combined by a fixed
// shuffle tree -- deterministic, and ~64x shorter than one thread walking every tile.
template <int K>
__device__ __forceinline__ void frame_sums(const double *part, int64_t tiles, const int (&col)[K], double (&out)[K])
{
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < K; ++k) out[k] = 0.0;
    for (int64_t t = lane; t < tiles; t += 64)
        for (int k = 0; k < K; ++k) out[k] += part[t * 12 + col[k]];
    for (int k = 0; k < K; ++k) {
        for (int o = 32; o > 0; o >>= 1) out[k] += __shfl_xor(out[k], o);
    }
}

// ---- P2: per frame: counts, means, max ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pre_means(PreArgs a, int min_ground, int err_code)
{
    const int f = blockIdx.x;
    if (f >= a.n_frames) return;
    const int lane = threadIdx.x;
    const int64_t n = pre_rows(a, f);
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    double *part = a.part + (int64_t)f * a.max_tiles * 12;
    // exclusive prefix of the per-tile ground counts (exact: integers), max of the per-tile maxima
    double run = 0.0, ym = -INFINITY;
    for (int64_t t0 = 0; t0 < tiles; t0 += 64) {
        const int64_t t = t0 + lane;
        const double c = t < tiles ? part[t * 12 + 0] : 0.0;
        if (t < tiles) ym = fmax(ym, part[t * 12 + 3]);
        double inc = c;                                          // inclusive scan across the wave
        for (int o = 1; o < 64; o <<= 1) { const double v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (t < tiles) part[t * 12 + 6] = run + inc - c;         // ground rows in earlier tiles
        run += __shfl(inc, 63);
    }
    for (int o = 32; o > 0; o >>= 1) ym = fmax(ym, __shfl_xor(ym, o));
    const int cols[2] = {1, 2};
    double sums[2];
    frame_sums<2>(part, tiles, cols, sums);
    if (lane == 0) {
        const double c = run;
        PreFrame &fr = a.fr[f];
        fr.n_ground = c;
        fr.xmean = c > 0 ? sums[0] / c : 0.0;
        fr.ymean = c > 0 ? sums[1] / c : 0.0;
        fr.xmean32 = (double)(float)fr.xmean;   // refined by k_pre_mean32 when the value is actually used
        fr.need_mean32 = 0;
        fr.ymax = fabs(ym);                                              // np.abs(np.max(...)), augmentation.py:233
        fr.unchanged = 0;
        fr.quad = 0; fr.ransac_trial = -1;
        if (c < (double)min_ground) {
            if (err_code) atomicCAS(&a.status[0], 0, err_code);          // snowfall: TypeError in the reference (Q7)
            fr.unchanged = 1;                                            // wet: frame returned unchanged
        }
    }
}

// ---- P2b/P2c (float32 rows): np.mean(range) exactly as NumPy computes it ----------------------------------------
// scipy.stats.linregress uses np.mean(x) of the float32 range column for the intercept (augmentation.py:216);
// NumPy sums float32 with its pairwise scheme (blocks of <= 128 values, 8 interleaved accumulators, halves split
// at a multiple of 8) and divides in float32.  On frames where the laser-power line nearly cancels that rounding
// is visible in the rewritten intensities, so it is reproduced operation for operation: the ground ranges are
// first compacted in row order, then one block per frame walks NumPy's recursion.
__global__ __launch_bounds__(PB) void k_pre_gather(PreArgs a)
{
    const int f = blockIdx.y;
    if (!a.fr[f].need_mean32) return;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    __shared__ int wc[4][4];
    const int tid = threadIdx.x, wv = tid >> 6;
    const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
    bool g[4];
    int pre[4];
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + tid;
        const double gn = r < n ? a.g_norm[base + r] : NAN;
        g[q] = gn == gn;
        const unsigned long long m = __ballot(g[q]);
        pre[q] = __popcll(m & lt);
        if ((tid & 63) == 0) wc[q][wv] = __popcll(m);
    }
    __syncthreads();
    int run = (int)a.part[((int64_t)f * a.max_tiles + blockIdx.x) * 12 + 6];
    for (int q = 0; q < 4; ++q) {
        int off = run;
        for (int ww = 0; ww < wv; ++ww) off += wc[q][ww];
        if (g[q]) a.cdist[base + off + pre[q]] = (float)a.g_dist[base + tile0 + q * PB + tid];
        run += wc[q][0] + wc[q][1] + wc[q][2] + wc[q][3];
    }
}

__device__ __forceinline__ float np_leaf_sum_f32(const float *v, int n)     // n <= 128
{
    if (n < 8) {
        float res = -0.0f;
        for (int i = 0; i < n; ++i) res += v[i];
        return res;
    }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = v[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += v[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += v[i];
    return res;
}

__device__ void lean_solve_frame_fwd(const PreArgs &a, int f, double *thr_poly);

__global__ __launch_bounds__(PB) void k_pre_mean32(PreArgs a, int *leaf_buf, int max_leaves, double *thr_poly_or_null)
{
    const int f = blockIdx.x;
    const int n = (int)a.fr[f].n_ground;
    if (n <= 0 || !a.fr[f].need_mean32) return;
    const float *v = a.cdist + a.frame_off[f];
    int *leaf_off = leaf_buf + (int64_t)f * 3 * max_leaves;      // per frame: offsets, lengths, sums (as float bits)
    int *leaf_len = leaf_off + max_leaves;
    float *leaf_sum = (float *)(leaf_len + max_leaves);
    const int PW_MAX_LEAVES = max_leaves;
    __shared__ int n_leaves;
    if (threadIdx.x == 0) {                     // depth-first walk of pairwise_sum's recursion: leaves in order
        int stack_off[40], stack_len[40], sp = 0, nl = 0;
        stack_off[0] = 0; stack_len[0] = n; sp = 1;
        while (sp > 0 && nl < PW_MAX_LEAVES) {
            --sp;
            const int o = stack_off[sp], l = stack_len[sp];
            if (l <= 128) { leaf_off[nl] = o; leaf_len[nl] = l; ++nl; }
            else {
                int n2 = l / 2;
                n2 -= n2 % 8;
                stack_off[sp] = o + n2; stack_len[sp] = l - n2; ++sp;        // right half: visited second
                stack_off[sp] = o; stack_len[sp] = n2; ++sp;                 // left half: visited first
            }
        }
        n_leaves = nl;
    }
    __syncthreads();                            // (global writes of this block are visible to it after the barrier)
    for (int i = threadIdx.x; i < n_leaves; i += PB) leaf_sum[i] = np_leaf_sum_f32(v + leaf_off[i], leaf_len[i]);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {                     // sum(l) = sum(left) + sum(right): post-order over the same tree
        // explicit evaluation stack: entries are either pending sizes (>= 0) or the "add" marker (-1)
        int cmd[96], sp = 0, next_leaf = 0, vs = 0;
        float val[48];
        cmd[sp++] = n;
        while (sp > 0) {
            const int c = cmd[--sp];
            if (c == -1) { const float r = val[--vs]; const float l = val[--vs]; val[vs++] = l + r; }
            else if (c <= 128) { val[vs++] = leaf_sum[next_leaf++]; }
            else {
                int n2 = c / 2;
                n2 -= n2 % 8;
                cmd[sp++] = -1; cmd[sp++] = c - n2; cmd[sp++] = n2;          // evaluate left, then right, then add
            }
        }
        const float total = 0.0f + val[0];      // np.add.reduce: identity + pairwise sum
        PreFrame &fr = a.fr[f];
        fr.xmean32 = (double)(total / (float)n);                             // _mean: float32 true_divide
        fr.p1 = fr.ymean - fr.p0 * fr.xmean32;                               // linregress intercept (augmentation.py:216)
        fr.pmin1 = fr.p1;                                                    // pmin = p (:250-251)
        if (thr_poly_or_null) lean_solve_frame_fwd(a, f, thr_poly_or_null);  // the lean chain: the quadratic waited for this intercept
    }
}

// searchsorted(edges, v, side='right') - 1 on edges = linspace(lo, hi, nb + 1), last edge inclusive
// (np.histogramdd).  Edge k is k * step + lo, the last one exactly hi.
__device__ __forceinline__ int hist_bin(double v, double lo, double hi, int nb)
{
    if (!(v >= lo) || !(v <= hi)) return -1;
    const double step = (hi - lo) / nb;
    int k = (int)floor((v - lo) / step);
    if (k < 0) k = 0;
    if (k > nb) k = nb;
    // settle against the edge values NumPy compares with
    while (k > 0 && !(((k == nb) ? hi : (double)k * step + lo) <= v)) --k;
    while (k < nb && (((k + 1 == nb) ? hi : (double)(k + 1) * step + lo) <= v)) ++k;
    if (k >= nb) k = nb - 1;                                             // v == last edge
    return k;
}

// ---- P3: centred second moments (np.cov inside linregress) + the 50 x 2555 histogram ----------------------------
__global__ __launch_bounds__(PB) void k_pre_moments(PreArgs a)
{
    const int f = blockIdx.y;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const PreFrame fr = a.fr[f];
    double v[2] = {0.0, 0.0};
    int32_t *hist = a.hist + (int64_t)f * HX * HY;
    int key[4] = {-1, -1, -1, -1};
    double gnv[4], gdv[4];                       // all loads of the tile in flight before the first use
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        gnv[q] = r < n ? a.g_norm[base + r] : NAN;
    }
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        gdv[q] = gnv[q] == gnv[q] ? a.g_dist[base + r] : 0.0;
    }
    for (int q = 0; q < 4; ++q) {
        const double gn = gnv[q];
        if (gn != gn) continue;
        const double gd = gdv[q];
        const double dx = gd - fr.xmean, dy = gn - fr.ymean;
        v[0] += dx * dx; v[1] += dx * dy;
        const int bx = hist_bin(gd, 10.0, 70.0, HX);                     // augmentation.py:232-233
        const int by = hist_bin(gn, 5.0, fr.ymax, HY);
        if (bx >= 0 && by >= 0) key[q] = bx * HY + by;
    }
    // Neighbouring rows are neighbouring azimuths of one laser: same range, similar intensity -- most rows of a tile hit the
    // same few bins, and same-address atomics serialise in L2.  The tile's 1024 keys are first counted in an LDS hash table
    // (open addressing, 2048 slots), then every distinct bin is added to the frame's histogram once.  (Counting per wave
    // with a ballot loop, one round per distinct key, was three quarters of this kernel's instructions.)
    __shared__ int t_key[2048], t_cnt[2048];
    for (int i = threadIdx.x; i < 2048; i += PB) { t_key[i] = -1; t_cnt[i] = 0; }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int k = key[q];
        if (k < 0) continue;
        unsigned slot = ((unsigned)k * 2654435761u) >> 21;
        for (;;) {
            const int prev = atomicCAS(&t_key[slot], -1, k);
            if (prev == -1 || prev == k) { atomicAdd(&t_cnt[slot], 1); break; }
            slot = (slot + 1) & 2047u;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += PB)
        if (t_key[i] >= 0) atomicAdd(&hist[t_key[i]], t_cnt[i]);
    __shared__ double sm[8];
    block_sum<2>(v, sm);
    if (threadIdx.x == 0) {
        double *o = a.part + ((int64_t)f * a.max_tiles + blockIdx.x) * 12;
        o[4] = v[0]; o[5] = v[1];
    }
}

// ---- P4: per range row, the sparsest occupied bin (first one on ties) ------------------------------------------
// hist[hist == 0] = len(ground); ymins = argpartition(hist, 2, axis=1)[:, 0]  (augmentation.py:234-236).  NumPy's
// portable selection leaves the FIRST minimum there (argmin); an all-empty row gives bin 0.
__global__ __launch_bounds__(PB) void k_pre_rowmin(PreArgs a)
{
    const int f = blockIdx.y, row = blockIdx.x;
    const int32_t *h = a.hist + ((int64_t)f * HX + row) * HY;
    const PreFrame fr = a.fr[f];
    const int ng = (int)fr.n_ground;
    int best = 0x7fffffff, bidx = 0x7fffffff;
    constexpr int TRIPS = (HY + PB - 1) / PB;
    int cs[TRIPS];
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {                                    // the row's counts in one round of loads (a load per trip of a loop waited for each)
        const int b = threadIdx.x + i * PB;
        cs[i] = b < HY ? h[b] : -1;
    }
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        int c = cs[i];
        if (c < 0) continue;
        if (c == 0) c = ng;
        if (c < best) { best = c; bidx = threadIdx.x + i * PB; }         // ascending b per thread: first minimum
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int ob = __shfl_down(best, o), oi = __shfl_down(bidx, o);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    __shared__ int sb[4], si[4];
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sb[w] < best || (sb[w] == best && si[w] < bidx)) { best = sb[w]; bidx = si[w]; }
        const double step = (fr.ymax - 5.0) / HY;
        const double edge = (bidx == HY) ? fr.ymax : (double)bidx * step + 5.0;   // yedges[ymins], augmentation.py:237
        a.rowmin[(int64_t)f * HX + row] = edge;
    }
}

// linregress(x, y) for a handful of points: slope = cov / var, intercept = ymean - slope * xmean
__device__ __forceinline__ void small_linregress(const double *x, const double *y, int n, double &slope, double &icpt)
{
    double xm = 0, ym = 0;
    for (int i = 0; i < n; ++i) { xm += x[i]; ym += y[i]; }
    xm /= n; ym /= n;
    double sxx = 0, sxy = 0;
    for (int i = 0; i < n; ++i) { sxx += (x[i] - xm) * (x[i] - xm); sxy += (x[i] - xm) * (y[i] - ym); }
    slope = (sxy / n) / (sxx / n);
    icpt = ym - slope * xm;
}

// ---- P5: the two lines ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pre_lines(PreArgs a, int xmean_f32)
{
    const int f = blockIdx.x;
    if (f >= a.n_frames) return;
    PreFrame &fr = a.fr[f];
    const int64_t n = pre_rows(a, f);
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    const int cols[2] = {4, 5};
    double mom[2];
    frame_sums<2>(a.part + (int64_t)f * a.max_tiles * 12, tiles, cols, mom);
    // min_vals > 5 (augmentation.py:238), x = centres of the surviving range rows (:240-241): the lanes fetch the 50 row minima in one round
    // and squeeze them, in row order, into LDS by a ballot (thread 0 reading them one after the other into scratch arrays: 37 us)
    __shared__ double xs[HX], ys[HX];
    const int lane = threadIdx.x;
    const double mv = lane < HX ? a.rowmin[(int64_t)f * HX + lane] : 0.0;
    const bool keep = lane < HX && mv > 5;
    const unsigned long long mask = __ballot(keep);
    if (keep) {
        const double xstep = (70.0 - 10.0) / HX;
        const double e0 = (double)lane * xstep + 10.0;
        const double e1 = (lane + 1 == HX) ? 70.0 : (double)(lane + 1) * xstep + 10.0;
        const int pos = __popcll(mask & ((1ull << lane) - 1ull));
        xs[pos] = (e0 + e1) / 2; ys[pos] = mv;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int m = __popcll(mask);
    const double sxx = mom[0], sxy = mom[1];
    const double ng = fr.n_ground;
    double slope = 0, icpt = 0;
    if (ng >= 3) {
        slope = (sxy / ng) / (sxx / ng);                                 // scipy linregress: ssxym / ssxm
        // np.mean of a float32 column is a float32; the intercept is ymean - slope * xmean (augmentation.py:216)
        const double xm = xmean_f32 ? fr.xmean32 : fr.xmean;
        icpt = fr.ymean - slope * xm;
    }
    fr.p0 = slope; fr.p1 = icpt;
    if (m > 3) small_linregress(xs, ys, m, fr.pmin0, fr.pmin1);         // augmentation.py:248-249
    else { fr.pmin0 = slope; fr.pmin1 = icpt; fr.need_mean32 = xmean_f32; }   // :250-251
}

// The caller's lines instead of the fitted ones (snowgpu_set_wet_lines: a host that fits them with its own NumPy, quirk Q8).
__global__ void k_pre_override_lines(PreArgs a)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.n_frames) return;
    PreFrame &fr = a.fr[f];
    fr.p0 = a.lines_override[4 * f]; fr.p1 = a.lines_override[4 * f + 1];
    fr.pmin0 = a.lines_override[4 * f + 2]; fr.pmin1 = a.lines_override[4 * f + 3];
    fr.need_mean32 = 0;
}

// ================================================================================================================
// estimation_method = 'poly' of the wet-ground model (augmentation.py:223-229, :243-246, ransac_polyfit :171-192).
// Laser power: np.polyfit(range, I / cos, 2) over the ground rows -- least squares, here through the normal equations in the
// centred and scaled variable u = (d - 60) / 60 (ranges live in [0, 120] m: the 3 x 3 system is then well conditioned in float64;
// NumPy scales the Vandermonde columns and solves by SVD -- same minimiser, agreement ~1e-12 relative).
// Noise level: ransac_polyfit(x, min_vals, order=2) over the <= 50 range rows of the histogram whose sparsest bin lies above 5 --
// fit over all points first; then k = 100 trials: n = 15 indices drawn with replacement, a quadratic through them, its inliers
// (|residual| < t = 0.1), and if there are more than d = 15 of them and more than f = 0.8 of all points a refit on the inliers,
// kept when its summed absolute residual over the inliers undercuts the best so far (the first fit's is summed over ALL points,
// as in the reference).  The reference draws from NumPy's process-global, unseeded generator (np.random.randint, :183), so two runs
// of the reference disagree; here trial t of frame f draws from Philox4x32-10 keyed by (seed; f, t): same cloud + same seed = same
// curve, on every run and GPU.  Parity is therefore unpinned by construction (DESIGN.md section 9 says how it is tested instead).
#define PQ_C 60.0
#define PQ_S 60.0
#define PQ_COLS 8      /* sum u^4, u^3, u^2, u, 1, u^2 y, u y, y */
#define RQ_N 15
#define RQ_K 100
#define RQ_T 0.1
#define RQ_D 15
#define RQ_F 0.8

// least-squares quadratic from the 8 sums in u; returns false for a singular system (fewer than 3 distinct abscissae)
__device__ __forceinline__ bool quad_solve_u(const double *q, double &c2, double &c1, double &c0)
{
    double G[3][4] = {{q[0], q[1], q[2], q[5]}, {q[1], q[2], q[3], q[6]}, {q[2], q[3], q[4], q[7]}};
    for (int i = 0; i < 3; ++i) {                                        // Gaussian elimination, partial pivoting
        int piv = i;
        for (int r = i + 1; r < 3; ++r) if (fabs(G[r][i]) > fabs(G[piv][i])) piv = r;
        if (piv != i) for (int k = 0; k < 4; ++k) { const double t = G[i][k]; G[i][k] = G[piv][k]; G[piv][k] = t; }
        if (!(fabs(G[i][i]) > 1e-13 * (fabs(q[0]) + fabs(q[4]) + 1.0))) return false;
        for (int r = i + 1; r < 3; ++r) {
            const double m = G[r][i] / G[i][i];
            for (int k = i; k < 4; ++k) G[r][k] -= m * G[i][k];
        }
    }
    double x[3];
    for (int i = 2; i >= 0; --i) {
        double t = G[i][3];
        for (int k = i + 1; k < 3; ++k) t -= G[i][k] * x[k];
        x[i] = t / G[i][i];
    }
    c2 = x[0]; c1 = x[1]; c0 = x[2];
    return true;
}
// y = c2 u^2 + c1 u + c0 with u = (d - C) / S, as coefficients of d (highest power first, np.polyfit's order)
__device__ __forceinline__ void quad_u_to_d(double c2, double c1, double c0, double *out)
{
    const double a = c2 / (PQ_S * PQ_S), b = c1 / PQ_S;
    out[0] = a;
    out[1] = b - 2.0 * a * PQ_C;
    out[2] = (a * PQ_C * PQ_C - b * PQ_C) + c0;
}
__device__ __forceinline__ double quad_eval_u(double c2, double c1, double c0, double u) { return (c2 * u + c1) * u + c0; }

__global__ __launch_bounds__(PB) void k_pre_quad_part(PreArgs a)
{
    const int f = blockIdx.y;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    double v[PQ_COLS] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        const double gn = r < n ? a.g_norm[base + r] : NAN;
        if (gn != gn) continue;
        const double u = (a.g_dist[base + r] - PQ_C) * (1.0 / PQ_S), u2 = u * u;
        v[0] += u2 * u2; v[1] += u2 * u; v[2] += u2; v[3] += u; v[4] += 1.0; v[5] += u2 * gn; v[6] += u * gn; v[7] += gn;
    }
    __shared__ double sm[4 * PQ_COLS];
    block_sum<PQ_COLS>(v, sm);
    if (threadIdx.x == 0) {
        double *o = a.qpart + ((int64_t)f * a.max_tiles + blockIdx.x) * PQ_COLS;
        for (int k = 0; k < PQ_COLS; ++k) o[k] = v[k];
    }
}

// ransac_polyfit(x, y, order=2) (augmentation.py:171-192) by one block of 128 threads: the fit over all m points (every thread, same
// arithmetic), trial `tid` per thread, the reference's "first trial that reaches the smallest error" by thread 0.  xs / ys: the m <= 50
// points in shared memory.  Returns (thread 0 only) the coefficients in u = (x - PQ_C) / PQ_S and the winning trial (-1: the first fit).
__device__ __forceinline__ void ransac_quad_block(const double *xs, const double *ys, int m, uint64_t seed, uint64_t f, double *s_err,
                                                  double (*s_fit)[3], double &b2, double &b1, double &b0, int &win)
{
    const int tid = threadIdx.x;
    auto fit = [&](auto &&weight, double &c2, double &c1, double &c0) -> bool {     // least squares over the points with weight(i) copies
        double q[PQ_COLS] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < m; ++i) {
            const double w = weight(i);
            if (w == 0.0) continue;
            const double u = (xs[i] - PQ_C) * (1.0 / PQ_S), u2 = u * u, y = ys[i];
            q[0] += w * (u2 * u2); q[1] += w * (u2 * u); q[2] += w * u2; q[3] += w * u; q[4] += w; q[5] += w * (u2 * y); q[6] += w * (u * y); q[7] += w * y;
        }
        return quad_solve_u(q, c2, c1, c0);
    };
    // the fit over all points and its summed absolute residual (:179-180)
    double best_err = 0.0;
    b2 = 0; b1 = 0; b0 = 0;
    if (!fit([](int) { return 1.0; }, b2, b1, b0)) { b2 = 0.0; b1 = 0.0; double sy = 0; for (int i = 0; i < m; ++i) sy += ys[i]; b0 = sy / m; }
    for (int i = 0; i < m; ++i) best_err += fabs(quad_eval_u(b2, b1, b0, (xs[i] - PQ_C) * (1.0 / PQ_S)) - ys[i]);
    // trial `tid` (:182-191)
    double t_err = INFINITY, t2 = 0, t1 = 0, t0 = 0;
    if (tid < RQ_K) {
        unsigned char cnt[HX];
        for (int i = 0; i < HX; ++i) cnt[i] = 0;
        for (int d4 = 0; d4 < (RQ_N + 3) / 4; ++d4) {                    // n indices in [0, m), with replacement
            uint32_t u[4];
            philox_u32x4(seed, f, (uint32_t)(tid * 4 + d4), 0x504F4C59u /* "POLY" */, u);
            for (int k = 0; k < 4 && d4 * 4 + k < RQ_N; ++k) cnt[(int)(((uint64_t)u[k] * (uint64_t)m) >> 32)]++;
        }
        double m2, m1, m0;
        if (fit([&](int i) { return (double)cnt[i]; }, m2, m1, m0)) {
            unsigned long long inl = 0;
            int n_in = 0;
            for (int i = 0; i < m; ++i)
                if (fabs(quad_eval_u(m2, m1, m0, (xs[i] - PQ_C) * (1.0 / PQ_S)) - ys[i]) < RQ_T) { inl |= 1ull << i; ++n_in; }
            if (n_in > RQ_D && (double)n_in > (double)m * RQ_F && fit([&](int i) { return ((inl >> i) & 1ull) ? 1.0 : 0.0; }, t2, t1, t0)) {
                t_err = 0.0;
                for (int i = 0; i < m; ++i)
                    if ((inl >> i) & 1ull) t_err += fabs(quad_eval_u(t2, t1, t0, (xs[i] - PQ_C) * (1.0 / PQ_S)) - ys[i]);
            }
        }
    }
    s_err[tid] = t_err; s_fit[tid][0] = t2; s_fit[tid][1] = t1; s_fit[tid][2] = t0;
    __syncthreads();
    win = -1;
    if (tid == 0) {                                                      // the reference's loop keeps the FIRST trial that reaches the smallest error
        for (int t = 0; t < RQ_K; ++t)
            if (s_err[t] < best_err) { best_err = s_err[t]; win = t; }
        if (win >= 0) { b2 = s_fit[win][0]; b1 = s_fit[win][1]; b0 = s_fit[win][2]; }
    }
}

// One block of 128 threads per frame: the power quadratic from the tile sums (wave 0), the noise quadratic by RANSAC (one trial per thread).
__global__ __launch_bounds__(128) void k_pre_quad_fit(PreArgs a, int err_code)
{
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    PreFrame &fr = a.fr[f];
    if (fr.unchanged) return;
    __shared__ double xs[HX], ys[HX], s_err[128];
    __shared__ double s_fit[128][3];
    __shared__ int s_m;
    if (tid < 64) {                                                      // np.polyfit(dist, normalised, 2): fixed-order sums over the tiles
        const int64_t n = pre_rows(a, f);
        const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
        const double *part = a.qpart + (int64_t)f * a.max_tiles * PQ_COLS;
        double q[PQ_COLS] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t t = lane; t < tiles; t += 64)
            for (int k = 0; k < PQ_COLS; ++k) q[k] += part[t * PQ_COLS + k];
        for (int k = 0; k < PQ_COLS; ++k)
            for (int o = 32; o > 0; o >>= 1) q[k] += __shfl_xor(q[k], o);
        if (tid == 0) {
            double c2, c1, c0;
            if (quad_solve_u(q, c2, c1, c0)) quad_u_to_d(c2, c1, c0, fr.pq);
            else { fr.pq[0] = 0.0; fr.pq[1] = fr.p0; fr.pq[2] = fr.p1; }   // degenerate ranges: the regression line
            int m = 0;                                                   // min_vals > 5 (augmentation.py:238), x = centres of those range rows (:240-241)
            const double xstep = (70.0 - 10.0) / HX;
            for (int r = 0; r < HX; ++r) {
                const double mv = a.rowmin[(int64_t)f * HX + r];
                if (mv > 5) {
                    const double e0 = (double)r * xstep + 10.0;
                    const double e1 = (r + 1 == HX) ? 70.0 : (double)(r + 1) * xstep + 10.0;
                    xs[m] = (e0 + e1) / 2; ys[m] = mv; ++m;
                }
            }
            s_m = m;
        }
    }
    __syncthreads();
    const int m = s_m;
    if (m == 0) {                                                        // np.polyfit on an empty vector: TypeError in the reference (augmentation.py:179)
        if (tid == 0) { if (err_code) atomicCAS(&a.status[0], 0, err_code); fr.mq[0] = 0.0; fr.mq[1] = fr.pmin0; fr.mq[2] = fr.pmin1; fr.quad = 1; }
        return;
    }
    if (m < 3) {
        // One or two usable range rows: np.polyfit(x, y, 2) (:179) answers an under-determined system with the MINIMUM-NORM solution of its
        // column-scaled Vandermonde system (lstsq on lhs / sqrt(sum lhs^2), then c / scale) and a RankWarning; no RANSAC trial can replace
        // it (a consensus set needs more than d = 15 points, :187), so ransac_polyfit returns exactly that fit.  Columns x^2, x, 1.
        if (tid == 0) {
            double sc[3] = {0, 0, 0}, A[2][3];
            for (int i = 0; i < m; ++i) { const double x = xs[i]; sc[0] += (x * x) * (x * x); sc[1] += x * x; sc[2] += 1.0; }
            for (int k = 0; k < 3; ++k) sc[k] = sqrt(sc[k]);
            for (int i = 0; i < m; ++i) { const double x = xs[i]; A[i][0] = sc[0] > 0 ? x * x / sc[0] : 0.0; A[i][1] = sc[1] > 0 ? x / sc[1] : 0.0; A[i][2] = 1.0 / sc[2]; }
            double w[2] = {0, 0};
            if (m == 1) {
                const double g = A[0][0] * A[0][0] + A[0][1] * A[0][1] + A[0][2] * A[0][2];
                w[0] = ys[0] / g;
            } else {
                const double g00 = A[0][0] * A[0][0] + A[0][1] * A[0][1] + A[0][2] * A[0][2], g11 = A[1][0] * A[1][0] + A[1][1] * A[1][1] + A[1][2] * A[1][2];
                const double g01 = A[0][0] * A[1][0] + A[0][1] * A[1][1] + A[0][2] * A[1][2], det = g00 * g11 - g01 * g01;
                if (fabs(det) > 1e-14 * g00 * g11) { w[0] = (g11 * ys[0] - g01 * ys[1]) / det; w[1] = (g00 * ys[1] - g01 * ys[0]) / det; }
                else { w[0] = w[1] = 0.5 * (ys[0] + ys[1]) / (g00 + g01); }          // the same abscissa twice: rank one
            }
            for (int k = 0; k < 3; ++k) {
                double c = 0;
                for (int i = 0; i < m; ++i) c += A[i][k] * w[i];
                fr.mq[k] = sc[k] > 0 ? c / sc[k] : 0.0;
            }
            fr.quad = 1;
        }
        return;
    }
    double b2, b1, b0;
    int win;
    ransac_quad_block(xs, ys, m, a.seed, (uint64_t)f, s_err, s_fit, b2, b1, b0, win);
    if (tid == 0) { quad_u_to_d(b2, b1, b0, fr.mq); fr.quad = 1; fr.ransac_trial = win; }
}

// debug / parity tap: ransac_polyfit on the caller's points (m <= 50), draws of (seed; frame): out = c2, c1, c0 (coefficients of x), trial kept
__global__ __launch_bounds__(128) void k_debug_ransac_quad(const double *x, const double *y, int m, uint64_t seed, uint64_t frame, double *out)
{
    __shared__ double xs[HX], ys[HX], s_err[128];
    __shared__ double s_fit[128][3];
    for (int i = threadIdx.x; i < m; i += 128) { xs[i] = x[i]; ys[i] = y[i]; }
    __syncthreads();
    double b2, b1, b0;
    int win;
    ransac_quad_block(xs, ys, m, seed, frame, s_err, s_fit, b2, b1, b0, win);
    if (threadIdx.x == 0) { quad_u_to_d(b2, b1, b0, out); out[3] = (double)win; }
}

extern "C" int sg_debug_ransac_quad(const double *d_x, const double *d_y, int m, uint64_t seed, uint64_t frame, double *d_out, void *stream)
{
    hipLaunchKernelGGL(k_debug_ransac_quad, dim3(1), dim3(128), 0, (hipStream_t)stream, d_x, d_y, m, seed, frame, d_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// the fitted curves of the last estimate, per frame: (power c2, c1, c0, noise c2, c1, c0, ground rows, RANSAC trial kept or -1);
// 'linear' frames report their lines as quadratics with c2 = 0
__global__ void k_pre_export_fit(PreArgs a, double *out)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.n_frames) return;
    const PreFrame &fr = a.fr[f];
    double *o = out + (int64_t)f * 8;
    if (fr.quad) { for (int k = 0; k < 3; ++k) { o[k] = fr.pq[k]; o[3 + k] = fr.mq[k]; } }
    else { o[0] = 0.0; o[1] = fr.p0; o[2] = fr.p1; o[3] = 0.0; o[4] = fr.pmin0; o[5] = fr.pmin1; }
    o[6] = fr.n_ground; o[7] = (double)fr.ransac_trial;
}

// ================================================================================================================
// Lean snowfall prepass.  The chain above (kept by the wet-ground model, whose per-row rewrite needs range, angle and I / cos of every
// ground row again) moves every ground row through three float64 scratch arrays -- written once, read twice: 2.8 GB per 256-sweep
// step when the snowfall path used it too (rounds 1 - 3), more than the per-beam kernels fetch.  The snowfall path needs less: k_lean_stats streams the rows ONCE and leaves, per 1024-row tile, everything that is a plain sum --
// count, sums and tile-centred second moments of (range, I / cos) for the regression line, the maximum for the histogram
// range, and the sums of the quadratic fit, which are LINEAR in the noise line (y_i = nf (pmin0 d_i + pmin1) c_i, so
// sum a y = nf (pmin0 sum a d c + pmin1 sum a c)) and can therefore be taken before the line is known; k_lean_hist streams
// the rows a second time for the 50 x 2555 histogram (its bin edges need the maximum), recomputing the three per-row values
// instead of loading them.  Tiles are combined per frame in a fixed order (Chan's pairwise update for the centred
// moments), so a batch is reproducible run to run.  No per-row scratch at all.
template <typename T>
__global__ __launch_bounds__(PB) void k_lean_stats(PreArgs a)
{
    const int f = blockIdx.y;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const T *rows = (const T *)a.rows;
    T rx[4], ry[4], rz[4], ri[4];                // all loads of the tile in flight before the first use
    bool valid[4];
    // rows to threads as the channel sort's first kernel deals them (wave w: rows [256 w, 256 w + 256) of the tile in four rounds of 64): the
    // statistics taken there and here are then the same sums in the same order -- the same bits whichever kernel a call uses
    const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + w * 256 + q * 64 + lane;
        valid[q] = r < n;
        const T *p = rows + (base + (valid[q] ? r : 0)) * 5;
        rx[q] = p[0]; ry[q] = p[1]; rz[q] = p[2]; ri[q] = p[3];
    }
    __shared__ double sm[58];
    SgLeanTile lt;
    lt.plane = a.plane; lt.delta = a.delta; lt.part = a.part; lt.max_tiles = a.max_tiles;
    lean_tile_stats<T>(lt, f, blockIdx.x, rx, ry, rz, ri, valid, sm);
}

// one wave per frame: exclusive prefix of the tile counts, means, maximum, centred moments and the fit's sums
__global__ __launch_bounds__(64) void k_lean_means(PreArgs a, int min_ground, int err_code)
{
    const int f = blockIdx.x;
    if (f >= a.n_frames) return;
    const int lane = threadIdx.x;
    const int64_t n = pre_rows(a, f);
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    double *part = a.part + (int64_t)f * a.max_tiles * LP_COLS;
    double run = 0.0, ym = -INFINITY, sx = 0.0, sy = 0.0;
    for (int64_t t0 = 0; t0 < tiles; t0 += 64) {
        const int64_t t = t0 + lane;
        const double c = t < tiles ? part[t * LP_COLS + LP_N] : 0.0;
        if (t < tiles) { ym = fmax(ym, part[t * LP_COLS + LP_YMAX]); sx += part[t * LP_COLS + LP_SX]; sy += part[t * LP_COLS + LP_SY]; }
        double inc = c;                                          // inclusive scan across the wave (exact: integers)
        for (int o = 1; o < 64; o <<= 1) { const double v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (t < tiles) part[t * LP_COLS + LP_PREFIX] = run + inc - c;    // ground rows in earlier tiles
        run += __shfl(inc, 63);
    }
    for (int o = 32; o > 0; o >>= 1) { ym = fmax(ym, __shfl_xor(ym, o)); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); }
    const double cnt = run;
    const double xm = cnt > 0 ? sx / cnt : 0.0, ymn = cnt > 0 ? sy / cnt : 0.0;
    // M2 = sum over tiles of (tile-centred M2 + n_t (tile mean - frame mean)^2): lane l takes tiles l, l + 64, .., fixed tree
    double mxx = 0.0, mxy = 0.0, q[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t t = lane; t < tiles; t += 64) {
        const double *o = part + t * LP_COLS;
        const double nt = o[LP_N];
        if (nt > 0) {
            const double dxm = o[LP_SX] / nt - xm, dym = o[LP_SY] / nt - ymn;
            mxx += o[LP_MXX] + nt * (dxm * dxm);
            mxy += o[LP_MXY] + nt * (dxm * dym);
        }
        for (int k = 0; k < 11; ++k) q[k] += o[LP_Q0 + k];
    }
    for (int o = 32; o > 0; o >>= 1) {
        mxx += __shfl_xor(mxx, o); mxy += __shfl_xor(mxy, o);
        for (int k = 0; k < 11; ++k) q[k] += __shfl_xor(q[k], o);
    }
    if (lane == 0) {
        PreFrame &fr = a.fr[f];
        fr.n_ground = cnt;
        fr.xmean = xm; fr.ymean = ymn;
        fr.xmean32 = (double)(float)xm;         // refined by k_pre_mean32 when the value is actually used
        fr.need_mean32 = 0;
        fr.ymax = fabs(ym);                                              // np.abs(np.max(...)), augmentation.py:233
        fr.unchanged = 0;
        fr.sxx = mxx; fr.sxy = mxy;
        fr.rows_done = 0;
        for (int k = 0; k < 11; ++k) fr.q[k] = q[k];
        if (cnt < (double)min_ground) {
            if (err_code) atomicCAS(&a.status[0], 0, err_code);          // TypeError in the reference (Q7)
            fr.unchanged = 1;
        }
    }
}

// the 50 x 2555 histogram from the rows themselves (second and last pass over them)
template <typename T>
__global__ __launch_bounds__(PB) void k_lean_hist(PreArgs a)
{
    const int f = blockIdx.y;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    // A frame that came unsorted (firing order) is read from the sort's sorted copy: neighbouring rows are then neighbouring azimuths
    // of one laser again and fall into the same few bins (the tile's LDS table below), where a firing-order tile holds 64 lasers'
    // rows and as many distinct bins (measured: 1.07 instead of 0.88 ms, beside the tiers).  The histogram does not depend on the
    // row order.  The per-tile ground counts describe the tiles of the ARRIVAL order, so only sorted frames can skip by them.
    const bool uns = a.frame_unsorted && a.frame_unsorted[f];
    if (!uns && a.part[((int64_t)f * a.max_tiles + blockIdx.x) * LP_COLS + LP_N] == 0.0) return;       // no ground row in this tile
    const double *pl = a.plane + 4 * f;
    const double w0 = pl[0], w1 = pl[1], w2 = pl[2], h = pl[3];
    const double wn = sqrt((w0 * w0 + w1 * w1) + w2 * w2);
    const double ymaxv = a.fr[f].ymax;
    const T *rows = (const T *)(uns ? a.srows : a.rows);
    int32_t *hist = a.hist + (int64_t)f * HX * HY;
    T rx[4], ry[4], rz[4], ri[4];
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        const T *p = rows + (base + (r < n ? r : 0)) * 5;
        rx[q] = p[0]; ry[q] = p[1]; rz[q] = p[2]; ri[q] = p[3];
    }
    int key[4] = {-1, -1, -1, -1};
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        double gd, gn, gc;
        if (r < n && lean_row<T>(a.delta, w0, w1, w2, h, wn, rx[q], ry[q], rz[q], ri[q], gd, gn, gc) && gn == gn) {
            const int bx = hist_bin(gd, 10.0, 70.0, HX);                 // augmentation.py:232-233
            const int by = hist_bin(gn, 5.0, ymaxv, HY);
            if (bx >= 0 && by >= 0) key[q] = bx * HY + by;
        }
    }
    // the tile's keys are counted in an LDS hash table first, then one atomic per distinct bin (see k_pre_moments)
    __shared__ int t_key[2048], t_cnt[2048];
    for (int i = threadIdx.x; i < 2048; i += PB) { t_key[i] = -1; t_cnt[i] = 0; }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int k = key[q];
        if (k < 0) continue;
        unsigned slot = ((unsigned)k * 2654435761u) >> 21;
        for (;;) {
            const int prev = atomicCAS(&t_key[slot], -1, k);
            if (prev == -1 || prev == k) { atomicAdd(&t_cnt[slot], 1); break; }
            slot = (slot + 1) & 2047u;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += PB)
        if (t_key[i] >= 0) atomicAdd(&hist[t_key[i]], t_cnt[i]);
}

// the two lines from the frame's centred moments (k_lean_means) and the histogram's row minima (one thread per frame)
__device__ __forceinline__ void lean_lines_frame(const PreArgs &a, int f, int xmean_f32)
{
    PreFrame &fr = a.fr[f];
    const double ng = fr.n_ground;
    double slope = 0, icpt = 0;
    if (ng >= 3) {
        slope = (fr.sxy / ng) / (fr.sxx / ng);                           // scipy linregress: ssxym / ssxm
        const double xm = xmean_f32 ? fr.xmean32 : fr.xmean;             // np.mean of a float32 column is a float32 (augmentation.py:216)
        icpt = fr.ymean - slope * xm;
    }
    fr.p0 = slope; fr.p1 = icpt;
    double xs[HX], ys[HX];
    int m = 0;
    const double xstep = (70.0 - 10.0) / HX;
    for (int r = 0; r < HX; ++r) {
        const double mv = ((const volatile double *)a.rowmin)[(int64_t)f * HX + r];   // (written by other blocks of this launch in k_lean_rowmin_solve: past the L1)
        if (mv > 5) {                                                    // augmentation.py:238
            const double e0 = (double)r * xstep + 10.0;
            const double e1 = (r + 1 == HX) ? 70.0 : (double)(r + 1) * xstep + 10.0;
            xs[m] = (e0 + e1) / 2; ys[m] = mv; ++m;                      // :240-241
        }
    }
    if (m > 3) small_linregress(xs, ys, m, fr.pmin0, fr.pmin1);         // augmentation.py:248-249
    else { fr.pmin0 = slope; fr.pmin1 = icpt; fr.need_mean32 = xmean_f32; }   // :250-251
}

__global__ __launch_bounds__(64) void k_lean_lines(PreArgs a, int xmean_f32)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f < a.n_frames) lean_lines_frame(a, f, xmean_f32);
}

// ground ranges compacted in row order for the frames that need NumPy's float32 mean (k_pre_mean32), from the rows
template <typename T>
__global__ __launch_bounds__(PB) void k_lean_gather(PreArgs a)
{
    const int f = blockIdx.y;
    if (!a.fr[f].need_mean32) return;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const double *pl = a.plane + 4 * f;
    const double w0 = pl[0], w1 = pl[1], w2 = pl[2], h = pl[3];
    const double wn = sqrt((w0 * w0 + w1 * w1) + w2 * w2);
    const T *rows = (const T *)a.rows;
    __shared__ int wc[4][4];
    const int tid = threadIdx.x, wv = tid >> 6;
    const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
    bool g[4];
    int pre[4];
    double gdv[4];
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + tid;
        double gn, gc;
        g[q] = false;
        if (r < n) {
            const T *p = rows + (base + r) * 5;
            g[q] = lean_row<T>(a.delta, w0, w1, w2, h, wn, p[0], p[1], p[2], p[3], gdv[q], gn, gc) && gn == gn;
        }
        const unsigned long long m = __ballot(g[q]);
        pre[q] = __popcll(m & lt);
        if ((tid & 63) == 0) wc[q][wv] = __popcll(m);
    }
    __syncthreads();
    int run = (int)a.part[((int64_t)f * a.max_tiles + blockIdx.x) * LP_COLS + LP_PREFIX];
    for (int q = 0; q < 4; ++q) {
        int off = run;
        for (int ww = 0; ww < wv; ++ww) off += wc[q][ww];
        if (g[q]) a.cdist[base + off + pre[q]] = (float)gdv[q];
        run += wc[q][0] + wc[q][1] + wc[q][2] + wc[q][3];
    }
}

// the quadratic from the frame's sums and its noise line (columns scaled by their norms, as np.polyfit does); one thread per frame
__device__ __forceinline__ void lean_solve_frame(const PreArgs &a, int f, double *thr_poly);
__device__ void lean_solve_frame_fwd(const PreArgs &a, int f, double *thr_poly) { lean_solve_frame(a, f, thr_poly); }
__device__ __forceinline__ void lean_solve_frame(const PreArgs &a, int f, double *thr_poly)
{
    const PreFrame &fr = a.fr[f];
    double *out = thr_poly + 3 * f;
    const double nn = fr.n_ground;
    if (nn < 3) { out[0] = out[1] = out[2] = 0.0; return; }
    const double *q = fr.q;
    const double nf = a.noise_floor, m0 = fr.pmin0, m1 = fr.pmin1;
    // sum a y with y = (nf (pmin0 d + pmin1)) c  (augmentation.py:252-253, simulation.py:462): linear in the line
    const double s5 = nf * (m0 * q[LQ_A2GC] + m1 * q[LQ_A2C]);
    const double s6 = nf * (m0 * q[LQ_A1GC] + m1 * q[LQ_A1C]);
    const double s7 = nf * (m0 * q[LQ_GC] + m1 * q[LQ_C]);
    const double c2 = sqrt(q[LQ_A2A2]), c1 = sqrt(q[LQ_A1A1]), c0 = sqrt(nn);
    double G[3][4] = {{q[LQ_A2A2] / (c2 * c2), q[LQ_A2A1] / (c2 * c1), q[LQ_A2] / (c2 * c0), s5 / c2},
                      {q[LQ_A2A1] / (c1 * c2), q[LQ_A1A1] / (c1 * c1), q[LQ_A1] / (c1 * c0), s6 / c1},
                      {q[LQ_A2] / (c0 * c2), q[LQ_A1] / (c0 * c1), nn / (c0 * c0), s7 / c0}};
    for (int i = 0; i < 3; ++i) {                                        // Gaussian elimination, partial pivoting
        int piv = i;
        for (int r = i + 1; r < 3; ++r) if (fabs(G[r][i]) > fabs(G[piv][i])) piv = r;
        if (piv != i) for (int k = 0; k < 4; ++k) { const double t = G[i][k]; G[i][k] = G[piv][k]; G[piv][k] = t; }
        for (int r = i + 1; r < 3; ++r) {
            const double m = G[r][i] / G[i][i];
            for (int k = i; k < 4; ++k) G[r][k] -= m * G[i][k];
        }
    }
    double x[3];
    for (int i = 2; i >= 0; --i) {
        double t = G[i][3];
        for (int k = i + 1; k < 3; ++k) t -= G[i][k] * x[k];
        x[i] = t / G[i][i];
    }
    out[0] = x[0] / c2; out[1] = x[1] / c1; out[2] = x[2] / c0;
    a.fr[f].poly[0] = out[0]; a.fr[f].poly[1] = out[1]; a.fr[f].poly[2] = out[2];
}

// The two lines and (unless the frame needs NumPy's float32 mean first) the quadratic of one frame by ONE WAVE (all 64 lanes call it): the
// lanes fetch the frame's 50 row minima in one round and squeeze out the rows without one (augmentation.py:238) by a ballot, in row order,
// into LDS (xs, ys: HX doubles each); lane 0 then runs the fits on them -- lean_lines_frame's statements.  (A thread per frame read the
// minima one after the other, 50 round trips, into arrays that lived in scratch memory: 46 us at the end of the prepass chain of a
// 256-sweep step, which ends the step's side branch.)  VOL: the minima were written by other blocks of the running launch -- read past the L1.
template <bool VOL>
__device__ __forceinline__ void lean_lines_wave(const PreArgs &a, int f, int xmean_f32, double *xs, double *ys, double *thr_poly)
{
    static_assert(HX <= 64, "one lane per range row");
    const int lane = threadIdx.x & 63;
    double mv = 0.0;
    if (lane < HX) mv = VOL ? ((const volatile double *)a.rowmin)[(int64_t)f * HX + lane] : a.rowmin[(int64_t)f * HX + lane];
    const bool keep = lane < HX && mv > 5;                               // augmentation.py:238
    const unsigned long long mask = __ballot(keep);
    if (keep) {
        const double xstep = (70.0 - 10.0) / HX;
        const double e0 = (double)lane * xstep + 10.0;
        const double e1 = (lane + 1 == HX) ? 70.0 : (double)(lane + 1) * xstep + 10.0;
        const int pos = __popcll(mask & ((1ull << lane) - 1ull));
        xs[pos] = (e0 + e1) / 2; ys[pos] = mv;                           // :240-241
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // written and read by the lanes of one wave: no barrier, but keep the order
    if (lane != 0) return;
    const int m = __popcll(mask);
    PreFrame &fr = a.fr[f];
    const double ng = fr.n_ground;
    double slope = 0, icpt = 0;
    if (ng >= 3) {
        slope = (fr.sxy / ng) / (fr.sxx / ng);                           // scipy linregress: ssxym / ssxm
        const double xm = xmean_f32 ? fr.xmean32 : fr.xmean;             // np.mean of a float32 column is a float32 (augmentation.py:216)
        icpt = fr.ymean - slope * xm;
    }
    fr.p0 = slope; fr.p1 = icpt;
    if (m > 3) small_linregress(xs, ys, m, fr.pmin0, fr.pmin1);         // augmentation.py:248-249
    else { fr.pmin0 = slope; fr.pmin1 = icpt; fr.need_mean32 = xmean_f32; }   // :250-251
    if (!fr.need_mean32) lean_solve_frame(a, f, thr_poly);
}

// ... after k_pre_rowmin (large batches): one wave per frame
__global__ __launch_bounds__(64) void k_lean_lines_solve(PreArgs a, int xmean_f32, double *thr_poly)
{
    const int f = blockIdx.x;
    if (f >= a.n_frames) return;
    __shared__ double xs[HX], ys[HX];
    lean_lines_wave<false>(a, f, xmean_f32, xs, ys, thr_poly);
}

// Small batches (up to 16 frames: bound by their chain of dependent launches): row minima of a frame's histogram (one block per range
// row: 50 per frame) and -- by whichever block completes the frame -- the two lines and, unless the frame needs NumPy's float32 mean
// first (k_lean_gather, k_pre_mean32), the quadratic, in ONE launch (a single 1024-thread block per frame did this before: 78 us of a
// single sweep's 460, the longest link of its prepass chain).  Large batches keep k_pre_rowmin + k_lean_lines_solve: "the block that
// completes the frame" costs a device-scope fence per block, and on this chip -- eight XCDs, each with its own L2 -- such a fence writes
// the XCD's L2 back: 12 800 of them made this kernel 0.74 ms long on 256 sweeps, against 0.10 ms for the two launches.
__global__ __launch_bounds__(PB) void k_lean_rowmin_solve(PreArgs a, int xmean_f32, double *thr_poly)
{
    const int f = blockIdx.y, row = blockIdx.x;
    const int32_t *h = a.hist + ((int64_t)f * HX + row) * HY;
    PreFrame &fr = a.fr[f];
    const int ng = (int)fr.n_ground;
    int best = 0x7fffffff, bidx = 0x7fffffff;
    for (int b = threadIdx.x; b < HY; b += PB) {
        int c = h[b];
        if (c == 0) c = ng;                                                  // hist[hist == 0] = len(ground) (augmentation.py:234-235)
        if (c < best) { best = c; bidx = b; }                                // ascending b per thread: first minimum
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int ob = __shfl_down(best, o), oi = __shfl_down(bidx, o);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    __shared__ int sb[4], si[4], s_last;
    __shared__ double s_xs[HX], s_ys[HX];
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bidx; }
    if (threadIdx.x == 0) s_last = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sb[w] < best || (sb[w] == best && si[w] < bidx)) { best = sb[w]; bidx = si[w]; }
        const double step = (fr.ymax - 5.0) / HY;
        a.rowmin[(int64_t)f * HX + row] = (bidx == HY) ? fr.ymax : (double)bidx * step + 5.0;   // yedges[ymins] (:237)
        __threadfence();                                                     // the row's minimum before the count that announces it
        s_last = atomicAdd(&fr.rows_done, 1) == HX - 1;                      // this block completed the frame
        if (s_last) __threadfence();
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) lean_lines_wave<true>(a, f, xmean_f32, s_xs, s_ys, thr_poly);   // (one thread did this: 50 loads in a row)
}

// ================================================================================================================
// wet ground (augmentation.py:88-159; phy_equations.py:35-108)

// phy_equations.py:35-67 fresnel_power(ain, n_in, n_out), from the sine and cosine of the incidence angle instead of the angle: the
// refraction angle only ever enters as its sine -- n_in / n_out sin(ain), clipped (:41-43) -- and its cosine, the root of 1 - sin^2
// (cos(arcsin(s)), :44-46), and the angle the chain hands on (total_transmittance, :81-83) is used the same way, so no arcsin, sine or
// cosine is taken here at all; the two amplitude pairs share their denominators' reciprocals, and `frac` (:47) enters as its reciprocal.
// The same numbers to a few 1e-16 (the wet path's intensities are float64 values compared at 1e-9 / 1e-7: tests/test_gpu_parity.py
// ::test_L6_wet_ground, test_gpu_fullsize.py) -- the library sin / arcsin / cos and 18 divisions per row were 0.88 ms of a fused
// 256-sweep step.
struct Fresnel { double rs, ts, rp, tp, s_out, c_out; };
__device__ __forceinline__ Fresnel fresnel_power(double si, double ci, double n_in, double n_out)
{
    Fresnel r;
    double s = si * n_in / n_out;                                    // :41
    s = s < -1 ? -1 : (s > 1 ? 1 : s);                               // :42-43
    const double co = sqrt(1.0 - s * s);                             // cos(aout), aout = arcsin(s) (:44-46)
    r.s_out = s; r.c_out = co;
    const double a = n_in * ci, b = n_out * co, c = n_out * ci, d = n_in * co;
    const double i1 = 1.0 / (a + b), i2 = 1.0 / (c + d);
    const double rs = (a - b) * i1, ts = 2 * a * i1;                 // :49-50
    const double rp = (c - d) * i2, tp = 2 * a * i2;                 // :51-52
    const double inv_frac = (n_out * co) / (ci * n_in);              // 1 / (cos(ain) n_in / n_out / cos(aout)) (:47)
    r.rs = rs * rs; r.ts = ts * ts * inv_frac; r.rp = rp * rp; r.tp = tp * tp * inv_frac;   // :54-57
    return r;
}

struct WetArgs {
    PreArgs p;
    double water_height, pavement_depth;
    int replace;
    uint8_t *cls;          // per row: 1 = non-ground, 2 = kept ground, 0 = dropped
    double *new_i;         // per row: rewritten intensity
    int32_t *tile_cnt;     // [frame][tile][2]
    int32_t *tile_base;    // [frame][tile][2]
    double *out_rows;
    int32_t *out_src;
    int64_t *out_counts;
    int32_t *out_flags;
    const int32_t *src_first;   // SgWetParams::src_first
};

template <typename T>
__global__ __launch_bounds__(PB) void k_wet_apply(WetArgs w)
{
    const PreArgs &a = w.p;
    const int f = blockIdx.y;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const PreFrame fr = a.fr[f];
    const T *rows = (const T *)a.rows;
    int cnt_a = 0, cnt_b = 0;
    double gns[4], gds[4], angs[4], ins[4];                              // the thread's four rows side by side: every load first
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        const bool in = r < n && !fr.unchanged;
        gns[q] = in ? a.g_norm[base + r] : NAN;
        const bool ground = gns[q] == gns[q];
        gds[q] = ground ? a.g_dist[base + r] : 0.0;
        angs[q] = ground ? a.g_ang[base + r] : 1.0;
        ins[q] = ground ? (double)rows[(base + r) * 5 + 3] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + threadIdx.x;
        if (r >= n) continue;
        uint8_t cls;
        double ni = 0.0;
        const double gn = gns[q];
        if (fr.unchanged) { cls = 1; }                                   // frame returned as is (augmentation.py:51-52)
        else if (gn != gn) { cls = 1; }
        else {
            const double gd = gds[q], ang = angs[q];
            double gs, gc;
            sg_sincos_0_2pi(ang, gs, gc);                                // the incidence angle lies in [0, pi] (an arccos)
            const double inten = ins[q];
            double rel, thr;
            if (fr.quad) {                                               // estimation_method = 'poly'
                const double gd2 = gd * gd;
                rel = a.power_factor * ((fr.pq[0] * gd2 + fr.pq[1] * gd) + fr.pq[2]);        // :228-229
                thr = a.noise_floor * ((fr.mq[0] * gd2 + fr.mq[1] * gd) + fr.mq[2]);         // :245-246
            } else {
                rel = a.power_factor * (fr.p0 * gd + fr.p1);             // :221
                thr = a.noise_floor * (fr.pmin0 * gd + fr.pmin1);        // :252-253
            }
            const double refl = inten / gc / rel;                        // :90
            double rho = refl < 0.05 ? 0.05 : (refl > 1 ? 1 : refl);     // :109 np.clip(reflectivities, 0.05, 1)
            const Fresnel aw = fresnel_power(gs, gc, 1.0003, 1.33);      // phy_equations.py:81
            const Fresnel wa = fresnel_power(aw.s_out, aw.c_out, 1.33, 1.0003);   // :83 (the angle inside the water)
            const double ts = aw.ts * rho * wa.ts / (1 - rho * wa.rs);   // :86
            const double tp = aw.tp * rho * wa.tp / (1 - rho * wa.rp);   // :89
            const double t = fmax(tp, ts);                               // augmentation.py:119
            double fw = w.water_height / w.pavement_depth;               // :122
            fw = fw < 0 ? 0 : (fw > 1 ? 1 : fw);
            const double tw = (1 - fw) * refl + fw * t / ang;            // :123
            double v = rel * gc * tw;                                    // :126
            v = v < 0 ? 0 : (v > inten ? inten : v);                     // np.clip(., 0, intensity)
            const double lim = thr * gc;
            if (v < lim) v = 0;                                          // :128, :131
            ni = v;
            cls = (v > lim) ? 2 : 0;                                     // :146
        }
        w.cls[base + r] = cls;
        if (cls == 2) w.new_i[base + r] = ni;                            // (read back for kept ground rows only: k_wet_scatter)
        cnt_a += cls == 1; cnt_b += cls == 2;
    }
    __shared__ int sa[4], sb[4];
    for (int o = 32; o > 0; o >>= 1) { cnt_a += __shfl_down(cnt_a, o); cnt_b += __shfl_down(cnt_b, o); }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = cnt_a; sb[threadIdx.x >> 6] = cnt_b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t *o = w.tile_cnt + ((int64_t)f * a.max_tiles + blockIdx.x) * 2;
        o[0] = sa[0] + sa[1] + sa[2] + sa[3];
        o[1] = sb[0] + sb[1] + sb[2] + sb[3];
    }
}

// per frame: tile offsets of the two output runs ([non-ground ; kept ground], augmentation.py:147-150).  One WAVE per frame: lane l takes
// tiles l, l + 64, .. (a thread per frame walked its 128 tiles twice, load after load: 52 us of a 256-sweep step).
__global__ __launch_bounds__(64) void k_wet_scan(WetArgs w)
{
    const PreArgs &a = w.p;
    const int f = blockIdx.x, lane = threadIdx.x;
    const int64_t n = pre_rows(a, f);
    const int64_t tiles = (n + SG_TILE - 1) / SG_TILE;
    const int32_t *c = w.tile_cnt + (int64_t)f * a.max_tiles * 2;
    int32_t *b = w.tile_base + (int64_t)f * a.max_tiles * 2;
    int na = 0, nb = 0;
    for (int64_t t0 = 0; t0 < tiles; t0 += 64) {
        const int64_t t = t0 + lane;
        const int ca = t < tiles ? c[2 * t] : 0, cb = t < tiles ? c[2 * t + 1] : 0;
        int ia = ca, ib = cb;
        for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_up(ia, o), y = __shfl_up(ib, o);
            if (lane >= o) { ia += x; ib += y; }
        }
        if (t < tiles) { b[2 * t] = na + ia - ca; b[2 * t + 1] = nb + ib - cb; }
        na += __shfl(ia, 63); nb += __shfl(ib, 63);
    }
    for (int64_t t = lane; t < tiles; t += 64) b[2 * t + 1] += na;                    // ground after non-ground (same lane wrote it)
    if (lane == 0) {
        w.out_counts[f] = na + nb;
        w.out_flags[f] = a.fr[f].unchanged;
    }
}

template <typename T>
__global__ __launch_bounds__(PB) void k_wet_scatter(WetArgs w)
{
    const PreArgs &a = w.p;
    const int f = blockIdx.y;
    const int64_t base = a.frame_off[f], n = pre_rows(a, f);
    const int64_t tile0 = (int64_t)blockIdx.x * SG_TILE;
    if (tile0 >= n) return;
    const T *rows = (const T *)a.rows;
    const int unchanged = a.fr[f].unchanged;
    __shared__ int wc[4][4][2];
    const int tid = threadIdx.x, wv = tid >> 6;
    const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
    uint8_t c[4];
    int pre[4];
    for (int q = 0; q < 4; ++q) {
        const int64_t r = tile0 + q * PB + tid;
        c[q] = r < n ? w.cls[base + r] : 0;
        const unsigned long long ma = __ballot(c[q] == 1), mb = __ballot(c[q] == 2);
        pre[q] = c[q] == 1 ? __popcll(ma & lt) : __popcll(mb & lt);
        if ((tid & 63) == 0) { wc[q][wv][0] = __popcll(ma); wc[q][wv][1] = __popcll(mb); }
    }
    __syncthreads();
    const int32_t *tb = w.tile_base + ((int64_t)f * a.max_tiles + blockIdx.x) * 2;
    int run[2] = {tb[0], tb[1]};
    // every load of the thread's four rows before the first store (the argument struct carries no `restrict`: behind a store to out_rows the
    // compiler may not start the next row's loads, and the kernel was a chain of four round trips per thread: 0.59 ms per 256 sweeps)
    T sx[4], sy[4], sz[4], si[4], sl[4];
    double nw[4];
    int32_t sf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = base + tile0 + q * PB + tid;
        const T *s = rows + (c[q] ? r : base) * 5;
        sx[q] = s[0]; sy[q] = s[1]; sz[q] = s[2]; si[q] = s[3]; sl[q] = s[4];
        nw[q] = c[q] == 2 ? w.new_i[r] : 0.0;
        sf[q] = (c[q] && w.src_first) ? w.src_first[r] : (int32_t)(r - base);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (c[q]) {
            const int k = c[q] - 1;
            int off = run[k];
            for (int ww = 0; ww < wv; ++ww) off += wc[q][ww][k];
            const int64_t dst = base + off + pre[q];
            double *d = w.out_rows + dst * 5;
            d[0] = (double)sx[q]; d[1] = (double)sy[q]; d[2] = (double)sz[q];
            d[3] = (c[q] == 2) ? nw[q] : (double)si[q];                  // augmentation.py:151-153
            double lab = (double)sl[q];
            if (!unchanged) {
                if (w.replace) lab = 0.0;                                // :155-156
                if (c[q] == 2) lab = 1.0;                                // :159
            }
            d[4] = lab;
            w.out_src[dst] = sf[q];
        }
        for (int k = 0; k < 2; ++k) run[k] += wc[q][0][k] + wc[q][1][k] + wc[q][2][k] + wc[q][3][k];
    }
}

// Source rows of a chained result (snowfall, then wet ground): final row -> snowfall row -> input row.
__global__ __launch_bounds__(PB) void k_compose_src(const int64_t *__restrict__ frame_off, const int64_t *__restrict__ counts,
                                                   const int32_t *__restrict__ second, const int32_t *__restrict__ first,
                                                   int32_t *__restrict__ out)
{
    const int f = blockIdx.y;
    const int64_t base = frame_off[f], n = counts[f];
    for (int64_t i = (int64_t)blockIdx.x * PB + threadIdx.x; i < n; i += (int64_t)gridDim.x * PB)
        out[base + i] = first[base + second[base + i]];
}

// ================================================================================================================
// host side

enum { B_GDIST = 0, B_GNORM, B_GCOS, B_PART, B_HIST, B_ROWMIN, B_FRAME, B_CLS, B_NEWI, B_TCNT, B_TBASE, B_CDIST, B_LEAF, B_QPART, B_N };
static_assert(B_N <= 16, "SgPrepassScratch holds 16 buffers");

static int ensure(SgPrepassScratch *s, int i, size_t bytes)
{
    if (bytes <= s->cap[i]) return 0;
    if (s->buf[i]) (void)hipFree(s->buf[i]);
    s->buf[i] = nullptr; s->cap[i] = 0;
    const size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&s->buf[i], want) != hipSuccess) return -1;
    s->cap[i] = want;
    return 0;
}

extern "C" void sg_prepass_release(SgPrepassScratch *s)
{
    for (int i = 0; i < 16; ++i) { if (s->buf[i]) (void)hipFree(s->buf[i]); s->buf[i] = nullptr; s->cap[i] = 0; }
}

#define LCHK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

static int estimate(SgPrepassScratch *s, PreArgs &a, int dtype, int64_t n_total, int64_t max_frame, int min_ground,
                    int err_code, bool exact_f32_mean, hipStream_t st)
{
    const int64_t max_tiles = (max_frame + SG_TILE - 1) / SG_TILE > 0 ? (max_frame + SG_TILE - 1) / SG_TILE : 1;
    a.max_tiles = max_tiles;
    const size_t n = (size_t)(n_total > 0 ? n_total : 1), nf = (size_t)a.n_frames;
    if (ensure(s, B_GDIST, n * 8) || ensure(s, B_GNORM, n * 8) || ensure(s, B_GCOS, n * 8) ||
        ensure(s, B_PART, nf * (size_t)max_tiles * 12 * 8) || ensure(s, B_HIST, nf * HX * HY * 4) ||
        ensure(s, B_ROWMIN, nf * HX * 8) || ensure(s, B_FRAME, nf * sizeof(PreFrame)) || (dtype == 0 && ensure(s, B_CDIST, n * 4)))
        return -1;
    a.g_dist = (double *)s->buf[B_GDIST]; a.g_norm = (double *)s->buf[B_GNORM]; a.g_ang = (double *)s->buf[B_GCOS];
    a.part = (double *)s->buf[B_PART]; a.hist = (int32_t *)s->buf[B_HIST]; a.rowmin = (double *)s->buf[B_ROWMIN];
    a.fr = (PreFrame *)s->buf[B_FRAME];
    a.cdist = (float *)s->buf[B_CDIST];
    hipError_t e = hipMemsetAsync(a.hist, 0, nf * HX * HY * 4, st);
    if (e != hipSuccess) return (int)e;
    dim3 grid((unsigned)max_tiles, (unsigned)a.n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_pre_ground<float>, grid, dim3(PB), 0, st, a);
    else hipLaunchKernelGGL(k_pre_ground<double>, grid, dim3(PB), 0, st, a);
    LCHK();
    hipLaunchKernelGGL(k_pre_means, dim3((unsigned)a.n_frames), dim3(64), 0, st, a, min_ground, err_code);
    LCHK();

    hipLaunchKernelGGL(k_pre_moments, grid, dim3(PB), 0, st, a);
    LCHK();
    hipLaunchKernelGGL(k_pre_rowmin, dim3(HX, (unsigned)a.n_frames), dim3(PB), 0, st, a);
    LCHK();
    hipLaunchKernelGGL(k_pre_lines, dim3((unsigned)a.n_frames), dim3(64), 0, st, a, (dtype == 0 && !a.rows_as_f64) ? 1 : 0);
    LCHK();
    if (a.lines_override) {
        hipLaunchKernelGGL(k_pre_override_lines, dim3((unsigned)((a.n_frames + 63) / 64)), dim3(64), 0, st, a);
        LCHK();
    }
    if (dtype == 0 && exact_f32_mean) {
        // only frames whose noise line fell back to p = linregress(range, I / cos) need the float32 mean; the two
        // kernels below leave at once for every other frame
        const int max_leaves = (int)(max_frame / 64 + 8);      // pairwise leaves hold 65..128 values
        if (ensure(s, B_LEAF, nf * 3 * (size_t)max_leaves * 4)) return -1;
        hipLaunchKernelGGL(k_pre_gather, grid, dim3(PB), 0, st, a);
        LCHK();
        hipLaunchKernelGGL(k_pre_mean32, dim3((unsigned)a.n_frames), dim3(PB), 0, st, a, (int *)s->buf[B_LEAF], max_leaves, (double *)nullptr);
        LCHK();
    }
    return 0;
}

// The snowfall prepass without per-row scratch: two passes over the rows (statistics; histogram), the rest per frame.
// (the tile partials' buffer, for a caller whose own row-streaming kernel fills it: sg_launch_sort with statistics)
extern "C" double *sg_prepass_reserve_tiles(SgPrepassScratch *s, int n_frames, int64_t max_frame)
{
    const int64_t max_tiles = (max_frame + SG_TILE - 1) / SG_TILE > 0 ? (max_frame + SG_TILE - 1) / SG_TILE : 1;
    if (ensure(s, B_PART, (size_t)n_frames * (size_t)max_tiles * LP_COLS * 8)) return nullptr;
    return (double *)s->buf[B_PART];
}

// The per-tile statistics as a kernel of their own, ahead of the prepass proper (the tiles are reserved: sg_prepass_reserve_tiles; the plane is
// known): on the prepass stream beside the sort and the scan instead of inside the sort's first pass.  sg_prepass_run is then told tiles_done = 1.
extern "C" int sg_prepass_stats_early(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t max_frame,
                                      const double *plane, void *stream)
{
    PreArgs a{};
    a.rows = rows; a.frame_off = frame_off; a.n_frames = n_frames; a.plane = plane; a.delta = 0.5; a.flat_earth = 0; a.cos_only = 1;
    const int64_t max_tiles = (max_frame + SG_TILE - 1) / SG_TILE > 0 ? (max_frame + SG_TILE - 1) / SG_TILE : 1;
    a.max_tiles = max_tiles;
    a.part = (double *)s->buf[B_PART];
    if (!a.part) return -1;
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_lean_stats<float>, grid, dim3(PB), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_lean_stats<double>, grid, dim3(PB), 0, (hipStream_t)stream, a);
    LCHK();
    return 0;
}

// The histogram of the snowfall prepass, cleared ahead of time: the fill depends on nothing of the batch, so it can run on the prepass
// stream beside the sort instead of in the prepass' own chain (sg_prepass_run is then told with hist_cleared = 1).
extern "C" int sg_prepass_clear_hist(SgPrepassScratch *s, int n_frames, void *stream)
{
    const size_t nf = (size_t)n_frames;
    if (ensure(s, B_HIST, nf * HX * HY * 4)) return -1;
    hipError_t e = hipMemsetAsync(s->buf[B_HIST], 0, nf * HX * HY * 4, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

static int lean_run(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off, int n_frames, int64_t n_total,
                    int64_t max_frame, const double *plane, double noise_floor, double *thr_poly, int32_t *status, hipStream_t st,
                    bool tiles_done, const void *srows, const int32_t *frame_unsorted, bool hist_cleared)
{
    PreArgs a{};
    a.rows = rows; a.srows = srows; a.frame_unsorted = srows ? frame_unsorted : nullptr; a.frame_off = frame_off; a.n_frames = n_frames; a.plane = plane; a.delta = 0.5; a.flat_earth = 0; a.cos_only = 1;
    a.noise_floor = noise_floor; a.power_factor = 15.0; a.status = status;
    const int64_t max_tiles = (max_frame + SG_TILE - 1) / SG_TILE > 0 ? (max_frame + SG_TILE - 1) / SG_TILE : 1;
    a.max_tiles = max_tiles;
    const size_t n = (size_t)(n_total > 0 ? n_total : 1), nf = (size_t)n_frames;
    if (ensure(s, B_PART, nf * (size_t)max_tiles * LP_COLS * 8) || ensure(s, B_HIST, nf * HX * HY * 4) ||
        ensure(s, B_ROWMIN, nf * HX * 8) || ensure(s, B_FRAME, nf * sizeof(PreFrame)) || (dtype == 0 && ensure(s, B_CDIST, n * 4)))
        return -1;
    a.part = (double *)s->buf[B_PART]; a.hist = (int32_t *)s->buf[B_HIST]; a.rowmin = (double *)s->buf[B_ROWMIN];
    a.fr = (PreFrame *)s->buf[B_FRAME]; a.cdist = (float *)s->buf[B_CDIST];
    if (!hist_cleared) {
        hipError_t e = hipMemsetAsync(a.hist, 0, nf * HX * HY * 4, st);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    if (!tiles_done) {                               // (else the channel sort's first kernel left the tile partials on its way over the rows)
        if (dtype == 0) hipLaunchKernelGGL(k_lean_stats<float>, grid, dim3(PB), 0, st, a);
        else hipLaunchKernelGGL(k_lean_stats<double>, grid, dim3(PB), 0, st, a);
        LCHK();
    }
    hipLaunchKernelGGL(k_lean_means, dim3((unsigned)n_frames), dim3(64), 0, st, a, 3, 7 /* SNOWGPU_E_GROUND */);
    LCHK();
    if (dtype == 0) hipLaunchKernelGGL(k_lean_hist<float>, grid, dim3(PB), 0, st, a);
    else hipLaunchKernelGGL(k_lean_hist<double>, grid, dim3(PB), 0, st, a);
    LCHK();
    if (n_frames <= 16) {
        hipLaunchKernelGGL(k_lean_rowmin_solve, dim3(HX, (unsigned)n_frames), dim3(PB), 0, st, a, dtype == 0 ? 1 : 0, thr_poly);
        LCHK();
    } else {
        hipLaunchKernelGGL(k_pre_rowmin, dim3(HX, (unsigned)n_frames), dim3(PB), 0, st, a);
        LCHK();
        hipLaunchKernelGGL(k_lean_lines_solve, dim3((unsigned)n_frames), dim3(64), 0, st, a, dtype == 0 ? 1 : 0, thr_poly);
        LCHK();
    }
    if (dtype == 0) {
        // only frames whose noise line fell back to p = linregress(range, I / cos) need NumPy's float32 mean of the ranges (and
        // their quadratic waits for it); the two kernels below leave at once for every other frame
        const int max_leaves = (int)(max_frame / 64 + 8);
        if (ensure(s, B_LEAF, nf * 3 * (size_t)max_leaves * 4)) return -1;
        hipLaunchKernelGGL(k_lean_gather<float>, grid, dim3(PB), 0, st, a);
        LCHK();
        hipLaunchKernelGGL(k_pre_mean32, dim3((unsigned)n_frames), dim3(PB), 0, st, a, (int *)s->buf[B_LEAF], max_leaves, thr_poly);
        LCHK();
    }
    return 0;
}

// Per-frame record of the lean prepass for a caller that finishes the fit itself (snowgpu_prepass_stats): SG_PRE_REC doubles.
__global__ void k_lean_export(PreArgs a, double *out)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.n_frames) return;
    const PreFrame &fr = a.fr[f];
    double *o = out + (int64_t)f * SG_PRE_REC;
    o[0] = fr.n_ground; o[1] = fr.xmean; o[2] = fr.xmean32; o[3] = fr.ymean; o[4] = fr.ymax; o[5] = fr.p0; o[6] = fr.p1;
    for (int k = 0; k < 11; ++k) o[7 + k] = fr.q[k];
}

// hist[hist == 0] = len(pointcloud_planes) (augmentation.py:234-235) where the histogram is made: the caller converts and selects only
__global__ __launch_bounds__(PB) void k_lean_fill_empty(PreArgs a)
{
    const int f = blockIdx.y;
    const int ng = (int)a.fr[f].n_ground;
    int32_t *h = a.hist + (int64_t)f * HX * HY;
    for (int i = blockIdx.x * PB + threadIdx.x; i < HX * HY; i += gridDim.x * PB)
        if (h[i] == 0) h[i] = ng;
}

__global__ void k_lean_force_mean32(PreArgs a)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < a.n_frames) a.fr[f].need_mean32 = 1;
}

// The lean prepass up to the histogram and the regression line p, for a caller that takes the per-row minima of the histogram
// with its own code (quirk Q8: np.argpartition): d_hist receives n_frames x 50 x 2555 int32, d_rec n_frames x SG_PRE_REC doubles
// (n_ground, mean range, NumPy's float32 mean of the ranges, mean I / cos, max I / cos, p slope, p intercept, the 11 sums of the
// quadratic fit in LQ_* order).
extern "C" int sg_prepass_stats_run(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off, int n_frames,
                                    int64_t n_total, int64_t max_frame, const double *plane, int32_t *d_hist, double *d_rec,
                                    int32_t *status, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    PreArgs a{};
    a.rows = rows; a.frame_off = frame_off; a.n_frames = n_frames; a.plane = plane; a.delta = 0.5; a.flat_earth = 0; a.cos_only = 1;
    a.noise_floor = 0.7; a.power_factor = 15.0; a.status = status;
    const int64_t max_tiles = (max_frame + SG_TILE - 1) / SG_TILE > 0 ? (max_frame + SG_TILE - 1) / SG_TILE : 1;
    a.max_tiles = max_tiles;
    const size_t n = (size_t)(n_total > 0 ? n_total : 1), nf = (size_t)n_frames;
    if (ensure(s, B_PART, nf * (size_t)max_tiles * LP_COLS * 8) || ensure(s, B_ROWMIN, nf * HX * 8) || ensure(s, B_FRAME, nf * sizeof(PreFrame)) ||
        (dtype == 0 && ensure(s, B_CDIST, n * 4)))
        return -1;
    a.part = (double *)s->buf[B_PART]; a.hist = d_hist; a.rowmin = (double *)s->buf[B_ROWMIN];
    a.fr = (PreFrame *)s->buf[B_FRAME]; a.cdist = (float *)s->buf[B_CDIST];
    hipError_t e = hipMemsetAsync(a.hist, 0, nf * HX * HY * 4, st);
    if (e != hipSuccess) return (int)e;
    dim3 grid((unsigned)max_tiles, (unsigned)n_frames);
    const unsigned fb = (unsigned)((n_frames + 63) / 64);
    if (dtype == 0) hipLaunchKernelGGL(k_lean_stats<float>, grid, dim3(PB), 0, st, a);
    else hipLaunchKernelGGL(k_lean_stats<double>, grid, dim3(PB), 0, st, a);
    LCHK();
    hipLaunchKernelGGL(k_lean_means, dim3((unsigned)n_frames), dim3(64), 0, st, a, 3, 7 /* SNOWGPU_E_GROUND */);
    LCHK();
    if (dtype == 0) hipLaunchKernelGGL(k_lean_hist<float>, grid, dim3(PB), 0, st, a);
    else hipLaunchKernelGGL(k_lean_hist<double>, grid, dim3(PB), 0, st, a);
    LCHK();
    hipLaunchKernelGGL(k_pre_rowmin, dim3(HX, (unsigned)n_frames), dim3(PB), 0, st, a);
    LCHK();
    hipLaunchKernelGGL(k_lean_lines, dim3(fb), dim3(64), 0, st, a, dtype == 0 ? 1 : 0);
    LCHK();
    if (dtype == 0) {        // NumPy's float32 mean of the ranges for EVERY frame: the caller's line may fall back to p (augmentation.py:250-251)
        const int max_leaves = (int)(max_frame / 64 + 8);
        if (ensure(s, B_LEAF, nf * 3 * (size_t)max_leaves * 4)) return -1;
        hipLaunchKernelGGL(k_lean_force_mean32, dim3(fb), dim3(64), 0, st, a);
        LCHK();
        hipLaunchKernelGGL(k_lean_gather<float>, grid, dim3(PB), 0, st, a);
        LCHK();
        hipLaunchKernelGGL(k_pre_mean32, dim3((unsigned)n_frames), dim3(PB), 0, st, a, (int *)s->buf[B_LEAF], max_leaves, (double *)nullptr);
        LCHK();
    }
    hipLaunchKernelGGL(k_lean_export, dim3(fb), dim3(64), 0, st, a, d_rec);
    LCHK();
    hipLaunchKernelGGL(k_lean_fill_empty, dim3(32, (unsigned)n_frames), dim3(PB), 0, st, a);     // (after the row minima: they read the raw counts)
    LCHK();
    return 0;
}

extern "C" int sg_prepass_run(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off, int n_frames,
                              int64_t n_total, int64_t max_frame, const double *plane, double noise_floor, double *thr_poly,
                              int32_t *status, void *stream, int tiles_done, const void *srows, const int32_t *frame_unsorted, int hist_cleared)
{
    return lean_run(s, rows, dtype, frame_off, n_frames, n_total, max_frame, plane, noise_floor, thr_poly, status, (hipStream_t)stream, tiles_done != 0,
                    srows, frame_unsorted, hist_cleared != 0);
}

extern "C" int sg_wet_run(SgPrepassScratch *s, const void *rows, int dtype, const int64_t *frame_off,
                          const int64_t *frame_cnt, int n_frames, int64_t n_total, int64_t max_frame, const double *plane, const SgWetParams *wp, double *out_rows, int32_t *out_src,
                          int64_t *out_counts, int32_t *out_flags, int32_t *status, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    WetArgs w{};
    PreArgs &a = w.p;
    a.lines_override = wp->lines;
    a.rows = rows; a.frame_off = frame_off; a.frame_cnt = frame_cnt; a.n_frames = n_frames; a.plane = plane; a.delta = wp->delta;
    a.flat_earth = wp->flat_earth; a.rows_as_f64 = 1; a.noise_floor = wp->noise_floor; a.power_factor = wp->power_factor; a.status = status;
    int rc = estimate(s, a, dtype, n_total, max_frame, 1000, 0, false, st);
    if (rc) return rc;
    const size_t n = (size_t)(n_total > 0 ? n_total : 1), nf = (size_t)n_frames;
    if (wp->estimation == 1) {                       // 'poly': the two quadratics replace the two lines
        if (ensure(s, B_QPART, nf * (size_t)a.max_tiles * PQ_COLS * 8)) return -1;
        a.qpart = (double *)s->buf[B_QPART]; a.seed = wp->seed;
        hipLaunchKernelGGL(k_pre_quad_part, dim3((unsigned)a.max_tiles, (unsigned)n_frames), dim3(PB), 0, st, a);
        LCHK();
        hipLaunchKernelGGL(k_pre_quad_fit, dim3((unsigned)n_frames), dim3(128), 0, st, a, 7 /* SNOWGPU_E_GROUND */);
        LCHK();
    }
    if (wp->fit_out) {
        hipLaunchKernelGGL(k_pre_export_fit, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0, st, a, wp->fit_out);
        LCHK();
    }
    if (ensure(s, B_CLS, n) || ensure(s, B_NEWI, n * 8) || ensure(s, B_TCNT, nf * (size_t)a.max_tiles * 2 * 4) ||
        ensure(s, B_TBASE, nf * (size_t)a.max_tiles * 2 * 4))
        return -1;
    w.water_height = wp->water_height; w.pavement_depth = wp->pavement_depth; w.replace = wp->replace;
    w.cls = (uint8_t *)s->buf[B_CLS]; w.new_i = (double *)s->buf[B_NEWI];
    w.tile_cnt = (int32_t *)s->buf[B_TCNT]; w.tile_base = (int32_t *)s->buf[B_TBASE];
    w.out_rows = out_rows; w.out_src = out_src; w.out_counts = out_counts; w.out_flags = out_flags; w.src_first = wp->src_first;
    dim3 grid((unsigned)a.max_tiles, (unsigned)n_frames);
    if (dtype == 0) hipLaunchKernelGGL(k_wet_apply<float>, grid, dim3(PB), 0, st, w);
    else hipLaunchKernelGGL(k_wet_apply<double>, grid, dim3(PB), 0, st, w);
    LCHK();
    hipLaunchKernelGGL(k_wet_scan, dim3((unsigned)n_frames), dim3(64), 0, st, w);
    LCHK();
    if (dtype == 0) hipLaunchKernelGGL(k_wet_scatter<float>, grid, dim3(PB), 0, st, w);
    else hipLaunchKernelGGL(k_wet_scatter<double>, grid, dim3(PB), 0, st, w);
    LCHK();
    return 0;
}

extern "C" int sg_launch_compose_src(const int64_t *frame_off, const int64_t *counts, int n_frames, int64_t max_frame,
                                     const int32_t *second, const int32_t *first, int32_t *out, void *stream)
{
    if (n_frames <= 0 || max_frame <= 0) return 0;
    const unsigned gx = (unsigned)std::min<int64_t>((max_frame + PB - 1) / PB, 64);
    hipLaunchKernelGGL(k_compose_src, dim3(gx, (unsigned)n_frames), dim3(PB), 0, (hipStream_t)stream, frame_off, counts, second, first, out);
    LCHK();
    return 0;
}
