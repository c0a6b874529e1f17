// sg_beam.h -- one LiDAR beam of the snowfall simulation, one thread per beam.
//
// Replaces, for one row of the point cloud, the Python loops
//   process_single_channel   simulation.py:50-194   (loop over beams :118, power loop :137-149)
//   get_occlusions           simulation.py:298-424  (loop over beams :338, O(K) NumPy per beam)
//   compute_occlusion_dict   simulation.py:231-295
//   geometry.*               geometry.py:14-223
//   received_power / xsi     simulation.py:547-569
//
// MI355X shape of the work
//   * the flake table is filed under 2048 azimuth bins (3.07 mrad, one beam width), each bin sorted
//     by range; a beam only visits the bins its wedge touches and stops at the first flake beyond
//     the hard target, so it tests ~20 candidates instead of ~18 000 -- the exact predicates of the
//     reference are then evaluated on that conservative superset, which leaves the result unchanged;
//   * the per-beam lists (near -> far intervals, later the scatterer list) live in LDS, strided by
//     the block size so that lane l always hits bank l whatever its private index is;
//   * everything that decides a label is float64 with no FMA contraction, and the float32 parts the
//     reference computes in the input dtype (range, beam azimuth, hard-target window) are float32.
#pragma once
#include "sg_math.h"
#include "sg_atan_cr.h"

// A pointer that was itself read from memory (the members of an SgTable) is a GENERIC pointer to the compiler: loads through it are
// flat_load, which count on the LDS counter as well as on the memory counter -- so every wait for a cross-lane read or an LDS list
// also waited for every record load in flight.  These tables live in device memory: say so.
#if defined(__HIP_DEVICE_COMPILE__)
#define SG_GLOBAL __attribute__((address_space(1)))
#else
#define SG_GLOBAL
#endif
template <typename P> __device__ __forceinline__ const SG_GLOBAL P *sg_gptr(const P *p) { return (const SG_GLOBAL P *)p; }

// Scans over the 64 lanes of a wave as DPP operations (row shifts inside a row of 16 lanes, then the last lane of a row broadcast to the
// rows after it): six VALU instructions and no trip through the LDS pipeline, where `__shfl_up` costs a cross-lane read, a wait, a compare
// and a select per step.  Every lane of the wave must be active.  The host harness runs waves of one lane: the identity.
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int sg_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int sg_wave_incl_add(int v)          // inclusive prefix sum
{
    v += sg_dpp<0x111, 0xf>(0, v); v += sg_dpp<0x112, 0xf>(0, v); v += sg_dpp<0x114, 0xf>(0, v); v += sg_dpp<0x118, 0xf>(0, v);   // row_shr:1, 2, 4, 8
    v += sg_dpp<0x142, 0xa>(0, v);                                   // row_bcast:15 into rows 1 and 3
    v += sg_dpp<0x143, 0xc>(0, v);                                   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int sg_wave_incl_max(int v)          // inclusive prefix maximum
{
    constexpr int I = (int)0x80000000;
    v = max(v, sg_dpp<0x111, 0xf>(I, v)); v = max(v, sg_dpp<0x112, 0xf>(I, v)); v = max(v, sg_dpp<0x114, 0xf>(I, v)); v = max(v, sg_dpp<0x118, 0xf>(I, v));
    v = max(v, sg_dpp<0x142, 0xa>(I, v));
    v = max(v, sg_dpp<0x143, 0xc>(I, v));
    return v;
}
__device__ __forceinline__ int sg_wave_last(int v) { return __builtin_amdgcn_readlane(v, 63); }   // lane 63's value, in a scalar register
#else
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int sg_dpp(int, int v) { return v; }
__device__ __forceinline__ int sg_wave_incl_add(int v) { return v; }
__device__ __forceinline__ int sg_wave_incl_max(int v) { return v; }
__device__ __forceinline__ int sg_wave_last(int v) { return v; }
#endif

// one step of the segmented fold of a pair loop (k_power_few, sg_wave_eval): take (sum, bin) of the lane the DPP control names if it belongs to the same beam
// and holds the larger sum (equal sums: the smaller bin)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void sg_fold_step(double &sm, int &kk, int oo)
{
    const int o2 = sg_dpp<CTRL, ROW_MASK>(oo, oo), k2 = sg_dpp<CTRL, ROW_MASK>(kk, kk);
    const int lo = sg_dpp<CTRL, ROW_MASK>(__double2loint(sm), __double2loint(sm)), hi = sg_dpp<CTRL, ROW_MASK>(__double2hiint(sm), __double2hiint(sm));
    const double s2 = __hiloint2double(hi, lo);
    if (o2 == oo && (s2 > sm || (s2 == sm && k2 < kk))) { sm = s2; kk = k2; }
}

// the whole fold: a segmented prefix maximum -- DPP row shifts, then the last lane of a row to the rows after it.  A lane whose source does
// not exist reads its own values back, which changes nothing; the last lane of a run of equal `oo` ends up with the run's first maximum
// whatever the tree, the runs being contiguous.  Every lane of the wave must be active.
__device__ __forceinline__ void sg_fold_runs(double &sm, int &kk, int oo)
{
    sg_fold_step<0x111, 0xf>(sm, kk, oo); sg_fold_step<0x112, 0xf>(sm, kk, oo); sg_fold_step<0x114, 0xf>(sm, kk, oo); sg_fold_step<0x118, 0xf>(sm, kk, oo);
    sg_fold_step<0x142, 0xa>(sm, kk, oo); sg_fold_step<0x143, 0xc>(sm, kk, oo);
}

// Owner of pair p = base + lane when the lanes' beams hold [excl, incl) of the wave's pair numbers: the first lane whose inclusive count
// exceeds p.  Every beam with pairs leaves its lane number at its first pair's place in the window (s_mark: one int per lane of the
// block), a beam that began before the window leads it, and a prefix maximum carries the marks forward: two LDS writes, one read and
// six DPP steps instead of a binary search of six dependent cross-lane reads.  Lanes past the wave's last pair get its last owner.
__device__ __forceinline__ int sg_pair_owner(int *s_mark, int tid, int base, int excl, int incl)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = tid & 63, wbase = tid & ~63;
    s_mark[tid] = -1;
    asm volatile("" ::: "memory");
    if (incl > excl && excl >= base && excl < base + 64) s_mark[wbase + (excl - base)] = lane;
    asm volatile("" ::: "memory");
    int m = ((volatile int *)s_mark)[tid];
    const unsigned long long cm = __ballot(excl < base && base < incl);
    if (lane == 0 && m < 0) m = cm ? __ffsll((long long)cm) - 1 : 0;
    return sg_wave_incl_max(m) & 63;
#else
    return 0;
#endif
}

// (beam, record) pairs a wave of sg_wave_scan takes per trip: its 64 lanes.  The host harness runs the scan as a wave of one lane
// (tests/host_harness/wave_vs_lane.cpp) and sets 1.
#ifndef SG_PAIR_WINDOW
#define SG_PAIR_WINDOW 64
#endif

template <typename T> struct SgReal;
template <> struct SgReal<float> { static constexpr bool is_f32 = true; };
template <> struct SgReal<double> { static constexpr bool is_f32 = false; };

struct SgBeamOut {
    int overflow;     // more flakes than the list holds: n_hits is still exact, nothing else is valid
    int range_error;  // window beyond the 1230-bin grid (reference: IndexError)
    int has_power;    // >= 1 flake kept: the received-power phase has work for this beam
    int n_flakes;     // scatterers before the hard target (the target sits at index n_flakes)
    int n_hits;       // flakes intersecting the beam (counted on after the list is full)
    int label;        // 0 / 1 / 2                                   (set by sg_beam_decide)
    int new_i;        // output intensity, labels 1 / 2
    int k_best;       // argmax bin of the power profile
    double diff2;     // 2 * (0.9 * max_intensity - new_i) for label 1, else 0
};

__device__ __forceinline__ int sg_bin_of(double theta, double inv_w, int nb)
{
    // theta may be slightly outside [0, 2 pi] after adding a margin
    if (theta < 0) theta += SG_TWO_PI;
    if (theta >= SG_TWO_PI) theta -= SG_TWO_PI;
    int b = (int)floor(theta * inv_w);
    if (b < 0) b = 0;
    if (b >= nb) b = nb - 1;
    return b;
}

// List views: element j of this thread's private array lives at base[j * stride + tid] -- LDS, strided by the block
// size (STRIDE > 0, a compile-time constant), or global memory strided by the lane count of the global-list tier
// (STRIDE == 0: run-time stride `rstride`).
#ifndef SG_DICT_SWEEP
#define SG_DICT_SWEEP 16     /* list capacities up to this one build the dict from sorted endpoints (0: the walk everywhere).  Same box, ms per
                                256 sweeps, 0 / 4 / 8: C2 3.99 / 3.95 / 3.92, C2far 8.37 / 8.36 / 8.24, C1 7.78 / 7.77 / 7.65; 8 / 16 with the 16-entry
                                kernel held to two waves per SIMD (SG_KP_WAVES_TIERS = 2: 18 registers spilled): C2 3.96 / 3.94, C2far 8.27 / 7.66,
                                C1 7.71 / 7.49, C4 18.23 / 18.15 -- at one wave per SIMD (no spills) C2far loses instead: 8.71 */
#endif
#include "sg_sortnet.h"
#define SG_IDX(j) ((STRIDE) ? ((j) * (STRIDE) + tid) : (int)((long long)(j) * rstride + tid))
#define SG_A1(j) s_a1[SG_IDX(j)]
#define SG_A2(j) s_a2[SG_IDX(j)]
#define SG_RHO(j) s_rho[SG_IDX(j)]
#define SG_RATIO(j) s_ratio[SG_IDX(j)]

// Bin window of a scatterer, packed beside a work-list entry in one 8-byte list cell: high word = k0 | k1 << 16 (both below
// 2^16: SG_RBINS = 1230), low word = a group start of the stage-A work list -- so that the work list can share the window
// column (k_power keeps three list columns in LDS instead of four: one more block per CU for the long-list tiers).
__device__ __forceinline__ int sg_kp_k0(double pk) { return __double2hiint(pk) & 0xffff; }
__device__ __forceinline__ int sg_kp_k1(double pk) { return (int)((unsigned)__double2hiint(pk) >> 16); }
__device__ __forceinline__ double sg_kp_make(int k0, int k1) { return __hiloint2double(k0 | (k1 << 16), 0); }
__device__ __forceinline__ void sg_work_put(double *cell, int g) { reinterpret_cast<int *>(cell)[0] = g; }   // low word only (little endian)
__device__ __forceinline__ int sg_work_get(const double *cell) { return reinterpret_cast<const int *>(cell)[0]; }

// Phases 1-2 and 3a for one beam (per lane).  Leaves the scatterer list in this lane's list column:
// s_a1[t] amplitude, s_a2[t] packed (k1, k0), s_rho[t] range, t = 0 .. n_flakes (hard target last).
template <typename T, int LMAX, int STRIDE> __device__ __forceinline__ void sg_beam_amp(T d_t, int S, int channel, const SgLasers *__restrict__ las, double *s_a1, double *s_a2, double *s_rho, double *s_ratio, int tid, SgBeamOut &out, int rstride = 0);

// Beam limits of a beam from its azimuth (simulation.py:96-101): theta_r = right, theta_l = left, each in [0, 2 pi].
__device__ __forceinline__ void sg_beam_limits(double theta_c, double beam_div_deg, double &theta_r, double &theta_l)
{
    const double half = (beam_div_deg / 2) * (SG_PI / 180.0);   // np.radians(beam_divergence / 2)
    theta_r = theta_c - half;                                   // :96
    theta_l = theta_c + half;                                   // :97
    if (theta_r < 0) theta_r = theta_r + SG_TWO_PI;             // :100
    if (theta_l < 0) theta_l = theta_l + SG_TWO_PI;
    if (theta_r > SG_TWO_PI) theta_r = theta_r - SG_TWO_PI;     // :101
    if (theta_l > SG_TWO_PI) theta_l = theta_l - SG_TWO_PI;
}

// Unoccluded ratio of a beam that met no flake (debug tap): compute_occlusion_dict's single slot over the whole wedge
// (simulation.py:260-293 with no intervals) -- (left - right) / beam divergence, not exactly 1 in floating point.
__device__ __forceinline__ double sg_clear_beam_ratio(double theta_c, double beam_div_deg)
{
    double ra, la;
    sg_beam_limits(theta_c, beam_div_deg, ra, la);
    if (ra > la) ra = ra - SG_TWO_PI;                           // :260-263
    const double e_min = ra < la ? ra : la, e_max = ra < la ? la : ra;
    const double w = e_min < e_max ? 0.0 + (-0.0 + (e_max - e_min)) : 0.0;      // np.sum of one addend
    return sg_clip01(w / (beam_div_deg * (SG_PI / 180.0)));
}

// ---- beam geometry (simulation.py:89-101; geometry.py:94-106, :133) ------------------------------------------------
struct SgBeamGeo {
    double d;                 // range, widened from the row dtype
    double theta_c;           // azimuth in [0, 2 pi]
    double theta_r, theta_l;  // beam limits
    double sr, cr, sl, cl;    // direction (sin, cos) of the two limit rays, to ~5e-16: the fast form of the distance test
    bool wrap;                // the wedge crosses the 0 / 2 pi seam (simulation.py:361)
    bool exact;               // every distance test by the reference's expression with the math library's tan (snowgpu_set_exact_math)
};

// the part of the geometry that follows from the azimuth alone (recomputed, not stored, where a beam's parameters travel)
__device__ __forceinline__ void sg_geo_limits(SgBeamGeo &g, double beam_div_deg)
{
    sg_beam_limits(g.theta_c, beam_div_deg, g.theta_r, g.theta_l);
    g.wrap = g.theta_r > g.theta_l;
}

template <typename T>
__device__ __forceinline__ SgBeamGeo sg_beam_geometry(T px, T py, T pz, double beam_div_deg, bool EXACT_TAN, T &d_t)
{
    SgBeamGeo g;
    if constexpr (SgReal<T>::is_f32) {
        d_t = sqrtf((px * px + py * py) + pz * pz);             // :89 np.linalg.norm in float32
        float tc = sg_atan2f(py, px);                           // :91
        if (tc < 0) tc = tc + (float)SG_TWO_PI;                 // :92 float32 add
        g.theta_c = (double)tc;
    } else {
        d_t = sqrt((px * px + py * py) + pz * pz);
        g.theta_c = sg_atan2_cr(py, px);                        // :91 in float64: rounded correctly (sg_atan_cr.h)
        if (g.theta_c < 0) g.theta_c = g.theta_c + SG_TWO_PI;
    }
    g.d = (double)d_t;
    sg_geo_limits(g, beam_div_deg);
    // directions of the limit rays theta_c -+ half: one sincos of the azimuth, rotated by the half divergence (a launch constant)
    double sc, cc, sh, ch;
    sg_sincos_0_2pi(g.theta_c, sc, cc);
    const double h = (beam_div_deg / 2) * (SG_PI / 180.0);
    if (h < 0.015625) {
        const double q = h * h;
        sh = h * (1.0 - q * (1.0 / 6) * (1.0 - q * (1.0 / 20) * (1.0 - q * (1.0 / 42))));
        ch = 1.0 - q * 0.5 * (1.0 - q * (1.0 / 12) * (1.0 - q * (1.0 / 30)));
    } else {
        sg_sincos_0_2pi(h, sh, ch);
    }
    g.sr = sc * ch - cc * sh; g.cr = cc * ch + sc * sh;
    g.sl = sc * ch + cc * sh; g.cl = cc * ch - sc * sh;
    g.exact = EXACT_TAN;
    return g;
}

// geometry.py:94-106 (angles_to_lines: a = -tan(theta), b = 1; vertical only at exactly pi/2, 3 pi/2) and :131-135
// (|a x + b y + 0| / sqrt(a^2 + b^2) < r), literally -- the reference's decision for one flake and one limit ray.
__device__ __forceinline__ bool sg_near_ray_reference(double theta, double fx, double fy, double fr, bool libm_tan)
{
    double a, b;
    if (theta == SG_PI / 2 || theta == 3 * SG_PI / 2) { a = 1.0; b = 0.0; }
    else { a = -(libm_tan ? tan(theta) : sg_tan_0_2pi(theta)); b = 1.0; }
    const double den = sqrt(a * a + b * b);                     // geometry.py:133
    const double num = fabs((fx * a + fy * b) + 0.0);
    return (num / den) < fr;
}

// The same decision without tangent, root and quotient wherever that is safe: the distance of the flake centre from the
// ray's line is |y cos(theta) - x sin(theta)|.  That value and the reference's quotient are both within
// ~1e-15 (|x| + |y|) of the true distance (rounding of theta_c -+ half, of the slope and of the products; the slope's
// own error cancels where it is large: |a| >> 1 means num / den = |x + y / a|), so outside a band a thousand times
// wider both give the same answer; inside the band (roughly one test in 1e9) and for NaNs the reference's expression decides.
// DEFER: the caller cannot afford the reference's expression where it stands (the pass over all rows: held to 96, now 80
// registers -- tangent, root and quotient inside its loop spilled, measured +5 % on the whole step); an undecided test is
// reported in `undecided` and the beam is redone by a kernel that can (sg_wave_scan).
template <bool DEFER = false>
__device__ __forceinline__ bool sg_near_ray(double theta, double s, double c, double fx, double fy, double fr, bool exact, bool &undecided)
{
    const double dist = fabs(fy * c - fx * s);
    const double band = 1e-12 * (fabs(fx) + fabs(fy));
    const bool clear = fabs(dist - fr) > band;                  // false for NaNs
    if constexpr (DEFER) {
        undecided = undecided || !clear;
        return dist < fr;
    } else {
        if (!exact && clear) return dist < fr;
        return sg_near_ray_reference(theta, fx, fy, fr, exact);
    }
}

// The reference's exact predicates for one flake against one beam (simulation.py:359-389; geometry.py:113-135, :193-223):
// does the disk intersect the wedge, and if so its interval angles (geometry.py:14-29: a limit ray that cuts the disk
// replaces the tangent angle on its side).
// The decision alone needs the FIRST HALF of a record -- azimuth, centre, radius: 32 of its 64 bytes (sg_common.h: SgEntry) -- and the pass
// over all rows reads the second half (range, bin flag, tangent angles) only of the records that intersect, about half of those tested
// (scripts/probe/bin_width_probe.cpp: 1.4 pairs per beam of a bench sweep, 49 % of them intersect):
// every lane of a record load is a cache access of its own (the L1's address unit is busy 70 % of that pass's cycles:
// profiles/r05_probe_l1_path.txt).  Measured: a quarter fewer accesses in the pair loop, the same time -- kept as the smaller load.
template <bool DEFER = false>
__device__ __forceinline__ bool sg_flake_test(const SgBeamGeo &g, double phi, double fx, double fy, double fr, bool &hit_r, bool &hit_l, bool &undecided)
{
    const bool centre = (g.theta_r <= phi && phi <= g.theta_l)                        // :359
                     || (g.wrap && g.theta_r - SG_TWO_PI <= phi && phi <= g.theta_l)  // :360
                     || (g.wrap && g.theta_r <= phi && phi <= g.theta_l + SG_TWO_PI); // :362
    const bool near_r = sg_near_ray<DEFER>(g.theta_r, g.sr, g.cr, fx, fy, fr, g.exact, undecided);      // geometry.py:131-135
    const bool near_l = sg_near_ray<DEFER>(g.theta_l, g.sl, g.cl, fx, fy, fr, g.exact, undecided);
    hit_r = near_r && sg_forward(g.theta_r, phi);                       // :379-384
    hit_l = near_l && sg_forward(g.theta_l, phi);                       // :379-385
    return centre || hit_r || hit_l;                                    // :389
}
template <bool DEFER = false>
__device__ __forceinline__ bool sg_flake_hits(const SgBeamGeo &g, const SgEntry &f, double &na1, double &na2, bool &undecided)
{
    bool hit_r, hit_l;
    const bool hit = sg_flake_test<DEFER>(g, f.phi, f.x, f.y, f.r, hit_r, hit_l, undecided);
    na1 = hit_r ? g.theta_r : f.t0;                                     // geometry.py:26
    na2 = hit_l ? g.theta_l : f.t1;                                     // geometry.py:27
    return hit;
}
__device__ __forceinline__ bool sg_flake_hits(const SgBeamGeo &g, const SgEntry &f, double &na1, double &na2)
{
    bool undecided = false;
    return sg_flake_hits<false>(g, f, na1, na2, undecided);
}

// ---- beam geometry + phase 1 (candidate scan) for one beam (per lane) -----------------------------------------------
// LMAX > 0: list capacity, a compile-time constant (lists in LDS).  LMAX == 0: the global-list tier, capacity `rcap`
// and stride `rstride` at run time.  The scan keeps counting after the list is full: out.n_hits is exact either way.
// Leaves the intersecting flakes, near -> far, in the list columns: s_a1[j], s_a2[j] interval angles (geometry.py:14-29),
// s_rho[j] range, j < min(n_hits, capacity); returns that length.  d_t / theta_c: the beam's range (row dtype) and azimuth.
// The table is filed under 2048 azimuth bins, each sorted by range: the beam visits only the bins its wedge touches and
// stops at the first flake beyond the hard target.
template <typename T, int LMAX, int STRIDE>
__device__ __forceinline__ int sg_beam_scan(T px, T py, T pz, const SgTable tab, double beam_div_deg, double *s_a1, double *s_a2,
                                            double *s_rho, int tid, SgBeamOut &out, T &d_t, double &theta_c,
                                            bool EXACT_TAN = false, int rstride = 0, int rcap = 0)
{
    constexpr bool HUGE_TIER = LMAX == 0;
    const int lcap = HUGE_TIER ? rcap : LMAX;
    out.overflow = 0; out.range_error = 0; out.diff2 = 0.0; out.has_power = 0; out.n_flakes = 0; out.n_hits = 0;
    out.label = 0; out.new_i = 0; out.k_best = 0;
    const SgBeamGeo g = sg_beam_geometry<T>(px, py, pz, beam_div_deg, EXACT_TAN, d_t);
    theta_c = g.theta_c;
    const double d = g.d;
    const int nb = (int)tab.n_bins;
    const int b_lo = sg_bin_of(g.theta_r - SG_BEAM_MARGIN, tab.inv_bin_w, nb);
    const int b_hi = sg_bin_of(g.theta_l + SG_BEAM_MARGIN, tab.inv_bin_w, nb);
    int span = b_hi - b_lo;
    if (span < 0) span += nb;
    int L = 0, hits = 0;
    int b = b_lo;
    for (int s = 0; s <= span; ++s) {
        const uint32_t e0 = tab.bin_start[b], e1 = tab.bin_start[b + 1];
        // software pipeline: the next record is requested before the current one is examined (the entry
        // array carries one spare record at its end, so e + 1 is always readable)
        SgEntry nxt = tab.entries[e0];
        for (uint32_t e = e0; e < e1; ++e) {
            const SgEntry f = nxt;
            nxt = tab.entries[e + 1];
            const double rho = f.rho;
            if (!(rho < d)) break;                              // :345 (bins are sorted by rho)
            if (s > 0 && !(f.flags & 1u)) continue;             // already met in an earlier bin
            double na1, na2;
            if (!sg_flake_hits(g, f, na1, na2)) continue;
            ++hits;
            if (L == lcap) continue;                            // list full: keep counting (the count picks the tier)
            int p = L;                                          // insertion sort by rho (:413-417)
            while (p > 0 && SG_RHO(p - 1) > rho) {
                SG_A1(p) = SG_A1(p - 1); SG_A2(p) = SG_A2(p - 1); SG_RHO(p) = SG_RHO(p - 1);
                --p;
            }
            SG_A1(p) = na1; SG_A2(p) = na2; SG_RHO(p) = rho;
            ++L;
        }
        if (++b == nb) b = 0;
    }
    out.n_hits = hits;
    if (hits > lcap) out.overflow = 1;
    return L;
}

// The last 16 bytes of a record -- range, bin flag, source row -- as ONE load issued where sg_tail_load stands and pinned where
// sg_tail_pin stands (a plain member read that only a branch uses is moved into that branch by the compiler, behind the wait for
// what decides the branch: a second round trip).
struct SgTail { double rho; uint32_t flags; };
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned int SgTailRaw __attribute__((ext_vector_type(4)));
__device__ __forceinline__ SgTailRaw sg_tail_load(const SG_GLOBAL SgEntry *fp) { return *reinterpret_cast<const SG_GLOBAL SgTailRaw *>(&fp->rho); }
__device__ __forceinline__ SgTail sg_tail_pin(SgTailRaw v)
{
    asm volatile("" : "+v"(v));
    SgTail t; t.rho = __hiloint2double((int)v.y, (int)v.x); t.flags = v.z;
    return t;
}
#else
typedef SgTail SgTailRaw;
__device__ __forceinline__ SgTailRaw sg_tail_load(const SgEntry *fp) { SgTail t; t.rho = fp->rho; t.flags = fp->flags; return t; }
__device__ __forceinline__ SgTail sg_tail_pin(SgTailRaw v) { return v; }
#endif

// COMPACT lists (sg_wave_scan): one word per intersecting flake instead of its two interval angles -- the record's index | bit 30: a limit
// ray cuts the disk on the right | bit 31: on the left (geometry.py:26-27: that ray's angle then replaces the tangent angle)
__device__ __forceinline__ uint32_t sg_hit_word(uint32_t e, bool hit_r, bool hit_l)
{
    return e | (hit_r ? 0x40000000u : 0u) | (hit_l ? 0x80000000u : 0u);
}
__device__ __forceinline__ void sg_hit_angles(uint32_t w, const SgEntry *__restrict__ entries, double theta_r, double theta_l, double &a1, double &a2)
{
    const SG_GLOBAL SgEntry *f = sg_gptr(entries) + (w & 0x3fffffffu);
    a1 = (w & 0x40000000u) ? theta_r : f->t0;                   // geometry.py:26
    a2 = (w & 0x80000000u) ? theta_l : f->t1;                   // geometry.py:27
}

// The same for a whole short list with ONE round of loads: the tangent angles of every listed record (16 bytes each, one load) are
// requested before the first is used, whether the word's bits will want them or not -- with the loads under those bits the compiler
// waited for each entry's pair before it asked for the next (four round trips for a four-entry list).  w[j], j >= L: any word (entry 0 is read).
template <int N>
__device__ __forceinline__ void sg_hit_angles_all(const uint32_t (&w)[N], int L, const SgEntry *__restrict__ entries, double theta_r, double theta_l,
                                                  double (&a1)[N], double (&a2)[N])
{
#if defined(__HIP_DEVICE_COMPILE__)
    SgTailRaw raw[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const SG_GLOBAL SgEntry *f = sg_gptr(entries) + (j < L ? (w[j] & 0x3fffffffu) : 0u);
        raw[j] = *reinterpret_cast<const SG_GLOBAL SgTailRaw *>(&f->t0);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        a1[j] = (w[j] & 0x40000000u) ? theta_r : __hiloint2double((int)raw[j].y, (int)raw[j].x);      // geometry.py:26
        a2[j] = (w[j] & 0x80000000u) ? theta_l : __hiloint2double((int)raw[j].w, (int)raw[j].z);      // geometry.py:27
    }
#else
    for (int j = 0; j < N; ++j) if (j < L) sg_hit_angles(w[j], entries, theta_r, theta_l, a1[j], a2[j]);
#endif
}

// ---- the same scan, one WAVE for its 64 beams ---------------------------------------------------------------------------
// A beam tests every record of its bins that is nearer than its target -- a handful for a ground return, dozens for a far
// wall -- and with one beam per lane a wave is as slow as its farthest beam (SQ counters: 45 % of the lanes active).  Here the
// wave flattens the work: every lane finds how many records each of its (first two) bins holds before its target (binary
// search: bins are sorted by range), a prefix sum over the wave numbers the (beam, record) pairs, and the lanes take 64
// pairs at a time whichever beams they belong to -- the owner's geometry travels by cross-lane reads.  A hit is appended to
// its owner's list through an LDS counter; every beam sorts its few entries by (range, scan order) afterwards, which is the
// order the per-lane scan produces.  Bins beyond the second (wedges wider than a bin) keep the per-lane loop.
// s_cnt: one int per lane of the block; s_key: LMAX ints per lane (the scan order of the stored entries); s_st: two ints per lane; s_mark: one
// int per lane (sg_pair_owner; the host harness passes none).
// DEFER (the pass over all rows in the default arithmetic): a beam one of whose distance tests falls inside the band of
// sg_near_ray is not decided here: SG_HITS_UNDECIDED is set in its flake count (with `overflow`), nothing is kept of its list, and the
// caller sends it to the global-list tier, whose scan carries the reference's expression.  About one test in 1e8 on ordinary input.
// COMPACT (the pass over all rows): the list keeps, per flake, its range (the sort key) and ONE word instead of the two interval angles --
// the record's index in the table | bit 30: the left angle is the beam's right limit | bit 31: the right angle is the beam's left limit
// (geometry.py:26-27; otherwise they are the record's tangent angles t0 / t1) -- in the column s_a1 points to, read as uint32_t; s_a2 is
// not used.  The caller resolves the word when it hands the list on (sg_hit_angles).  19 KB of LDS per 256 beams instead of 31.
template <typename T, int LMAX, int STRIDE, bool DEFER = false, bool COMPACT = false>
__device__ __forceinline__ int sg_wave_scan(bool act, T px, T py, T pz, const SgTable tab, double beam_div_deg, double *s_a1, double *s_a2,
                                            double *s_rho, int *s_cnt, int *s_key, int *s_st, int tid, SgBeamOut &out, T &d_t, double &theta_c,
                                            bool EXACT_TAN, double *ov_blk = nullptr, int ov_cap = 0, int *s_mark = nullptr)
{
    // ov_blk: overflow slot of the block's column 0 (the slots follow the columns), or null.  Flakes LMAX .. ov_cap - 1 of a beam go
    // to its slot as they are met (the caller adds the first LMAX and the header if the beam ends up within ov_cap).
    out.overflow = 0; out.range_error = 0; out.diff2 = 0.0; out.has_power = 0; out.n_flakes = 0; out.n_hits = 0;
    out.label = 0; out.new_i = 0; out.k_best = 0;
    const int lane = tid & 63, wbase = tid & ~63;
    SgBeamGeo g{};
    uint32_t st0 = 0, st2 = 0;
    int n0 = 0, n1 = 0, span = -1, b_lo = 0, nb = 1;
    d_t = 0; theta_c = 0.0;
    if (act) {
        g = sg_beam_geometry<T>(px, py, pz, beam_div_deg, EXACT_TAN, d_t);
        theta_c = g.theta_c;
        nb = (int)tab.n_bins;
        b_lo = sg_bin_of(g.theta_r - SG_BEAM_MARGIN, tab.inv_bin_w, nb);
        const int b_hi = sg_bin_of(g.theta_l + SG_BEAM_MARGIN, tab.inv_bin_w, nb);
        span = b_hi - b_lo;
        if (span < 0) span += nb;
        const int b_nx = (b_lo + 1 == nb) ? 0 : b_lo + 1;
        const SG_GLOBAL uint32_t *g_start = sg_gptr(tab.bin_start);
        const SG_GLOBAL SgEntry *g_ent = sg_gptr(tab.entries);
        // Bin starts and the coarse range index: EIGHT independent loads, every one issued before the first is used -- no load
        // under a condition of its own (the compiler kept such a load behind a wait for the earlier ones: bin starts, then the
        // second bin's end, then the index -- three round trips where the addresses need none of the loaded values).
        const uint32_t s0 = g_start[b_lo], s0e = g_start[b_lo + 1], s1 = g_start[b_nx], s1e = g_start[b_nx + 1];
        const bool has_q = tab.bin_q != nullptr;
        uint32_t c0 = 0, c1 = 0, u0 = 0, u1 = 0;
        int kk = 0;
        if (has_q) {       // the coarse range index brackets the prefix (counts below the multiples of SG_QSTEP_M around d): the search
            // below then looks at the one or two records in between instead of halving the whole bin
            const double dq = g.d * (1.0 / SG_QSTEP_M);
            kk = dq < (double)(SG_QSTEPS - 1) ? (int)dq : SG_QSTEPS - 1;
            const int k1 = kk < SG_QSTEPS - 1 ? kk + 1 : kk;      // (the last step has no upper count: its load repeats the lower one)
            const SG_GLOBAL uint32_t *q0 = sg_gptr(tab.bin_q) + (size_t)b_lo * SG_QSTEPS, *q1 = sg_gptr(tab.bin_q) + (size_t)b_nx * SG_QSTEPS;
            c0 = q0[kk]; u0 = q0[k1]; c1 = q1[kk]; u1 = q1[k1];
        }
        st0 = s0; st2 = s1;
        uint32_t hi0 = s0e, hi1 = span >= 1 ? s1e : s1;
        uint32_t lo0 = st0, lo1 = st2;                          // records with rho < d: a prefix of each (sorted) bin
        if (has_q) {
            lo0 = st0 + c0;
            if (kk < SG_QSTEPS - 1) hi0 = st0 + u0;
            if (span >= 1) { lo1 = st2 + c1; if (kk < SG_QSTEPS - 1) hi1 = st2 + u1; }
            if (!(g.d == g.d)) { lo0 = hi0 = st0; lo1 = hi1 = st2; }      // NaN target: no record is nearer
        }
        while (lo0 < hi0 || lo1 < hi1) {
            const uint32_t m0 = (lo0 + hi0) >> 1, m1 = (lo1 + hi1) >> 1;
            const double r0 = lo0 < hi0 ? g_ent[m0].rho : 0.0, r1 = lo1 < hi1 ? g_ent[m1].rho : 0.0;
            if (lo0 < hi0) { if (r0 < g.d) lo0 = m0 + 1; else hi0 = m0; }
            if (lo1 < hi1) { if (r1 < g.d) lo1 = m1 + 1; else hi1 = m1; }
        }
        n0 = (int)(lo0 - st0); n1 = (int)(lo1 - st2);
    }
    s_cnt[tid] = 0;                                             // same wave writes and bumps it: LDS operations of a wave keep their order
    // first records of the beam's two bins: read by whichever lane tests one of its records -- through LDS, not a cross-lane
    // register read, so that they need no register during the loop (the kernel sits at the edge of a fifth wave per SIMD)
    s_st[2 * tid] = (int)st0; s_st[2 * tid + 1] = (int)st2;
    asm volatile("" ::: "memory");
    const int cnt = n0 + n1;
    const int incl = sg_wave_incl_add(cnt);
    const int excl = incl - cnt;
    const int total = sg_wave_last(incl);
    const unsigned long long ent_bits = (unsigned long long)tab.entries;
    for (int base = 0; base < total; base += SG_PAIR_WINDOW) {
        const int p = base + lane;
        const bool valid = p < total;
#if defined(SG_SCAN_OWNER_MARKS)
        const int o = sg_pair_owner(s_mark, tid, base, excl, incl);    // first lane whose inclusive count exceeds p
#else
        // (sg_pair_owner here -- marks and a prefix maximum instead of the search -- was measured: the pass 6 % slower; the search's cross-lane
        // reads overlap the record loads of the trip before, the marks' LDS writes and 24 B of spilled registers did not pay for them)
        int lo = 0, hi = 63;                                    // owner = first lane whose inclusive count exceeds p
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            const int v = __shfl(incl, mid);
            if (v > p) hi = mid; else lo = mid + 1;
        }
        const int o = lo & 63;
#endif
        const int j = p - __shfl(excl, o);
        const int n0o = __shfl(n0, o);
        const uint32_t st0o = (uint32_t)s_st[2 * (wbase + o)], st2o = (uint32_t)s_st[2 * (wbase + o) + 1];
        const uint32_t e = j < n0o ? st0o + (uint32_t)j : st2o + (uint32_t)(j - n0o);
        SgBeamGeo og;
        og.d = __shfl(g.d, o); og.theta_c = __shfl(g.theta_c, o);
        og.sr = __shfl(g.sr, o); og.cr = __shfl(g.cr, o); og.sl = __shfl(g.sl, o); og.cl = __shfl(g.cl, o);
        og.exact = EXACT_TAN;
        const SgEntry *oent = (const SgEntry *)(((unsigned long long)__shfl((unsigned)(ent_bits >> 32), o) << 32) | (unsigned long long)__shfl((unsigned)ent_bits, o));
        if (valid) {
            sg_geo_limits(og, beam_div_deg);
            const SG_GLOBAL SgEntry *fp = sg_gptr(oent) + e;
            const double phi = fp->phi, fx = fp->x, fy = fp->y, fr = fp->r;         // the record's first half
            // ... and its last 16 bytes (range, bin flag) in the SAME round of loads: read only by the half that intersect, they cost a
            // second round trip per trip of this loop when they wait for the test's outcome (sg_tail_load / sg_tail_pin)
            const SgTailRaw tl_raw = sg_tail_load(fp);
            bool und = false, hit_r, hit_l;
            const bool hit = sg_flake_test<DEFER>(og, phi, fx, fy, fr, hit_r, hit_l, und);
            const SgTail tl = sg_tail_pin(tl_raw);
            if (hit) {
                const double rho = tl.rho;
                const uint32_t flags = tl.flags;
                if (!(j >= n0o && !(flags & 1u))) {                                 // a flake filed under both bins counts once
                    const int col = wbase + o;
                    const int pos = und ? LMAX + ov_cap : atomicAdd(&s_cnt[col], 1);
                    if (pos < LMAX) {
                        if constexpr (COMPACT) reinterpret_cast<uint32_t *>(s_a1)[pos * STRIDE + col] = sg_hit_word(e, hit_r, hit_l);
                        else { s_a1[pos * STRIDE + col] = hit_r ? og.theta_r : fp->t0; s_a2[pos * STRIDE + col] = hit_l ? og.theta_l : fp->t1; }
                        s_rho[pos * STRIDE + col] = rho;
                        s_key[pos * STRIDE + col] = p;
                    } else if (pos < ov_cap) {
                        double *sp = ov_blk + (size_t)col * SG_OV_STRIDE + 2 + 3 * pos;
                        sp[0] = hit_r ? og.theta_r : fp->t0; sp[1] = hit_l ? og.theta_l : fp->t1; sp[2] = rho;
                    }
                }
            }
            // (an undecided test of a record that would not have counted -- filed under an earlier bin too -- sends the beam round as well: harmless)
            if (DEFER && und) atomicOr(&s_cnt[wbase + o], SG_HITS_UNDECIDED);            // (later appends of this beam then land nowhere)
        }
    }
    int hits = ((volatile int *)s_cnt)[tid];                    // bumped by other lanes of this wave
    bool undecided = DEFER && (hits & SG_HITS_UNDECIDED);
    if (undecided) hits &= ~SG_HITS_UNDECIDED;
    int L = hits < LMAX ? hits : LMAX;
    if (act && span >= 2) {                                     // wedges wider than a bin: the further bins, per lane
        int b = b_lo + 2 >= nb ? b_lo + 2 - nb : b_lo + 2;
        int key = 0x40000000;
        for (int s = 2; s <= span; ++s) {
            const uint32_t e0 = tab.bin_start[b], e1 = tab.bin_start[b + 1];
            for (uint32_t e = e0; e < e1; ++e) {
                const SgEntry f = tab.entries[e];
                if (!(f.rho < g.d)) break;
                ++key;
                if (!(f.flags & 1u)) continue;
                double na1, na2;
                if (!sg_flake_hits<DEFER>(g, f, na1, na2, undecided)) continue;
                ++hits;
                if (L == LMAX) {
                    if (hits <= ov_cap) {
                        double *sp = ov_blk + (size_t)tid * SG_OV_STRIDE + 2 + 3 * (hits - 1);
                        sp[0] = na1; sp[1] = na2; sp[2] = f.rho;
                    }
                    continue;
                }
                if constexpr (COMPACT) reinterpret_cast<uint32_t *>(s_a1)[L * STRIDE + tid] = sg_hit_word(e, na1 != f.t0, na2 != f.t1);   // (an angle equal to the tangent angle bit for bit may as well be read from the record)
                else { s_a1[L * STRIDE + tid] = na1; s_a2[L * STRIDE + tid] = na2; }
                s_rho[L * STRIDE + tid] = f.rho; s_key[L * STRIDE + tid] = key;
                ++L;
            }
            if (++b == nb) b = 0;
        }
    }
    // order by (range, scan order): insertion sort of <= LMAX entries (simulation.py:413-417)
    if constexpr (COMPACT) {
        uint32_t *s_rec = reinterpret_cast<uint32_t *>(s_a1);
        for (int i = 1; i < L; ++i) {
            const double r = s_rho[i * STRIDE + tid];
            const uint32_t w = s_rec[i * STRIDE + tid];
            const int k = s_key[i * STRIDE + tid];
            int q = i;
            while (q > 0 && (s_rho[(q - 1) * STRIDE + tid] > r || (s_rho[(q - 1) * STRIDE + tid] == r && s_key[(q - 1) * STRIDE + tid] > k))) {
                s_rho[q * STRIDE + tid] = s_rho[(q - 1) * STRIDE + tid]; s_rec[q * STRIDE + tid] = s_rec[(q - 1) * STRIDE + tid];
                s_key[q * STRIDE + tid] = s_key[(q - 1) * STRIDE + tid];
                --q;
            }
            s_rho[q * STRIDE + tid] = r; s_rec[q * STRIDE + tid] = w; s_key[q * STRIDE + tid] = k;
        }
    } else
    for (int i = 1; i < L; ++i) {
        const double r = s_rho[i * STRIDE + tid], x1 = s_a1[i * STRIDE + tid], x2 = s_a2[i * STRIDE + tid];
        const int k = s_key[i * STRIDE + tid];
        int q = i;
        while (q > 0 && (s_rho[(q - 1) * STRIDE + tid] > r || (s_rho[(q - 1) * STRIDE + tid] == r && s_key[(q - 1) * STRIDE + tid] > k))) {
            s_rho[q * STRIDE + tid] = s_rho[(q - 1) * STRIDE + tid]; s_a1[q * STRIDE + tid] = s_a1[(q - 1) * STRIDE + tid];
            s_a2[q * STRIDE + tid] = s_a2[(q - 1) * STRIDE + tid]; s_key[q * STRIDE + tid] = s_key[(q - 1) * STRIDE + tid];
            --q;
        }
        s_rho[q * STRIDE + tid] = r; s_a1[q * STRIDE + tid] = x1; s_a2[q * STRIDE + tid] = x2; s_key[q * STRIDE + tid] = k;
    }
    out.n_hits = hits;
    if (hits > LMAX) out.overflow = 1;
    if (undecided) { out.n_hits = hits | SG_HITS_UNDECIDED; out.overflow = 1; }
    return L;
}

// ---- phase 2: compute_occlusion_dict (simulation.py:252-295) for one beam (per lane) ------------------------------
// In: the L intersecting flakes of sg_beam_scan in the list columns, the beam's azimuth and range.  Out: the occlusion dict
// in s_rho[t], s_ratio[t], t = 0 .. S (scatterers near -> far, the hard target last at range d); returns S.
// The reference sorts the unique endpoints, gives every elementary slot to the nearest flake covering it and sums the slot
// widths per owner.  Owner of the slot starting at endpoint e is the first (nearest) j with a1_j <= e < a2_j, so each
// owner's sum can be produced by walking the endpoints inside its own interval -- no sorted endpoint array, no assignment
// array.
// KEEP_RHO = false (k_power, list capacities up to 16): the flakes' ranges are NOT in the list columns -- s_rho is never touched;
// instead *srcmap receives, 4 bits per dict entry t, the list index j the entry came from, and the caller fetches the ranges
// itself afterwards (one list column less in LDS).  The debug tap is then the caller's too.
template <int LMAX, int STRIDE, bool KEEP_RHO = true>
__device__ __forceinline__ int sg_beam_dict(int L, double theta_c, double d, double beam_div_deg, double *s_a1, double *s_a2,
                                            double *s_rho, double *s_ratio, int tid, int dbg_cap, int32_t *dbg_count,
                                            double *dbg_rj, double *dbg_ratio, int rstride = 0, unsigned long long *srcmap = nullptr)
{
    constexpr bool HUGE_TIER = LMAX == 0;
    static_assert(KEEP_RHO || (LMAX > 0 && LMAX <= 16), "the source map holds 16 four-bit entries");
    unsigned long long smap = 0;
    // One pass over the list entries (a1_q, a2_q).  Short lists (capacity <= 8) are walked fully unrolled with every load
    // issued up front -- the walk is a chain of LDS round trips otherwise, one per entry -- the longer ones four at a time.
    auto for_entries = [&](auto &&body) {
        if constexpr (!HUGE_TIER && LMAX <= 8) {
#pragma unroll
            for (int q = 0; q < LMAX; ++q) { const double q1 = SG_A1(q), q2 = SG_A2(q); if (q < L) body(q, q1, q2); }
        } else {
#pragma unroll 4
            for (int q = 0; q < L; ++q) { const double q1 = SG_A1(q), q2 = SG_A2(q); body(q, q1, q2); }
        }
    };
    double theta_r, theta_l;
    sg_beam_limits(theta_c, beam_div_deg, theta_r, theta_l);
    double ra = theta_r, la = theta_l;
    if (ra > la) {                                              // :260-263
        ra = ra - SG_TWO_PI;
        for (int j = 0; j < L; ++j) {
            const double a1 = SG_A1(j);
            if (a1 > SG_A2(j)) SG_A1(j) = a1 - SG_TWO_PI;
        }
    }
    const double delta = beam_div_deg * (SG_PI / 180.0);        // np.radians(beam_divergence), :289
    int S = 0;                                                  // scatterers kept (dict entries before -1)
    double e_min = ra < la ? ra : la, e_max = ra < la ? la : ra;
    for (int j = 0; j < L; ++j) {
        const double lo = SG_A1(j), hi = SG_A2(j);
        if (lo < e_min) e_min = lo;
        if (hi < e_min) e_min = hi;
        if (lo > e_max) e_max = lo;
        if (hi > e_max) e_max = hi;
    }
    SgNpSum acc;
    // the slots of owner j, walked inside its own interval: exact for any number of slots (NumPy's blocked sum)
    auto owner_walk = [&](int j, bool &made) -> double {
        const double lo = SG_A1(j), hi = SG_A2(j);
        made = false;
        acc.reset();
        double e = lo;
        while (e < hi) {                                        // slots i1 .. i2-1 (:277-282)
            bool pre = false;
            double nxt = hi;
            for_entries([&](int q, double q1, double q2) {
                if (q < j && q1 <= e && e < q2) pre = true;     // a nearer flake already owns it (:284)
                if (q1 > e && q1 < nxt) nxt = q1;
                if (q2 > e && q2 < nxt) nxt = q2;
            });
            if (ra > e && ra < nxt) nxt = ra;
            if (la > e && la < nxt) nxt = la;
            if (!pre) { acc.push(nxt - e); made = true; }       // :266 diffs, :285-286
            e = nxt;
        }
        return acc.result();
    };
    // One walk over all elementary slots, left to right: each goes to its owner's running sum (kept in the ratio
    // column; the slots of one owner arrive in the order of diffs[assignment == j]) or to the hard target.
    // A running sum is NumPy's sum for fewer than 8 addends; an owner with more is redone by owner_walk.
    double tgt_sum;
    // "has this owner got a slot yet, and how many": LMAX <= 16: 4 bits per owner in cnt, saturating at 8; LMAX <= 64: one
    // bit per owner; global-list tier: the ratio column itself, -1 = no slot yet (slot widths are positive)
    unsigned long long cnt = 0;
    if constexpr (HUGE_TIER) for (int j = 0; j < L; ++j) SG_RATIO(j) = -1.0;
    acc.reset();
#if SG_DICT_SWEEP
    if constexpr (!HUGE_TIER && LMAX <= SG_DICT_SWEEP) {
        // The same slots, left to right, without the walk's search for the next endpoint: the 2 LMAX + 2 endpoints (unused list entries
        // and nothing else stand in as e_max) sorted once by a fixed compare-exchange network in registers (sg_sortnet.h), then every
        // pair of neighbours that differ is a slot [e, nxt) -- np.unique's (:265) --, its owner the first list entry that covers it
        // (:284).  Every lane runs the same instructions whatever its list: the walk's trip count varied from lane to lane.
        constexpr int NE = 2 * LMAX + 2;
        // The owner test reads the intervals from registers: all of them at up to 8 entries (32 registers), eight at a time beyond that --
        // a first round gives every slot one of the first eight entries covers to the nearest of them and marks it, the second round looks
        // among the other eight for the slots left and sends what nobody covers to the hard target.  Each owner's slots, and the hard
        // target's, still arrive in slot order: an owner belongs to one round.
        constexpr int HALF = LMAX <= 8 ? LMAX : 8;
        static_assert(LMAX <= 8 || LMAX == 16, "two rounds of eight entries");
        double s[NE];
        double qa[HALF], qb[HALF];
#pragma unroll
        for (int q = 0; q < LMAX; ++q) {
            const double q1 = SG_A1(q), q2 = SG_A2(q);
            s[2 * q] = q < L ? q1 : e_max; s[2 * q + 1] = q < L ? q2 : e_max;
            if (q < HALF) { qa[q] = q < L ? q1 : INFINITY; qb[q] = q2; }              // (an unused entry covers nothing)
        }
        s[2 * LMAX] = ra; s[2 * LMAX + 1] = la;
        sg_sort_net<NE>(s);
        // one slot to its owner's running sum (the ratio column; 4 bits of `cnt` count the owner's slots, saturating at 8)
        auto credit = [&](int own, double w) {
            const unsigned c = (unsigned)(cnt >> (4 * own)) & 15u;
            SG_RATIO(own) = c ? SG_RATIO(own) + w : w;
            if (c < 8) cnt += 1ull << (4 * own);
        };
        [[maybe_unused]] unsigned long long claimed = 0ull;       // 16 entries: the slots the first round gave to one of the first eight
        if constexpr (LMAX > 8) {
#pragma unroll
            for (int i = 0; i + 1 < NE; ++i) {
                const double e = s[i], nxt = s[i + 1];
                if (!(e < nxt)) continue;
                int own = -1;
#pragma unroll
                for (int q = HALF - 1; q >= 0; --q) if (qa[q] <= e && e < qb[q]) own = q;
                if (own >= 0) { credit(own, nxt - e); claimed |= 1ull << i; }
            }
#pragma unroll
            for (int q = 0; q < HALF; ++q) {
                const double q1 = SG_A1(HALF + q), q2 = SG_A2(HALF + q);
                qa[q] = HALF + q < L ? q1 : INFINITY; qb[q] = q2;
            }
        }
#pragma unroll
        for (int i = 0; i + 1 < NE; ++i) {
            const double e = s[i], nxt = s[i + 1];
            if (!(e < nxt)) continue;                           // equal neighbours: one endpoint
            if constexpr (LMAX > 8) { if ((claimed >> i) & 1ull) continue; }
            int own = -1;
#pragma unroll
            for (int q = HALF - 1; q >= 0; --q) if (qa[q] <= e && e < qb[q]) own = (LMAX > 8 ? HALF : 0) + q;   // nearest flake covering the slot (:284)
            const double w = nxt - e;
            if (own < 0) acc.push(w);                           // nobody claimed it: hard target (:292-293)
            else credit(own, w);
        }
    } else
#endif
    {
        double e = e_min;
        while (e < e_max) {
            int own = -1;
            double nxt = e_max;
            for_entries([&](int q, double q1, double q2) {
                if (own < 0 && q1 <= e && e < q2) own = q;      // nearest flake covering the slot (:284)
                if (q1 > e && q1 < nxt) nxt = q1;
                if (q2 > e && q2 < nxt) nxt = q2;
            });
            if (ra > e && ra < nxt) nxt = ra;
            if (la > e && la < nxt) nxt = la;
            const double w = nxt - e;
            if (own < 0) acc.push(w);                           // nobody claimed it: hard target (:292-293)
            else if constexpr (HUGE_TIER) {
                const double have = SG_RATIO(own);
                SG_RATIO(own) = have < 0 ? w : have + w;
            } else if constexpr (LMAX <= 16) {
                const unsigned c = (unsigned)(cnt >> (4 * own)) & 15u;
                SG_RATIO(own) = c ? SG_RATIO(own) + w : w;
                if (c < 8) cnt += 1ull << (4 * own);
            } else {
                const bool first = !((cnt >> own) & 1ull);
                SG_RATIO(own) = first ? w : SG_RATIO(own) + w;
                cnt |= 1ull << own;
            }
            e = nxt;
        }
    }
    tgt_sum = acc.result();
    for (int j = 0; j < L; ++j) {
        bool redo;
        if constexpr (!HUGE_TIER && LMAX <= 16) {
            const unsigned c = (unsigned)(cnt >> (4 * j)) & 15u;
            if (c == 0) continue;                               // every slot already owned by nearer flakes
            redo = c >= 8;
        } else {
            if constexpr (HUGE_TIER) { if (SG_RATIO(j) < 0) continue; }
            else { if (!((cnt >> j) & 1ull)) continue; }
            // fewer than 7 endpoints strictly inside the interval -> fewer than 8 slots
            const double lo = SG_A1(j), hi = SG_A2(j);
            int inside = (ra > lo && ra < hi) + (la > lo && la < hi);
            for_entries([&](int, double q1, double q2) { inside += (q1 > lo && q1 < hi) + (q2 > lo && q2 < hi); });
            redo = inside >= 7;
        }
        double sum = 0.0 + SG_RATIO(j);
        if (redo) { bool made; sum = owner_walk(j, made); }
        if constexpr (KEEP_RHO) {
            const double rho = SG_RHO(j);
            SG_RHO(S) = rho;                                    // S <= j: in-place compaction
        } else smap |= (unsigned long long)j << (4 * S);
        SG_RATIO(S) = sg_clip01(sum / delta);                   // :288-290
        ++S;
    }
    if constexpr (KEEP_RHO) SG_RHO(S) = d;
    SG_RATIO(S) = sg_clip01(tgt_sum / delta);
    const int n_dict = S + 1;
    if constexpr (KEEP_RHO) {
        if (dbg_count) {
            *dbg_count = n_dict;
            for (int t = 0; t < n_dict && t < dbg_cap; ++t) { dbg_rj[t] = SG_RHO(t); dbg_ratio[t] = SG_RATIO(t); }
        }
    } else if (srcmap) *srcmap = smap;
    return S;
}

// Phases 1, 2 and 3a in place (the tiers that do not hand their beams over): scan, dict, amplitudes.
template <typename T, int LMAX, int STRIDE>
__device__ __forceinline__ void sg_beam(T px, T py, T pz, int channel, const SgTable tab,
                                        const SgLasers *__restrict__ las,
                                        double beam_div_deg, double *s_a1, double *s_a2, double *s_rho,
                                        double *s_ratio, int tid, SgBeamOut &out, int dbg_cap,
                                        int32_t *dbg_count, double *dbg_rj, double *dbg_ratio,
                                        bool EXACT_TAN = false, int rstride = 0, int rcap = 0)
{
    T d_t;
    double theta_c;
    const int L = sg_beam_scan<T, LMAX, STRIDE>(px, py, pz, tab, beam_div_deg, s_a1, s_a2, s_rho, tid, out, d_t, theta_c, EXACT_TAN,
                                                rstride, rcap);
    if (out.overflow) return;
    const int S = sg_beam_dict<LMAX, STRIDE>(L, theta_c, (double)d_t, beam_div_deg, s_a1, s_a2, s_rho, s_ratio, tid, dbg_cap, dbg_count,
                                             dbg_rj, dbg_ratio, rstride);
    if (S == 0) return;                                         // :133 no snowflake in this beam -> label 0
    sg_beam_amp<T, LMAX, STRIDE>(d_t, S, channel, las, s_a1, s_a2, s_rho, s_ratio, tid, out, rstride);
}

// ---- phase 3a (per lane): amplitude and bin window of every scatterer (simulation.py:137-146) --------
// In: s_rho[t], s_ratio[t] for t = 0 .. S (the hard target last, range d).  Out: s_a1[t] amplitude, s_a2[t] packed (k1, k0).
template <typename T, int LMAX, int STRIDE>
__device__ __forceinline__ void sg_beam_amp(T d_t, int S, int channel, const SgLasers *__restrict__ las, double *s_a1, double *s_a2,
                                            double *s_rho, double *s_ratio, int tid, SgBeamOut &out, int rstride)
{
    constexpr bool F32 = SgReal<T>::is_f32;
    const int n_dict = S + 1;
    const int ch = channel;
    const int max_i = las->max_i[ch];
    const double c_tau = 299792458.0 * 1e-8;                    // c * tau_h
    const double beta_0 = 1 * 1e-6 / SG_PI;                     // :108
    const double i_snow = 0.9 * max_i;                          // :140
    const double ca_p0 = i_snow / beta_0;                       // :141 (also used for the hard target, Q1)
    for (int t = 0; t < n_dict; ++t) {
        int k0, k1;
        double amp;
        const double ratio = SG_RATIO(t);
        if (F32 && t == S) {                                    // hard target keeps its float32 range
            const float r = (float)d_t;
            k0 = (int)ceilf(r * 10.0f);                         // :145
            float ee = r + (float)c_tau;                        // :146 (float32 under NEP 50)
            ee = ee * 10.0f;
            ee = floorf(ee) + 1.0f;
            k1 = (int)ee;
            const float r2 = r * r;                             // r_j ** 2 in float32 (:549)
            amp = (((ca_p0 * beta_0) * ratio) * sg_xsi(r)) / (double)r2;
        } else {
            const double r = SG_RHO(t);
            k0 = (int)ceil(r * 10);                             // :145
            k1 = (int)(floor((r + c_tau) * 10) + 1);            // :146
            amp = (((ca_p0 * beta_0) * ratio) * sg_xsi(r)) / (r * r);   // :549
        }
        if (k1 > SG_RBINS) { out.range_error = 1; k1 = SG_RBINS; }      // reference: IndexError (:149)
        if (k0 < 0) k0 = 0;
        SG_A1(t) = amp;
        SG_A2(t) = sg_kp_make(k0, k1);
    }
    out.n_flakes = S;
    out.has_power = 1;
}

// The same for k_power's three-column layout: in: s_rr[t] = ratio of dict entry t (t = 0 .. S, the hard target last),
// rho_of(t) = range of flake entry t < S (fetched by the caller: the hand-over queue); out: s_a1[t] amplitude, s_x[t] window
// (sg_kp_make; the low word is the stage-A work list's), s_rr[t] = range -- the ratio's cell, overwritten once it has been used.
template <typename T, int LMAX, int STRIDE, typename RhoOf>
__device__ __forceinline__ void sg_beam_amp3(T d_t, int S, int channel, const SgLasers *__restrict__ las, double *s_a1, double *s_x,
                                             double *s_rr, int tid, SgBeamOut &out, RhoOf rho_of)
{
    constexpr bool F32 = SgReal<T>::is_f32;
    constexpr int rstride = 0;
    const int max_i = las->max_i[channel];
    const double c_tau = 299792458.0 * 1e-8;                    // c * tau_h
    const double beta_0 = 1 * 1e-6 / SG_PI;                     // :108
    const double i_snow = 0.9 * max_i;                          // :140
    const double ca_p0 = i_snow / beta_0;                       // :141 (also used for the hard target, Q1)
    for (int t = 0; t <= S; ++t) {
        int k0, k1;
        double amp, r_out;
        const double ratio = s_rr[SG_IDX(t)];
        if (F32 && t == S) {                                    // hard target keeps its float32 range
            const float r = (float)d_t;
            k0 = (int)ceilf(r * 10.0f);                         // :145
            float ee = r + (float)c_tau;                        // :146 (float32 under NEP 50)
            ee = ee * 10.0f;
            ee = floorf(ee) + 1.0f;
            k1 = (int)ee;
            const float r2 = r * r;                             // r_j ** 2 in float32 (:549)
            amp = (((ca_p0 * beta_0) * ratio) * sg_xsi(r)) / (double)r2;
            r_out = (double)r;
        } else {
            const double r = t == S ? (double)d_t : rho_of(t);
            k0 = (int)ceil(r * 10);                             // :145
            k1 = (int)(floor((r + c_tau) * 10) + 1);            // :146
            amp = (((ca_p0 * beta_0) * ratio) * sg_xsi(r)) / (r * r);   // :549
            r_out = r;
        }
        if (k1 > SG_RBINS) { out.range_error = 1; k1 = SG_RBINS; }      // reference: IndexError (:149)
        if (k0 < 0) k0 = 0;
        s_a1[SG_IDX(t)] = amp;
        s_x[SG_IDX(t)] = sg_kp_make(k0, k1);
        s_rr[SG_IDX(t)] = r_out;
    }
    out.n_flakes = S;
    out.has_power = 1;
}

// sin(u) for u in [-0.3, 3.5]: one step of reduction against pi (hi + lo) and the odd Taylor polynomial to
// x^21 on [-pi/2, pi/2] (truncation 1.3e-18 relative, < 1 ULP overall).  The sign is dropped by the caller's
// square.  glibc's sin, which NumPy calls, is correctly rounded almost everywhere, so this agrees with it to
// the last bit or one next to it; see DESIGN.md "phase 3 arithmetic".
__device__ __forceinline__ double sg_sin_0_pi(double u)
{
    const double PI_HI = 3.141592653589793, PI_LO = 1.2246467991473532e-16;
    double x = u;
    if (u > 1.5707963267948966) x = (u - PI_HI) - PI_LO;
    const double x2 = x * x;
    double p = -1.9572941063391263e-20;                // -1/21!
    p = __builtin_fma(p, x2, 8.2206352466243295e-18);  //  1/19!
    p = __builtin_fma(p, x2, -2.8114572543455206e-15); // -1/17!
    p = __builtin_fma(p, x2, 7.6471637318198164e-13);  //  1/15!
    p = __builtin_fma(p, x2, -1.6059043836821613e-10); // -1/13!
    p = __builtin_fma(p, x2, 2.5052108385441720e-08);  //  1/11!
    p = __builtin_fma(p, x2, -2.7557319223985893e-06); // -1/9!
    p = __builtin_fma(p, x2, 1.9841269841269841e-04);  //  1/7!
    p = __builtin_fma(p, x2, -8.3333333333333332e-03); // -1/5!
    p = __builtin_fma(p, x2, 1.6666666666666666e-01);  //  1/3!  (sign applied below)
    return __builtin_fma(-(x * x2), p, x);
}

// A * sin^2(pi (R - r) / (c tau_h))  (simulation.py:549)
template <bool EXACT>
__device__ __forceinline__ double sg_power_term(double amp, double Rk, double r)
{
    const double c_tau = 299792458.0 * 1e-8;
    double sn;
    if (EXACT) sn = sin((SG_PI * (Rk - r)) / c_tau);
    else sn = sg_sin_0_pi((SG_PI * (Rk - r)) * (1.0 / c_tau));
    return amp * (sn * sn);
}

// R[k] of simulation.py:116 without a table: n = rint(k * step * 100) is the grid value in centimetres and
// n / 100 is recovered with one Newton step on n * 0.01 -- bit-identical to np.round(np.linspace(...), 2) for
// all 1230 bins (checked exhaustively on the host, tests/test_host_logic.py).
__device__ __forceinline__ double sg_range_bin(int k)
{
    const double step = (120 + 299792458.0 * 1e-8) / (SG_RBINS - 1);
    const double n = rint(((double)k * step) * 100.0);
    const double q = n * 0.01;
    return __builtin_fma(__builtin_fma(-q, 100.0, n), 0.01, q);
}

// ---- phase 3b (per lane): received power on the 10 cm grid and its first maximum (simulation.py:135-151) -----
// Each lane walks the bins of its own beam.  Same sums in the same order as the reference's `i[k] += ...` loop:
// flakes covering bin k in range order, then the hard target.
//   * NB consecutive bins are carried together -- NB independent sine chains per scatterer;
//   * EXACT PRUNING.  Every term is A_t * sin^2(.) <= A_t, so a bin's sum is bounded by U = sum of the amplitudes
//     of the scatterers whose windows reach it.  The window of the largest amplitude is evaluated first (its peak
//     sample is >= 0.997 A); after that a group of bins with U (1 + 1e-9) < best cannot hold the maximum, nor tie
//     with it, and is skipped without evaluating a single sine.  np.argmax's "first maximum" (:151) is kept by
//     preferring the smaller bin on equal sums, whatever the evaluation order.  On typical beams one or two of
//     the S + 1 windows survive.

// full sums of bins k .. k+NB-1 (flakes from index t_from on, then the hard target) folded into (best, k_best)
template <int STRIDE, bool EXACT, int NB>
__device__ __forceinline__ void sg_eval_group(int k, int t_from, int S, const double *__restrict__ rgrid, const double *s_a1,
                                              const double *s_a2, const double *s_rho, int tid, int tk0, int tk1, double tamp,
                                              double td, double &best, int &k_best, int rstride)
{
    double R[NB], sm[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int kk = k + i;
        R[i] = EXACT ? rgrid[kk < SG_RBINS ? kk : SG_RBINS - 1] : sg_range_bin(kk);
        sm[i] = 0.0;                                         // :135 np.zeros
    }
    for (int t = t_from; t < S; ++t) {
        const double pk = s_a2[SG_IDX(t)];
        const int q0 = sg_kp_k0(pk), q1 = sg_kp_k1(pk);
        if (q0 >= k + NB) break;                             // flake windows start in range order
        if (q1 <= k) continue;
        const double amp = s_a1[SG_IDX(t)], r = s_rho[SG_IDX(t)];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (k + i >= q0 && k + i < q1) sm[i] += sg_power_term<EXACT>(amp, R[i], r);   // :149
    }
    if (tk0 < k + NB && tk1 > k) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (k + i >= tk0 && k + i < tk1) sm[i] += sg_power_term<EXACT>(tamp, R[i], td);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (sm[i] > best || (sm[i] == best && k + i < k_best)) { best = sm[i]; k_best = k + i; }   // first maximum (:151)
}

// Two stages so that the lanes of a wave spend their time in the same code:
//   A  one step per scatterer: from bounds only, find the few bins of its window that can hold the maximum and put
//      them (as groups of NB bins) on a short per-lane work list in LDS (s_work, WCAP slots);
//   B  evaluate the listed groups exactly.
// Bounds -- exact prunings: a bin that is not listed can neither hold nor tie the maximum.
//   floor   the bin nearest to the peak of the strongest scatterer lies within 0.055 m of it (the grid step is 0.10
//           or 0.11 m), so its sum, hence the maximum, is >= A_max cos^2(pi 0.055 / (c tau_h)) = 0.99668 A_max.
//   zone    every term is A sin^2(.) <= A.  If a bin's sum reaches `need`, then for EACH scatterer t covering it
//           A_t sin^2(u_t) >= need - (sum of the amplitudes of the other scatterers whose windows overlap t's),
//           i.e. sin^2(u_t) >= q_t.  With cos^2(x) <= 1 - x^2 + x^4/3 that confines the bin to
//           |u_t - pi/2| <= sqrt(1.26 (1 - q_t)) for q_t >= 1/2: a zone of a few bins around the peak of t.
//           Applied to the strongest scatterer covering the bin (the others then being the weaker overlapping
//           windows only), every bin that matters lies in such a zone; q_t < 1/2 keeps the whole window.
// WCAP: slots of the per-lane work list (0: run-time capacity `rwcap`, the global-list tier)
// sg_power_plan: stage A; returns the number of listed groups still to be evaluated (a full list is evaluated on the
// spot, by this lane alone).  sg_lane_power: stages A and B by one lane.  sg_wave_eval: stage B by the whole wave.
template <int STRIDE, bool EXACT, int NB, int WCAP>
__device__ __forceinline__ int sg_power_plan(int S, const double *__restrict__ rgrid, const double *s_a1, const double *s_a2,
                                             const double *s_rho, double *s_work, int tid, double &best, int &k_best,
                                             int rstride = 0, int rwcap = 0)
{
    best = 0.0;
    k_best = 0;
    const int wcap = WCAP ? WCAP : rwcap;
    const double c_tau = 299792458.0 * 1e-8;
    const double step = (120 + c_tau) / (SG_RBINS - 1);
    const double tpk = s_a2[SG_IDX(S)];
    const int tk0 = sg_kp_k0(tpk), tk1 = sg_kp_k1(tpk);
    const double tamp = s_a1[SG_IDX(S)], td = s_rho[SG_IDX(S)];
    double amax = tamp;
    for (int t = 0; t < S; ++t) amax = fmax(amax, s_a1[SG_IDX(t)]);
    const double floor_ = 0.9966 * amax;
    int nw = 0;
    auto flush = [&]() {
        for (int w = 0; w < nw; ++w)
            sg_eval_group<STRIDE, EXACT, NB>(sg_work_get(&s_work[SG_IDX(w)]), 0, S, rgrid, s_a1, s_a2, s_rho, tid, tk0,
                                             tk1, tamp, td, best, k_best, rstride);
        nw = 0;
    };
    // ---- stage A ----
    for (int t = 0; t <= S; ++t) {                // the hard target is scatterer S
        const double A = s_a1[SG_IDX(t)];
        if (!(A > 0.0)) continue;                 // adds nothing anywhere; the bins it covers belong to others' zones
        const double pk = s_a2[SG_IDX(t)];
        const int k0 = sg_kp_k0(pk), k1 = sg_kp_k1(pk);
        // Each bin is the business of the strongest scatterer covering it (ties: the nearer one).  So this window
        // answers only for its bins outside stronger overlapping windows -- those trim it from the left (lo_trim) or
        // from the right (hi_trim); windows are in range order and of (almost) equal length -- and only the weaker
        // overlapping windows can add to its bins (oth).
        double oth = 0.0;
        int lo_trim = k0, hi_trim = k1;
        auto visit = [&](int j, int q0, int q1) {
            const double Aj = s_a1[SG_IDX(j)];
            if (Aj > A || (Aj == A && j < t)) {
                if (q0 <= k0) { if (q1 > lo_trim) lo_trim = q1; }
                else if (q1 >= k1) { if (q0 < hi_trim) hi_trim = q0; }
            } else oth += Aj;
        };
        for (int j = t - 1; j >= 0; --j) {
            const double pj = s_a2[SG_IDX(j)];
            if (sg_kp_k1(pj) <= k0) break;
            visit(j, sg_kp_k0(pj), sg_kp_k1(pj));
        }
        for (int j = t + 1; j <= S; ++j) {
            const double pj = s_a2[SG_IDX(j)];
            if (sg_kp_k0(pj) >= k1) break;
            visit(j, sg_kp_k0(pj), sg_kp_k1(pj));
        }
        const double need = fmax(best, floor_);
        if ((A + oth) * (1.0 + 1e-9) < need) continue;
        const double q = (need * (1.0 - 1e-9) - oth * (1.0 + 1e-9)) / A;
        int ka = k0, kb = k1 - 1;
        if (q >= 0.5) {
            const double om = q < 1.0 ? 1.0 - q : 0.0;
            const double delta = (double)sqrtf((float)(1.26 * om)) * (1.0 + 1e-6) + 1e-6;
            const double Rc = s_rho[SG_IDX(t)] + c_tau / 2;
            const double D = delta * (c_tau / SG_PI) + 0.006;      // + half a centimetre of grid rounding
            // bins whose grid value k * step (+- 0.005 of rounding, inside D) lies in [Rc - D, Rc + D]: one or two of them
            const int za = (int)ceil((Rc - D) * (1.0 / step)), zb = (int)floor((Rc + D) * (1.0 / step));
            if (za > ka) ka = za;
            if (zb < kb) kb = zb;
        }
        if (lo_trim > ka) ka = lo_trim;
        if (hi_trim - 1 < kb) kb = hi_trim - 1;
        for (int g = ka; g <= kb; g += NB) {
            if (nw == wcap) flush();
            sg_work_put(&s_work[SG_IDX(nw)], g);
            ++nw;
        }
    }
    return nw;
}

template <int STRIDE, bool EXACT, int NB, int WCAP>
__device__ __forceinline__ void sg_lane_power(int S, const double *__restrict__ rgrid, const double *s_a1, const double *s_a2,
                                              const double *s_rho, double *s_work, int tid, double &best, int &k_best,
                                              int rstride = 0, int rwcap = 0)
{
    const int nw = sg_power_plan<STRIDE, EXACT, NB, WCAP>(S, rgrid, s_a1, s_a2, s_rho, s_work, tid, best, k_best, rstride, rwcap);
    // ---- stage B ----
    const double tpk = s_a2[SG_IDX(S)];
    const int tk0 = sg_kp_k0(tpk), tk1 = sg_kp_k1(tpk);
    const double tamp = s_a1[SG_IDX(S)], td = s_rho[SG_IDX(S)];
    for (int w = 0; w < nw; ++w)
        sg_eval_group<STRIDE, EXACT, NB>(sg_work_get(&s_work[SG_IDX(w)]), 0, S, rgrid, s_a1, s_a2, s_rho, tid, tk0, tk1, tamp, td,
                                         best, k_best, rstride);
}

// Stage B by the whole wave (every lane of the wave calls it, nw = 0 for a lane without a beam).  The lanes list 0 .. WCAP
// groups each, and a group costs a dozen sines per scatterer that reaches it: lane by lane, most of a wave idles behind the
// lane with the longest list (SQ counters: 21 % of the lanes active in this stage, which is 60 % of k_power's instructions).
// Here the listed groups of all 64 beams are numbered by a prefix sum and taken 64 at a time, whichever beam they belong to
// -- the evaluating lane reads the owner's scatterer columns -- and every owner then folds the results of its groups, in
// any order (the first-maximum rule does not depend on it).  colbase: column of the wave's lane 0.
template <int STRIDE, bool EXACT, int NB>
__device__ __forceinline__ void sg_wave_eval(int nw, int S, const double *__restrict__ rgrid, const double *s_a1, const double *s_a2,
                                             const double *s_rho, const double *s_work, int colbase, double &best, int &k_best)
{
    static_assert(STRIDE > 0, "LDS lists only");
    const int lane = (int)(threadIdx.x & 63);
    const int incl = sg_wave_incl_add(nw);
    const int excl = incl - nw;
    const int total = sg_wave_last(incl);
    [[maybe_unused]] const int maxn = sg_wave_last(sg_wave_incl_max(nw));
    for (int base = 0; base < total; base += 64) {
        const int p = base + lane;
        const bool valid = p < total;
        int lo = 0, hi = 63;                                    // owner = first lane whose inclusive count exceeds p
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            const int v = __shfl(incl, mid);
            if (v > p) hi = mid; else lo = mid + 1;
        }
        const int o = lo & 63;
        const int j = p - __shfl(excl, o);
        const int So = __shfl(S, o);
        double gbest = -1.0;
        int gk = 0x7fffffff;
        if (valid) {
            const int tid = colbase + o;                        // the owner's column (SG_IDX)
            constexpr int rstride = 0;
            const int g = sg_work_get(&s_work[SG_IDX(j)]);
            const double tpk = s_a2[SG_IDX(So)];
            sg_eval_group<STRIDE, EXACT, NB>(g, 0, So, rgrid, s_a1, s_a2, s_rho, tid, sg_kp_k0(tpk), sg_kp_k1(tpk),
                                             s_a1[SG_IDX(So)], s_rho[SG_IDX(So)], gbest, gk, 0);
        }
#if !defined(SG_EVAL_FOLD_LOOP)
        // a beam's groups are neighbouring pairs: fold every run towards its last lane (sg_fold_runs), and the owner reads that one lane --
        // three cross-lane reads per trip instead of three per group of the wave's longest list
        sg_fold_runs(gbest, gk, valid ? o : 64 + lane);
        {
            const int qs = excl - base, qe = incl - 1 - base < 63 ? incl - 1 - base : 63;   // my run inside this window
            const double gb = __shfl(gbest, qe & 63);
            const int k = __shfl(gk, qe & 63);
            if (nw > 0 && qe >= 0 && qs < 64 && (gb > best || (gb == best && k < k_best))) { best = gb; k_best = k; }
        }
#else
        for (int r = 0; r < maxn; ++r) {                        // (cross-lane reads outside divergent code)
            const int q = excl + r - base;
            const double gb = __shfl(gbest, q & 63);
            const int k = __shfl(gk, q & 63);
            if (r < nw && q >= 0 && q < 64 && (gb > best || (gb == best && k < k_best))) { best = gb; k_best = k; }
        }
#endif
    }
}

// ---- phase 3c (per lane): focal term, clipping, attenuate-or-scatter decision (simulation.py:152-188) -------
// d = the beam's range as float64 (the value of the row dtype).  A scattered point (label 2) moves to
// d_max = k_best / 10 - c tau / 2 on its ray; the row is rebuilt from the record by sg_scatter_scale below.
__device__ __forceinline__ void sg_beam_decide(double d, int channel, const SgLasers *__restrict__ las, double best, int k_best,
                                               SgBeamOut &out)
{
    const int max_i = las->max_i[channel], min_i = las->min_i[channel];
    const double f_slope = las->focal_slope[channel], f_offset = las->focal_offset[channel];   // (the laser's four constants in one round of loads)
    const double c_tau = 299792458.0 * 1e-8;
    double i_max = best;                                        // :152
    const double d_max = ((double)k_best / 10) - (c_tau / 2);   // :153
    const double t1 = 1 - d_max / 120;
    const double i_snow = 0.9 * max_i;
    i_max += max_i * f_slope * fabs(f_offset - t1 * t1);        // :155
    if (i_max < min_i) i_max = min_i;                           // :156
    if (i_max > max_i) i_max = max_i;
    long long new_i = (long long)i_max;                         // :162 / :182
    if (fabs(d_max - d) < 2 * (1.0 / 10)) {                     // :158
        out.label = 1;                                          // :160
        out.diff2 = 2.0 * (i_snow - (double)new_i);             // :170 (Q2)
    } else {
        out.label = 2;                                          // :174
    }
    if (new_i < min_i) new_i = min_i;                           // :186
    if (new_i > max_i) new_i = max_i;
    out.new_i = (int)new_i;                                     // :188
    out.k_best = k_best;
}

// scale = d_max / d of a scattered point (simulation.py:153, :176); the moved coordinates are (T)((double)p * scale)
// (:178-180: a float32 column times a float64 scalar, stored back into the row dtype)
__device__ __forceinline__ double sg_scatter_scale(int k_best, double d)
{
    const double c_tau = 299792458.0 * 1e-8;
    const double d_max = ((double)k_best / 10) - (c_tau / 2);
    return d_max / d;
}

__device__ __forceinline__ uint32_t sg_pack_record(const SgBeamOut &o)
{
    return (uint32_t)(o.new_i & 255) | ((uint32_t)o.label << 8) | ((uint32_t)o.k_best << 12);
}
