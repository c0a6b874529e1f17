// snowgpu_rows.hip -- the later capacity tiers as ROW kernels: G lanes of a wave work on ONE beam.
//
// A beam of a later tier meets 5 .. 63 flakes (simulation.py:338-424 hands compute_occlusion_dict, :231-295, that many
// intervals).  One beam per lane -- rounds 1-3 -- keeps the interval list of every beam in LDS (a 16-entry tier: 35 KB per wave,
// ONE wave per SIMD), walks it entry by entry for each of the ~2 L elementary slots (O(L^2) dependent LDS round trips), and
// scans the beam's table records one after the other (a chain of L2 latencies): 14 000 VALU instructions and 5 us per beam,
// at half the lanes (profiles/r04a_*_pmc.txt).  Here a row of G = 8 / 16 / 64 lanes owns the beam and lane j owns list entry j:
//   scan    the row tests G table records per step (1 KB of contiguous 64-byte records), hits are numbered by a ballot and
//           land in lane order; a rank sort by (range, scan order) puts them near -> far (simulation.py:413-417);
//   dict    every elementary slot costs one ballot (owner = first covering lane, simulation.py:284) and one row-minimum (the next
//           endpoint) instead of a walk over the list; each lane keeps the running sum of the slots it owns, which IS NumPy's
//           sum for fewer than 8 addends -- a beam where a flake collects 8 or more is put on a (nearly always empty) redo list
//           and finished by the PAIRWISE instantiation of the same kernel, which streams NumPy's blocked pairwise sum (SgNpSum:
//           32 more registers, which the common case should not pay for); the hard target's slots, often more than 8, are summed
//           NumPy's way by eight lanes of the row in every case -- so the ratios stay bit-identical to
//           diffs[assignment == j].sum() (:289-293);
//   power   amplitudes lane-parallel (one division per scatterer, all at once), the pruning bounds of sg_power_plan with the
//           other scatterers read from a 2 KB LDS scratch, and the surviving (scatterer, bin) pairs of the whole wave
//           numbered by a prefix sum and evaluated one per lane, whichever beam they belong to.
// Nothing but that scratch lives in LDS, so the kernel runs at the occupancy its registers allow, and the tier needs neither
// hand-over buffers nor a second kernel: scan, dict and received power of a listed beam are one pass.
// Same arithmetic as sg_beam.h (sg_flake_hits, sg_power_term, sg_beam_decide, ...): same bits.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "sg_beam.h"
#include "sg_kutil.h"

#define RW_BLOCK 256
#ifndef RW_WAVES
#define RW_WAVES 4        /* waves per SIMD the row kernels are compiled for (<= 128 VGPRs) */
#endif

struct RwSlot { double a, b, c; int k; int pad; };     // 32 bytes per lane of LDS scratch

#define RW_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")   /* LDS traffic of ONE wave keeps its order; this keeps the compiler's */

template <int G>
struct Row {
    static constexpr int RPW = 64 / G;                    // beams per wave
    __device__ static __forceinline__ int lane() { return (int)(threadIdx.x & 63); }
    __device__ static __forceinline__ int lj() { return (int)(threadIdx.x & (G - 1)); }
    __device__ static __forceinline__ int base() { return (int)(threadIdx.x & 63) & ~(G - 1); }
    __device__ static __forceinline__ unsigned long long mask(bool p)
    {
        const unsigned long long m = __ballot(p);
        if constexpr (G == 64) return m;
        else return (m >> base()) & ((1ull << G) - 1ull);
    }
    __device__ static __forceinline__ unsigned long long below() { return (1ull << lj()) - 1ull; }
    __device__ static __forceinline__ double rmin(double v)
    {
#pragma unroll
        for (int o = 1; o < G; o <<= 1) { const double w = __shfl_xor(v, o); v = w < v ? w : v; }
        return v;
    }
    __device__ static __forceinline__ double rmax(double v)
    {
#pragma unroll
        for (int o = 1; o < G; o <<= 1) { const double w = __shfl_xor(v, o); v = w > v ? w : v; }
        return v;
    }
    __device__ static __forceinline__ int rmaxi(int v)
    {
#pragma unroll
        for (int o = 1; o < G; o <<= 1) { const int w = __shfl_xor(v, o); v = w > v ? w : v; }
        return v;
    }
};

// One walk over the elementary slots of a row's beam (simulation.py:266-293): lane j holds the interval [a1, a2) of flake j
// (+inf, +inf beyond the list).  Returns the lane's own sum and the hard target's sum (replicated over the row) and how many slots
// each collected.
//   own sums   PAIRWISE = false: running sums -- exact for fewer than 8 addends per owner (np.add.reduce adds them left to right);
//              true: NumPy's blocked pairwise sum for any number (SgNpSum: the rare repeat for an owner with 8 or more slots).
//   target     ALWAYS NumPy's sum, whatever the count -- the gaps between the flakes of a long list easily make 8 or more
//              unowned slots.  Lane-parallel: addend i of the target goes to lane i mod 8 of the row (its `pend`); when a block of
//              eight is complete the eight lanes commit it to their accumulators r (NumPy's eight interleaved partial sums); at the
//              end the accumulators are combined pairwise and the 0 .. 7 pending addends follow left to right -- operation for
//              operation what loops_utils.h.src's pairwise_sum does for n <= 128 (a beam has at most 2 * 63 + 2 endpoints).
template <int G, bool PAIRWISE>
__device__ __forceinline__ void rw_slot_walk(bool row_on, double a1, double a2, double ra, double la, double e_min, double e_max,
                                             double &own_sum, int &own_cnt, double &tgt_sum, int &tgt_cnt)
{
    using R = Row<G>;
    static_assert(G >= 8, "the target's pairwise sum uses eight lanes of the row");
    SgNpSum acc_o;
    if constexpr (PAIRWISE) acc_o.reset();
    own_sum = 0.0; own_cnt = 0; tgt_cnt = 0;
    double t_r = 0.0, t_pend = 0.0;                                   // lanes 0 .. 7 of the row: accumulator and pending addend
    int t_blocks = 0;
    double e = e_min;
    const int me = R::lj();
    while (__any(row_on && e < e_max)) {
        const bool go = row_on && e < e_max;                          // row-uniform
        const bool cover = go && a1 <= e && e < a2;                   // :284 (the nearest covering flake owns the slot)
        const unsigned long long cm = R::mask(cover);
        double nxt = e_max;
        if (a1 > e && a1 < nxt) nxt = a1;
        if (a2 > e && a2 < nxt) nxt = a2;
        if (ra > e && ra < nxt) nxt = ra;
        if (la > e && la < nxt) nxt = la;
        nxt = R::rmin(nxt);
        if (go) {
            const double w = nxt - e;                                 // :266 diffs
            const int own = cm ? __ffsll((long long)cm) - 1 : -1;
            if (own < 0) {                                            // nobody claimed it: hard target (:292-293)
                const int slot8 = tgt_cnt & 7;
                if (me == slot8) t_pend = w;
                if (slot8 == 7) {                                     // a block of eight is complete
                    if (me < 8) t_r = t_blocks ? t_r + t_pend : t_pend;
                    ++t_blocks;
                }
                ++tgt_cnt;
            } else if (own == me) {
                if constexpr (PAIRWISE) acc_o.push(w); else own_sum = own_cnt ? own_sum + w : w;
                ++own_cnt;
            }
            e = nxt;
        }
    }
    if constexpr (PAIRWISE) own_sum = acc_o.result(); else own_sum = 0.0 + own_sum;
    // the target: ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the pending addends oldest first, then 0.0 + .
    {
        const int rb = R::base();
        double r[8], p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { r[j] = __shfl(t_r, rb + j); p[j] = __shfl(t_pend, rb + j); }
        double res = t_blocks ? ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7])) : -0.0;
        const int c = tgt_cnt & 7;
#pragma unroll
        for (int j = 0; j < 7; ++j) if (j < c) res += p[j];
        tgt_sum = 0.0 + res;
    }
}

// The pruning bounds of sg_power_plan for ONE scatterer (amplitude A, window [k0, k1), range r, index t in dict order) against
// the others of its beam, read from the row's LDS scratch (S flakes) and the hard target (index S).  Returns the zone [ka, kb]
// of bins that can hold the maximum (empty: kb < ka).
template <int G>
__device__ __forceinline__ void rw_zone(const RwSlot *slots /* row's scratch */, int S, int t, double A, int k0, int k1, double r,
                                        double tamp, int tk0, int tk1, double floor_, int &ka, int &kb)
{
    const double c_tau = 299792458.0 * 1e-8;
    const double step = (120 + c_tau) / (SG_RBINS - 1);
    ka = 0; kb = -1;
    if (!(A > 0.0)) return;                                           // adds nothing anywhere
    double oth = 0.0;
    int lo_trim = k0, hi_trim = k1;
    auto visit = [&](int j, double Aj, int q0, int q1) {
        if (j == t) return;
        if (j < t ? !(q1 > k0) : !(q0 < k1)) return;                  // windows are in range order: no overlap
        if (Aj > A || (Aj == A && j < t)) {
            if (q0 <= k0) { if (q1 > lo_trim) lo_trim = q1; }
            else if (q1 >= k1) { if (q0 < hi_trim) hi_trim = q0; }
        } else oth += Aj;
    };
    for (int j = 0; j < S; ++j) {
        const double Aj = slots[j].a;
        const int kk = slots[j].k;
        visit(j, Aj, kk & 0xffff, kk >> 16);
    }
    visit(S, tamp, tk0, tk1);
    const double need = floor_;
    if ((A + oth) * (1.0 + 1e-9) < need) return;
    const double q = (need * (1.0 - 1e-9) - oth * (1.0 + 1e-9)) / A;
    ka = k0; kb = k1 - 1;
    if (q >= 0.5) {
        const double om = q < 1.0 ? 1.0 - q : 0.0;
        const double delta = (double)sqrtf((float)(1.26 * om)) * (1.0 + 1e-6) + 1e-6;
        const double Rc = r + c_tau / 2;
        const double D = delta * (c_tau / SG_PI) + 0.006;
        const int za = (int)ceil((Rc - D) * (1.0 / step)), zb = (int)floor((Rc + D) * (1.0 / step));
        if (za > ka) ka = za;
        if (zb < kb) kb = zb;
    }
    if (lo_trim > ka) ka = lo_trim;
    if (hi_trim - 1 < kb) kb = hi_trim - 1;
}

// Phase 1 for a row: candidate scan, G table records per step, then a rank sort by (range, scan order) (simulation.py:338-417).
// Every lane of the wave calls it; `act` is row-uniform.  Leaves flake j of the beam (near -> far) in lane j: (a1, a2) its interval
// angles (geometry.py:14-29), rho its range; returns min(hits, G), `nh` = the exact number of hits.
template <int G>
__device__ __forceinline__ int rw_scan_sort(bool act, const SgBeamGeo &geo, const SgTable &tab, RwSlot *row_slots, int &nh, double &a1, double &a2,
                                            double &rho)
{
    using R = Row<G>;
    const int lj = R::lj();
    nh = 0;                                                               // hits so far (row-uniform)
    {
        const int nb = act ? (int)tab.n_bins : 1;
        int b_lo = 0, span = -1;
        if (act) {
            b_lo = sg_bin_of(geo.theta_r - SG_BEAM_MARGIN, tab.inv_bin_w, nb);
            const int b_hi = sg_bin_of(geo.theta_l + SG_BEAM_MARGIN, tab.inv_bin_w, nb);
            span = b_hi - b_lo;
            if (span < 0) span += nb;
        }
        int b = b_lo;
        for (int s = 0; __any(act && s <= span); ++s) {
            const bool bin_on = act && s <= span;                         // row-uniform
            uint32_t e0 = 0, cnt = 0;
            if (bin_on) {                                                 // records of the bin nearer than the target: a prefix (bins are sorted by range)
                e0 = tab.bin_start[b];
                uint32_t lo = e0, hi = tab.bin_start[b + 1];
                if (tab.bin_q) {
                    const double dq = geo.d * (1.0 / SG_QSTEP_M);
                    const int kk = dq < (double)(SG_QSTEPS - 1) ? (int)dq : SG_QSTEPS - 1;
                    const uint32_t *qq = tab.bin_q + (size_t)b * SG_QSTEPS + kk;
                    lo = e0 + qq[0];
                    if (kk < SG_QSTEPS - 1) hi = e0 + qq[1];
                    if (!(geo.d == geo.d)) lo = hi = e0;
                }
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (tab.entries[mid].rho < geo.d) lo = mid + 1; else hi = mid;
                }
                cnt = lo - e0;
            }
            for (uint32_t p0 = 0; __any(bin_on && p0 < cnt); p0 += G) {
                const uint32_t p = p0 + (uint32_t)lj;
                bool hit = false;
                double na1 = 0.0, na2 = 0.0, r = 0.0;
                if (bin_on && p < cnt) {
                    const SgEntry fl = tab.entries[e0 + p];
                    r = fl.rho;
                    if (!(s > 0 && !(fl.flags & 1u))) hit = sg_flake_hits(geo, fl, na1, na2);   // a flake filed under several bins counts once
                }
                const unsigned long long hm = R::mask(hit);
                if (hit) {
                    const int pos = nh + (int)__popcll(hm & R::below());
                    if (pos < G) { RwSlot &sl = row_slots[pos]; sl.a = na1; sl.b = na2; sl.c = r; }
                }
                nh += (int)__popcll(hm);
            }
            if (bin_on && ++b == nb) b = 0;
        }
    }
    const int L = nh < G ? nh : G;
    // order by (range, scan order): rank sort through the scratch (simulation.py:413-417)
    RW_LDS_FENCE();
    a1 = INFINITY; a2 = INFINITY; rho = INFINITY;
    if (act && lj < L) { const RwSlot sl = row_slots[lj]; a1 = sl.a; a2 = sl.b; rho = sl.c; }
    int rank = 0;
    for (int i = 0; i < G; ++i) {
        if (!__any(act && i < L)) break;
        if (act && i < L && lj < L) {
            const double ri = row_slots[i].c;
            rank += (ri < rho) || (ri == rho && i < lj);
        }
    }
    RW_LDS_FENCE();
    if (act && lj < L) { RwSlot &sl = row_slots[rank]; sl.a = a1; sl.b = a2; sl.c = rho; }
    RW_LDS_FENCE();
    if (act && lj < L) { const RwSlot sl = row_slots[lj]; a1 = sl.a; a2 = sl.b; rho = sl.c; }
    RW_LDS_FENCE();
    return L;
}

template <typename T, int G, bool EXACT, bool PAIRWISE>
__global__ __launch_bounds__(RW_BLOCK, PAIRWISE ? 2 : RW_WAVES) void k_rows(SgBeamArgs a)
{
    using R = Row<G>;
    constexpr int RPW = R::RPW;
    __shared__ RwSlot s_slot[RW_BLOCK];
    __shared__ double s_res[RW_BLOCK];
    __shared__ int s_resk[RW_BLOCK];
    const int tid = threadIdx.x, lane = tid & 63, lj = R::lj(), rb = R::base(), wbase = tid & ~63;
    RwSlot *row_slots = s_slot + wbase + rb;
    const int n_las = a.las->n;
    int64_t work_n = PAIRWISE ? a.redo_cnt[a.cls] : a.tier_info[a.cls];      // PAIRWISE: the beams the first instantiation deferred
    if (work_n > a.work_hi) work_n = a.work_hi;
    const int64_t work_off = (int64_t)a.cls * a.tier_stride;
    const int32_t *work_list = PAIRWISE ? a.redo_list : a.tier_list;
    const int waves = (int)gridDim.x * (RW_BLOCK / 64);
    const int wave = (int)blockIdx.x * (RW_BLOCK / 64) + (tid >> 6);
    const double c_tau = 299792458.0 * 1e-8;
    const double delta_beam = a.beam_div_deg * (SG_PI / 180.0);          // np.radians(beam_divergence), :289
    for (int64_t chunk = (int64_t)wave * RPW; chunk < work_n; chunk += (int64_t)waves * RPW) {
        const int64_t bi = chunk + (rb / G);
        const bool live = bi < work_n;                                    // row-uniform
        int32_t g = 0;
        int f = 0, ch = 0;
        T px = 1, py = 0, pz = 0;
        SgTable tab{};
        bool act = false;
        if (live) {
            g = work_list[work_off + bi];
            f = sg_frame_of(a, g);
            const T *row = sg_row<T>(a, f, g);
            px = row[0]; py = row[1]; pz = row[2];
            ch = (int)row[4];                                             // a flagged beam was simulated: valid channel
            tab = a.frame_tables[(int64_t)f * n_las + ch];
            act = tab.entries != nullptr;
        }
        T d_t = 0;
        SgBeamGeo geo{};
        if (act) geo = sg_beam_geometry<T>(px, py, pz, a.beam_div_deg, EXACT, d_t);
        // ---- phase 1: candidate scan (G records per step) and rank sort ------------------------------------------------------
        int nh;
        double a1, a2, rho;
        const int L = rw_scan_sort<G>(act, geo, tab, row_slots, nh, a1, a2, rho);
        if (act && nh > G && lj == 0) {                                   // a listed beam fits its tier by construction
            atomicCAS(&a.status[0], 0, 6 /* SNOWGPU_E_OVERFLOW */);
            atomicCAS(&a.status[1], -1, g);
        }
        // ---- phase 2: compute_occlusion_dict (simulation.py:252-295) ----------------------------------------------------
        const bool row_on = act && L > 0;
        double ra = geo.theta_r, la = geo.theta_l;
        if (act && ra > la) {                                             // :260-263
            ra = ra - SG_TWO_PI;
            if (lj < L && a1 > a2) a1 = a1 - SG_TWO_PI;
        }
        double e_min = ra < la ? ra : la, e_max = ra < la ? la : ra;
        {
            double mn = e_min, mx = e_max;
            if (lj < L) { mn = fmin(mn, fmin(a1, a2)); mx = fmax(mx, fmax(a1, a2)); }
            e_min = R::rmin(mn); e_max = R::rmax(mx);
        }
        if (!(lj < L)) { a1 = INFINITY; a2 = INFINITY; }
        double own_sum, tgt_sum;
        int own_cnt, tgt_cnt;
        rw_slot_walk<G, PAIRWISE>(row_on, a1, a2, ra, la, e_min, e_max, own_sum, own_cnt, tgt_sum, tgt_cnt);
        bool deferred = false;                                            // row-uniform
        if constexpr (!PAIRWISE) {
            const int worst = R::rmaxi(own_cnt);
            deferred = row_on && worst >= 8;                              // a flake owns 8 or more slots: NumPy's pairwise blocks (rare)
            if (deferred && lj == 0) a.redo_list[work_off + atomicAdd(&a.redo_cnt[a.cls], 1)] = g;
        }
        const bool has = row_on && !deferred && lj < L && own_cnt > 0;    // the flake owns a slot: it enters the dict (:288)
        const unsigned long long hm = R::mask(has);
        const int S = (int)__popcll(hm), t_me = (int)__popcll(hm & R::below());
        const double ratio = has ? sg_clip01(own_sum / delta_beam) : 0.0; // :288-290
        const double tratio = sg_clip01(tgt_sum / delta_beam);            // :292-293
        if (a.dbg_count && row_on && !deferred) {
            double *drj = a.dbg_rj + (int64_t)g * a.dbg_cap, *dra = a.dbg_ratio + (int64_t)g * a.dbg_cap;
            if (has && t_me < a.dbg_cap) { drj[t_me] = rho; dra[t_me] = ratio; }
            if (lj == 0) {
                a.dbg_count[g] = S + 1;
                if (S < a.dbg_cap) { drj[S] = geo.d; dra[S] = tratio; }
            }
        } else if (a.dbg_count && act && lj == 0) {                       // no flake met after all: the hard target alone
            a.dbg_count[g] = 1;
            a.dbg_rj[(int64_t)g * a.dbg_cap] = geo.d;
            a.dbg_ratio[(int64_t)g * a.dbg_cap] = sg_clear_beam_ratio(geo.theta_c, a.beam_div_deg);
        }
        // ---- phase 3a: amplitude and bin window of every scatterer (simulation.py:137-146), one lane each -----------------
        const bool pow_on = row_on && !deferred && S > 0;                 // S == 0: no flake owns a slot -> label 0 (:133)
        int range_err = 0;
        double tamp = 0.0, td = geo.d;
        int tk0 = 0, tk1 = 0;
        double amp = 0.0;
        int k0 = 0, k1 = 0;
        if (pow_on) {
            const int max_i = a.las->max_i[ch];
            const double beta_0 = 1 * 1e-6 / SG_PI;                       // :108
            const double ca_p0 = (0.9 * max_i) / beta_0;                  // :140-141 (also used for the hard target, Q1)
            if (has) {
                k0 = (int)ceil(rho * 10);                                 // :145
                k1 = (int)(floor((rho + c_tau) * 10) + 1);                // :146
                amp = (((ca_p0 * beta_0) * ratio) * sg_xsi(rho)) / (rho * rho);   // :549
            }
            if constexpr (SgReal<T>::is_f32) {                            // the hard target keeps its float32 range
                const float r = (float)d_t;
                tk0 = (int)ceilf(r * 10.0f);
                float ee = r + (float)c_tau;
                ee = ee * 10.0f;
                ee = floorf(ee) + 1.0f;
                tk1 = (int)ee;
                const float r2 = r * r;
                tamp = (((ca_p0 * beta_0) * tratio) * sg_xsi(r)) / (double)r2;
            } else {
                const double r = (double)d_t;
                tk0 = (int)ceil(r * 10);
                tk1 = (int)(floor((r + c_tau) * 10) + 1);
                tamp = (((ca_p0 * beta_0) * tratio) * sg_xsi(r)) / (r * r);
            }
            if (k1 > SG_RBINS) { range_err = 1; k1 = SG_RBINS; }          // reference: IndexError (:149)
            if (tk1 > SG_RBINS) { range_err = 1; tk1 = SG_RBINS; }
            if (k0 < 0) k0 = 0;
            if (tk0 < 0) tk0 = 0;
        }
        if (__any(range_err != 0)) {
            if (range_err && (has || lj == 0)) { atomicCAS(&a.status[0], 0, 4 /* SNOWGPU_E_RANGE */); atomicCAS(&a.status[1], -1, g); }
        }
        // scatterers to lanes 0 .. S - 1 of the row (dict order), and into the scratch for the others to read
        RW_LDS_FENCE();
        if (pow_on && has) { RwSlot &sl = row_slots[t_me]; sl.a = amp; sl.c = rho; sl.k = k0 | (k1 << 16); }
        RW_LDS_FENCE();
        double A = 0.0, Rr = 0.0;
        int K0 = 0, K1 = 0;
        if (pow_on && lj < S) { const RwSlot sl = row_slots[lj]; A = sl.a; Rr = sl.c; K0 = sl.k & 0xffff; K1 = sl.k >> 16; }
        // ---- phase 3b, stage A: the bins that can hold the maximum (bounds of sg_power_plan) ----------------------------------
        int ka = 0, kb = -1, tka = 0, tkb = -1;
        if (pow_on) {
            double amax = R::rmax(lj < S ? A : 0.0);
            amax = fmax(amax, tamp);
            const double floor_ = 0.9966 * amax;
            if (lj < S) rw_zone<G>(row_slots, S, lj, A, K0, K1, Rr, tamp, tk0, tk1, floor_, ka, kb);
            if (lj == 0) rw_zone<G>(row_slots, S, S, tamp, tk0, tk1, td, tamp, tk0, tk1, floor_, tka, tkb);   // the hard target's zone: lane 0 carries it
        }
        // ---- stage B: the wave numbers its (scatterer, bin) pairs and every lane evaluates one ----------------------------------
        const int c_own = (pow_on && kb >= ka) ? kb - ka + 1 : 0, c_tgt = (pow_on && lj == 0 && tkb >= tka) ? tkb - tka + 1 : 0;
        const int cnt = c_own + c_tgt;
        int incl = cnt;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const int excl = incl - cnt;
        const int total = __shfl(incl, 63);
        double best = 0.0;
        int k_best = 0;
        for (int base = 0; base < total; base += 64) {
            const int p = base + lane;
            const bool valid = p < total;
            int lo = 0, hi = 63;                                          // owner = first lane whose inclusive count exceeds p
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi) >> 1;
                const int v = __shfl(incl, mid);
                if (v > p) hi = mid; else lo = mid + 1;
            }
            const int o = lo & 63;
            const int j = p - __shfl(excl, o);
            const int co = __shfl(c_own, o), kao = __shfl(ka, o), tkao = __shfl(tka, o);
            const int So = __shfl(S, o);
            const double tampo = __shfl(tamp, o), tdo = __shfl(td, o);
            const int tk0o = __shfl(tk0, o), tk1o = __shfl(tk1, o);
            double sm = -1.0;
            int kk = 0x7fffffff;
            const int orow = o & ~(G - 1);
            if (valid) {
                kk = j < co ? kao + j : tkao + (j - co);
                const double Rk = EXACT ? a.rgrid[kk < SG_RBINS ? kk : SG_RBINS - 1] : sg_range_bin(kk);
                const RwSlot *os = s_slot + wbase + orow;
                sm = 0.0;                                                 // :135 np.zeros
                for (int i = 0; i < So; ++i) {                            // flakes covering the bin, in range order ...
                    const int kw = os[i].k;
                    if (kk >= (kw & 0xffff) && kk < (kw >> 16)) sm += sg_power_term<EXACT>(os[i].a, Rk, os[i].c);   // :149
                }
                if (kk >= tk0o && kk < tk1o) sm += sg_power_term<EXACT>(tampo, Rk, tdo);                             // ... then the hard target
            }
            // every row folds the results that are its own: first maximum (:151) = larger sum, smaller bin on equal sums
            RW_LDS_FENCE();
            s_res[tid] = sm; s_resk[tid] = valid ? (kk | (orow << 16)) : -1;
            RW_LDS_FENCE();
            double rbest = -1.0;
            int rk = 0x7fffffff;
            for (int i = lj; i < 64; i += G) {
                const int kr = s_resk[wbase + i];
                if (kr >= 0 && (kr >> 16) == rb) {
                    const double v = s_res[wbase + i];
                    const int k = kr & 0xffff;
                    if (v > rbest || (v == rbest && k < rk)) { rbest = v; rk = k; }
                }
            }
#pragma unroll
            for (int ox = 1; ox < G; ox <<= 1) {
                const double v = __shfl_xor(rbest, ox);
                const int k = __shfl_xor(rk, ox);
                if (v > rbest || (v == rbest && k < rk)) { rbest = v; rk = k; }
            }
            if (rbest > best || (rbest == best && rk < k_best)) { best = rbest; k_best = rk; }
            RW_LDS_FENCE();
        }
        // ---- phase 3c: decision and record (one lane per beam writes) ----------------------------------------------------------
        SgBeamOut o;
        o.overflow = 0; o.range_error = 0; o.diff2 = 0.0; o.has_power = 0; o.n_flakes = 0; o.n_hits = 0; o.label = 0; o.new_i = 0; o.k_best = 0;
        uint32_t rec = 0;
        if (pow_on) {
            sg_beam_decide(geo.d, ch, a.las, best, k_best, o);
            rec = sg_pack_record(o);
        }
        if (live && !deferred && lj == 0) a.rec[g] = rec;
        sg_add_diff2(a.diff2, live && !deferred && lj == 0, f, (long long)o.diff2);
    }
}


// ------------------------------------------------------------------------------------------------
#define RW_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

template <typename T, int G>
static int launch_rows_t(const SgBeamArgs *a, hipStream_t st)
{
    const int64_t n = (int64_t)a->work_hi - a->work_lo;
    if (n <= 0) return 0;
    int dev_id = 0, cus = 256;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    if (cus <= 0) cus = 256;
    constexpr int RPW = 64 / G, WPB = RW_BLOCK / 64;
    // persistent waves: what the chip holds (8 blocks of 256 per CU at most), fewer when the class cannot be longer
    const int64_t want = (n + (int64_t)RPW * WPB - 1) / ((int64_t)RPW * WPB);
    const unsigned blocks = (unsigned)std::min<int64_t>(want, (int64_t)cus * 8);
    if (a->exact_math) hipLaunchKernelGGL((k_rows<T, G, true, false>), dim3(blocks), dim3(RW_BLOCK), 0, st, *a);
    else hipLaunchKernelGGL((k_rows<T, G, false, false>), dim3(blocks), dim3(RW_BLOCK), 0, st, *a);
    RW_CHECK_LAUNCH();
    // the beams it deferred (an owner with 8 or more elementary slots): nearly always none -- a small grid that leaves at once
    const unsigned redo_blocks = std::min(blocks, 64u);
    if (a->exact_math) hipLaunchKernelGGL((k_rows<T, G, true, true>), dim3(redo_blocks), dim3(RW_BLOCK), 0, st, *a);
    else hipLaunchKernelGGL((k_rows<T, G, false, true>), dim3(redo_blocks), dim3(RW_BLOCK), 0, st, *a);
    RW_CHECK_LAUNCH();
    return 0;
}

// class a->cls of the tier lists, capacity lmax (8, 16 or 63), entries [0, work_hi)
extern "C" int sg_launch_rows(const SgBeamArgs *a, int dtype, int lmax, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        if (lmax <= 8) return launch_rows_t<float, 8>(a, st);
        if (lmax <= 16) return launch_rows_t<float, 16>(a, st);
        return launch_rows_t<float, 64>(a, st);
    }
    if (lmax <= 8) return launch_rows_t<double, 8>(a, st);
    if (lmax <= 16) return launch_rows_t<double, 16>(a, st);
    return launch_rows_t<double, 64>(a, st);
}
