// sg_few.h -- a beam that met a FEW flakes (1 .. N, N <= 3), start to finish in registers (k_power_few).
//
// Phases 2, 3a and 3b of the general path (sg_beam.h: sg_beam_dict, sg_beam_amp, sg_power_plan + sg_eval_group) written out for
// lists of at most N flakes with every loop unrolled and every index a compile-time constant: the same expressions in the same
// order, so every intermediate is the value the general path computes (tests/test_kernel_math.py runs both on the host over
// millions of random beams and compares bit for bit).  What the short list buys:
//   * compute_occlusion_dict (simulation.py:252-295) walks at most 2 N + 1 elementary slots, fewer than eight per owner: np.sum of
//     an owner's slot widths is a running sum (sg_math.h: SgNpSum only blocks from eight addends on);
//   * stage A (which bins can hold the first maximum of the received power) looks at <= N neighbours of a scatterer;
//   * stage B evaluates single bins, not groups of four: the kernel spreads the (beam, bin) pairs of a wave over its lanes.
// Exactness of the pruning: sg_beam.h, "phase 3b".  A bin outside every zone can neither hold nor tie the maximum, and the zones
// are evaluated completely, so the first maximum over them is np.argmax's (simulation.py:151) whatever the order.
#pragma once
#include "sg_beam.h"

template <int N>
struct SgFew {
    double amp[N], rho[N];    // the flakes that own a slot, near -> far: S of them (the others: amp 0, empty window)
    int k0[N], k1[N];         // their bin windows [k0, k1)
    double tamp, d;           // the hard target
    int tk0, tk1;
    int S;
};

// Phases 2 and 3a.  L flakes (1 <= L <= N) in range order: interval angles a1, a2 (geometry.py:14-29), range rho.  Returns S, the
// number of flakes that own an elementary slot (0: the beam keeps label 0, simulation.py:133).
template <typename T, int N>
__device__ __forceinline__ int sg_few_prep(double d, double theta_c, int L, const double (&a1_in)[N], const double (&a2_in)[N], const double (&rho)[N],
                                           int channel, const SgLasers *__restrict__ las, double beam_div_deg, SgFew<N> &p, SgBeamOut &out)
{
    constexpr bool F32 = SgReal<T>::is_f32;
    static_assert(N >= 1 && N <= 3, "an owner must stay below eight slots (2 N + 1 <= 7)");
    const int max_i = las->max_i[channel];                     // (asked for here, used in phase 3a: the load flies while phase 2 computes)
    // -- phase 2
    double a1[N], a2[N];
    double theta_r, theta_l;
    sg_beam_limits(theta_c, beam_div_deg, theta_r, theta_l);
    double ra = theta_r, la = theta_l;
    const bool wrap = ra > la;                                  // :260-263
    if (wrap) ra = ra - SG_TWO_PI;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        a1[j] = a1_in[j]; a2[j] = a2_in[j];
        if (wrap && j < L && a1[j] > a2[j]) a1[j] = a1[j] - SG_TWO_PI;
    }
    const double delta = beam_div_deg * (SG_PI / 180.0);        // :289
    double e_min = ra < la ? ra : la, e_max = ra < la ? la : ra;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (j < L) {
            const double lo = a1[j], hi = a2[j];
            if (lo < e_min) e_min = lo;
            if (hi < e_min) e_min = hi;
            if (lo > e_max) e_max = lo;
            if (hi > e_max) e_max = hi;
        }
    }
    double R[N], tsum = -0.0;                                   // slot widths per owner, of the hard target: running sums
    bool has[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { R[j] = 0.0; has[j] = false; }
    for (double e = e_min; e < e_max;) {
        int own = -1;
        double nxt = e_max;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j < L) {
                if (own < 0 && a1[j] <= e && e < a2[j]) own = j;    // nearest flake covering the slot (:284)
                if (a1[j] > e && a1[j] < nxt) nxt = a1[j];
                if (a2[j] > e && a2[j] < nxt) nxt = a2[j];
            }
        }
        if (ra > e && ra < nxt) nxt = ra;
        if (la > e && la < nxt) nxt = la;
        const double w = nxt - e;
        if (own < 0) tsum += w;                                 // :292-293
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (own == j) { R[j] = has[j] ? R[j] + w : w; has[j] = true; }
        e = nxt;
    }
    // -- phase 3a for the owners, closed up near -> far (simulation.py:137-146)
    const double c_tau = 299792458.0 * 1e-8;
    const double beta_0 = 1 * 1e-6 / SG_PI;
    const double i_snow = 0.9 * max_i;
    const double ca_p0 = i_snow / beta_0;
#pragma unroll
    for (int t = 0; t < N; ++t) { p.amp[t] = 0.0; p.rho[t] = 0.0; p.k0[t] = 0; p.k1[t] = 0; }
    int S = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (j < L && has[j]) {
            const double ratio = sg_clip01((0.0 + R[j]) / delta);   // :288-290
            const double r = rho[j];
            int k0 = (int)ceil(r * 10);                             // :145
            int k1 = (int)(floor((r + c_tau) * 10) + 1);            // :146
            const double amp = (((ca_p0 * beta_0) * ratio) * sg_xsi(r)) / (r * r);   // :549
            if (k1 > SG_RBINS) { out.range_error = 1; k1 = SG_RBINS; }
            if (k0 < 0) k0 = 0;
#pragma unroll
            for (int t = 0; t <= j; ++t)
                if (t == S) { p.amp[t] = amp; p.rho[t] = r; p.k0[t] = k0; p.k1[t] = k1; }
            ++S;
        }
    }
    p.S = S;
    p.d = d;
    p.tamp = 0.0; p.tk0 = 0; p.tk1 = 0;
    if (S == 0) return 0;
    const double ratio_t = sg_clip01((0.0 + tsum) / delta);
    if constexpr (F32) {                                        // the hard target keeps its float32 range
        const float r = (float)(T)d;
        p.tk0 = (int)ceilf(r * 10.0f);
        float ee = r + (float)c_tau;
        ee = ee * 10.0f;
        ee = floorf(ee) + 1.0f;
        p.tk1 = (int)ee;
        const float r2 = r * r;
        p.tamp = (((ca_p0 * beta_0) * ratio_t) * sg_xsi(r)) / (double)r2;
    } else {
        p.tk0 = (int)ceil(d * 10);
        p.tk1 = (int)(floor((d + c_tau) * 10) + 1);
        p.tamp = (((ca_p0 * beta_0) * ratio_t) * sg_xsi(d)) / (d * d);
    }
    if (p.tk1 > SG_RBINS) { out.range_error = 1; p.tk1 = SG_RBINS; }
    if (p.tk0 < 0) p.tk0 = 0;
    out.n_flakes = S;
    out.has_power = 1;
    return S;
}

// Stage A for scatterer TT of the list (TT < N: flake TT, if it exists; TT == N: the hard target), given the best sum so far: its
// bins [ka, kb] that can hold or tie the maximum (kb < ka: none).  The neighbour walks are sg_power_plan's: nearer flakes downwards,
// further ones upwards and the hard target last, each walk ending at the first window that does not reach this one.
template <int N, int TT>
__device__ __forceinline__ void sg_few_zone(const SgFew<N> &p, double best, int &ka, int &kb)
{
    constexpr bool TARGET = TT == N;
    const double c_tau = 299792458.0 * 1e-8;
    const double step = (120 + c_tau) / (SG_RBINS - 1);
    ka = 0; kb = -1;
    double A, r;
    int k0, k1;
    if constexpr (TARGET) { A = p.tamp; r = p.d; k0 = p.tk0; k1 = p.tk1; }
    else { if (TT >= p.S) return; A = p.amp[TT]; r = p.rho[TT]; k0 = p.k0[TT]; k1 = p.k1[TT]; }
    if (!(A > 0.0)) return;
    double oth = 0.0;
    int lo_trim = k0, hi_trim = k1;
    auto visit = [&](double Aj, int q0, int q1, bool first) {   // first: neighbour j is listed before this scatterer
        if (Aj > A || (Aj == A && first)) {
            if (q0 <= k0) { if (q1 > lo_trim) lo_trim = q1; }
            else if (q1 >= k1) { if (q0 < hi_trim) hi_trim = q0; }
        } else oth += Aj;
    };
    bool go = true;
#pragma unroll
    for (int j = N - 1; j >= 0; --j) {                          // nearer scatterers, downwards
        if (j < TT && j < p.S && go) {
            if (p.k1[j] <= k0) go = false;
            else visit(p.amp[j], p.k0[j], p.k1[j], true);
        }
    }
    if constexpr (!TARGET) {
        go = true;
#pragma unroll
        for (int j = 0; j < N; ++j) {                           // further flakes, upwards ...
            if (j > TT && j < p.S && go) {
                if (p.k0[j] >= k1) go = false;
                else visit(p.amp[j], p.k0[j], p.k1[j], false);
            }
        }
        if (go && !(p.tk0 >= k1)) visit(p.tamp, p.tk0, p.tk1, false);   // ... and the hard target, the last of the list
    }
    double amax = p.tamp;
#pragma unroll
    for (int j = 0; j < N; ++j) if (j < p.S) amax = fmax(amax, p.amp[j]);
    const double need = fmax(best, 0.9966 * amax);
    if ((A + oth) * (1.0 + 1e-9) < need) return;
    const double q = (need * (1.0 - 1e-9) - oth * (1.0 + 1e-9)) / A;
    ka = k0; kb = k1 - 1;
    if (q >= 0.5) {
        const double om = q < 1.0 ? 1.0 - q : 0.0;
        const double dl = (double)sqrtf((float)(1.26 * om)) * (1.0 + 1e-6) + 1e-6;
        const double Rc = r + c_tau / 2;
        const double D = dl * (c_tau / SG_PI) + 0.006;
        const int za = (int)ceil((Rc - D) * (1.0 / step)), zb = (int)floor((Rc + D) * (1.0 / step));
        if (za > ka) ka = za;
        if (zb < kb) kb = zb;
    }
    if (lo_trim > ka) ka = lo_trim;
    if (hi_trim - 1 < kb) kb = hi_trim - 1;
}

// the received power of bin k (simulation.py:135, :149): the flakes' terms in range order, then the hard target's
template <bool EXACT, int N>
__device__ __forceinline__ double sg_few_bin(const double (&amp)[N], const double (&rho)[N], const int (&k0)[N], const int (&k1)[N],
                                             double tamp, double d, int tk0, int tk1, int k, const double *__restrict__ rgrid)
{
    const double Rk = EXACT ? rgrid[k < SG_RBINS ? k : SG_RBINS - 1] : sg_range_bin(k);
    double sm = 0.0;
#pragma unroll
    for (int t = 0; t < N; ++t)
        if (k >= k0[t] && k < k1[t]) sm += sg_power_term<EXACT>(amp[t], Rk, rho[t]);   // (a flake that owns nothing: empty window)
    if (k >= tk0 && k < tk1) sm += sg_power_term<EXACT>(tamp, Rk, d);
    return sm;
}
