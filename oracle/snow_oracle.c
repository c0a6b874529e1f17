/*
 * snow_oracle.c -- CPU restatement of the snowfall hot path of SysCV/LiDAR_snow_sim.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (lidar_snow_sim_amd/, bench.py's
 * timed GPU leg) may link, import or call this file.  It is used by tests/, by
 * __graft_entry__.smoke() as the checker and by bench.py's `cpu_baseline` leg.
 *
 * Parity status: PINNED.  Every function below is checked against golden vectors produced by
 * importing the reference itself in the build container (tests/golden/make_golden.py), with
 * NumPy's SIMD dispatch disabled so that NumPy's float paths are glibc libm -- the same libm
 * this file links.  See DESIGN.md "Oracle and the two reference flavours".
 *
 * All citations are relative to the reference checkout (tools/snowfall/...).
 * Scalar, single-threaded, float64 unless the reference computes in the input dtype.
 * Compile with -ffp-contract=off: the reference (NumPy) never fuses a multiply with an add.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SO_PI 3.141592653589793 /* np.pi, simulation.py:26 / geometry.py:10 */

typedef struct {
    double x, y, r;   /* table row (x, y, disk radius), simulation.py:329-330 */
    double rho;       /* np.linalg.norm([x, y], axis=0), simulation.py:332 */
    double phi;       /* arctan2(y, x) wrapped to [0, 2pi], simulation.py:351-352 */
    double t0, t1;    /* tangent angles (right, left), geometry.py:32-80 */
    int32_t bad;      /* != 0: the reference would raise / mis-align on this row */
} so_flake;

/* ------------------------------------------------------------------------------------------ */
/* np.add.reduce on a contiguous float64 vector: 0.0 + pairwise_sum(a, n)
 * (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum).  Used wherever the reference
 * writes `diffs[mask].sum()` (simulation.py:289, 292). */
static double np_pairwise(const double *a, int64_t n)
{
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
    }
}
static double np_sum(const double *a, int64_t n) { return 0.0 + np_pairwise(a, n); }

static double clip01(double v) /* np.clip(v, 0, 1), simulation.py:290, 293 */
{
    if (v < 0.0) return 0.0;
    if (v > 1.0) return 1.0;
    return v;
}

/* geometry.py:68-70 and :219-221 -- "is this ray within 90 deg of that direction" */
static int forward_of(double ray, double centre)
{
    double d = ray - centre;
    return (fabs(d) < SO_PI / 2) || (fabs(d - 2 * SO_PI) < SO_PI / 2) || (fabs(d + 2 * SO_PI) < SO_PI / 2);
}

/* ------------------------------------------------------------------------------------------ */
/* Per-flake quantities.  The reference recomputes them for every beam (simulation.py:351-354,
 * :405; geometry.py:138-190, :32-80) but none of them depends on the beam, so hoisting is
 * value-preserving. */
void so_prepare_table(const double *xyr, int64_t K, so_flake *out)
{
    for (int64_t k = 0; k < K; k++) {
        so_flake f;
        double x = xyr[3 * k], y = xyr[3 * k + 1], r = xyr[3 * k + 2];
        f.x = x; f.y = y; f.r = r; f.bad = 0;
        f.rho = sqrt(x * x + y * y);                       /* simulation.py:332, :413 */
        f.phi = atan2(y, x);                               /* simulation.py:351 */
        if (f.phi < 0) f.phi = f.phi + 2 * SO_PI;          /* simulation.py:352 */

        /* geometry.tangents_from_origin, geometry.py:161-185 */
        double a[2], b[2];
        double disc = r * sqrt(x * x + y * y - r * r);     /* :161 */
        if (fabs(x) - r == 0) {                            /* :163 case_1 */
            a[0] = 1.0; b[0] = 0.0;                        /* :166 */
            a[1] = (y * y - x * x) / (2 * x * y); b[1] = -1.0; /* :167 */
        } else {
            a[0] = (-x * y + disc) / (r * r - x * x);      /* :169 */
            a[1] = (-x * y - disc) / (r * r - x * x);      /* :170 */
            b[0] = -1.0; b[1] = -1.0;                      /* :171-172 */
        }
        /* geometry.tangent_lines_to_tangent_angles, geometry.py:50-78 */
        double ang[2];
        for (int i = 0; i < 2; i++) {
            double ray1 = atan(-a[i] / b[i]);              /* :50 */
            double ray2 = ray1 + SO_PI;                    /* :51 (before ray1 is fixed up) */
            if (ray1 < 0) ray1 = ray1 + 2 * SO_PI;         /* :54 */
            ray1 = fabs(ray1);                             /* :55 */
            if (b[i] == 0) { ray1 = SO_PI / 2; ray2 = 3 * SO_PI / 2; } /* :58-59 */
            int ok1 = forward_of(ray1, f.phi), ok2 = forward_of(ray2, f.phi); /* :66-70 */
            if (ok1 + ok2 != 1) f.bad = 1;                 /* :72 would raise or mis-align */
            ang[i] = ok1 ? ray1 : ray2;
        }
        double lo = ang[0] < ang[1] ? ang[0] : ang[1];     /* :74 sort(axis=1) */
        double hi = ang[0] < ang[1] ? ang[1] : ang[0];
        if (hi - lo > SO_PI) { f.t0 = hi; f.t1 = lo; }     /* :77-78 swap across the seam */
        else { f.t0 = lo; f.t1 = hi; }
        if (!(f.rho > r)) f.bad = 1;                       /* disk contains the origin: sqrt(<0) */
        out[k] = f;
    }
}

/* ------------------------------------------------------------------------------------------ */
typedef struct { double a1, a2, rho; int64_t src; } so_iv;

static int iv_cmp(const void *pa, const void *pb)
{
    const so_iv *a = (const so_iv *)pa, *b = (const so_iv *)pb;
    if (a->rho < b->rho) return -1;
    if (a->rho > b->rho) return 1;
    return (a->src > b->src) - (a->src < b->src);
}
static int dbl_cmp(const void *pa, const void *pb)
{
    double a = *(const double *)pa, b = *(const double *)pb;
    return (a > b) - (a < b);
}
static int64_t find_eq(const double *e, int64_t n, double v) /* binary_angle_search, :197-228 */
{
    int64_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        if (e[mid] == v) return mid;
        if (e[mid] > v) hi = mid - 1; else lo = mid + 1;
    }
    return -1;
}

typedef struct {
    so_iv *iv; double *ends; double *diffs; double *tmp; int64_t *assign; int64_t cap;
} so_scratch;

static int scratch_init(so_scratch *s, int64_t K)
{
    s->cap = K;
    s->iv = (so_iv *)malloc(sizeof(so_iv) * (size_t)(K + 1));
    s->ends = (double *)malloc(sizeof(double) * (size_t)(2 * K + 4));
    s->diffs = (double *)malloc(sizeof(double) * (size_t)(2 * K + 4));
    s->tmp = (double *)malloc(sizeof(double) * (size_t)(2 * K + 4));
    s->assign = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * K + 4));
    return (s->iv && s->ends && s->diffs && s->tmp && s->assign) ? 0 : -1;
}
static void scratch_free(so_scratch *s)
{
    free(s->iv); free(s->ends); free(s->diffs); free(s->tmp); free(s->assign);
}

/* compute_occlusion_dict (simulation.py:231-295) on the near->far sorted rows s->iv[0..L). */
static int64_t occlusion_tail(so_scratch *s, int64_t L, double theta_r, double theta_l, double d,
                              double beam_div_deg, int64_t *key, double *rj, double *ratio)
{
    double ra = theta_r, la = theta_l;
    if (ra > la) {                                         /* :260 */
        ra = ra - 2 * SO_PI;                               /* :261 */
        for (int64_t j = 0; j < L; j++)
            if (s->iv[j].a1 > s->iv[j].a2) s->iv[j].a1 = s->iv[j].a1 - 2 * SO_PI; /* :262-263 */
    }
    int64_t ne = 0;
    s->ends[ne++] = ra;
    for (int64_t j = 0; j < L; j++) { s->ends[ne++] = s->iv[j].a1; s->ends[ne++] = s->iv[j].a2; }
    s->ends[ne++] = la;
    qsort(s->ends, (size_t)ne, sizeof(double), dbl_cmp);   /* :265 sorted(set(...)) */
    int64_t nu = 0;
    for (int64_t i = 0; i < ne; i++)
        if (nu == 0 || s->ends[i] != s->ends[nu - 1]) s->ends[nu++] = s->ends[i];
    int64_t ns = nu - 1;                                   /* :266-267 */
    for (int64_t k = 0; k < ns; k++) { s->diffs[k] = s->ends[k + 1] - s->ends[k]; s->assign[k] = -1; }

    double delta = beam_div_deg * (SO_PI / 180.0);         /* np.radians, :289 */
    int64_t n_out = 0;
    for (int64_t j = 0; j < L; j++) {                      /* :273 */
        int64_t i1 = find_eq(s->ends, nu, s->iv[j].a1);    /* :277 */
        int64_t i2 = find_eq(s->ends, nu, s->iv[j].a2);    /* :278 */
        int made = 0;
        for (int64_t k = (i1 < 0 ? 0 : i1); k < i2; k++)   /* :282 */
            if (s->assign[k] == -1) { s->assign[k] = j; made = 1; }
        if (made) {                                        /* :288-290 */
            int64_t m = 0;
            for (int64_t k = 0; k < ns; k++) if (s->assign[k] == j) s->tmp[m++] = s->diffs[k];
            key[n_out] = j; rj[n_out] = s->iv[j].rho; ratio[n_out] = clip01(np_sum(s->tmp, m) / delta);
            n_out++;
        }
    }
    int64_t m = 0;                                         /* :292-293 */
    for (int64_t k = 0; k < ns; k++) if (s->assign[k] == -1) s->tmp[m++] = s->diffs[k];
    key[n_out] = -1; rj[n_out] = d; ratio[n_out] = clip01(np_sum(s->tmp, m) / delta);
    n_out++;
    return n_out;
}

/* geometry.angles_to_lines, geometry.py:94-106: (a, b) of one beam-limit ray */
static void limit_line(double theta, double *a, double *b)
{
    if (theta == SO_PI / 2 || theta == 3 * SO_PI / 2) { *a = 1.0; *b = 0.0; }
    else { *a = -tan(theta); *b = 1.0; }
}

/*
 * One beam of get_occlusions (simulation.py:338-422) + compute_occlusion_dict (:231-295).
 * `d` is the hard-target range already widened to double (the comparison at :345 is made in
 * float64 whatever the range dtype is).  Output: the dict in insertion order, as parallel arrays
 * key[] (index into the near->far list, or -1), rj[] (flake range; the -1 entry's range is NOT
 * written here -- the caller owns its dtype), ratio[].  Returns the number of entries (>= 1).
 */
static int64_t beam_dict(const so_flake *fl, int64_t K, double theta_r, double theta_l, double d,
                         double beam_div_deg, so_scratch *s,
                         int64_t *key, double *rj, double *ratio, int64_t *n_intersect)
{
    double ar, br, al, bl;
    limit_line(theta_r, &ar, &br);                         /* :333 */
    limit_line(theta_l, &al, &bl);
    double den_r = sqrt(ar * ar + br * br);                /* geometry.py:133 */
    double den_l = sqrt(al * al + bl * bl);
    int wrap = theta_r > theta_l;                          /* :361, :363 */
    int64_t L = 0;
    for (int64_t k = 0; k < K; k++) {
        const so_flake *f = &fl[k];
        if (!(f->rho < d)) continue;                       /* :345 */
        double phi = f->phi;
        int centre = (theta_r <= phi && phi <= theta_l)                         /* :359 */
                  || (wrap && theta_r - 2 * SO_PI <= phi && phi <= theta_l)     /* :360 */
                  || (wrap && theta_r <= phi && phi <= theta_l + 2 * SO_PI);    /* :362 */
        /* geometry.distances_of_points_to_lines, geometry.py:131-135 (c = 0) */
        double dist_r = fabs(((f->x * ar + f->y * br) + 0.0) / den_r);
        double dist_l = fabs(((f->x * al + f->y * bl) + 0.0) / den_l);
        int hit_r = (dist_r < f->r) && forward_of(theta_r, phi);   /* :379-384 */
        int hit_l = (dist_l < f->r) && forward_of(theta_l, phi);   /* :379-385 */
        if (!(centre || hit_r || hit_l)) continue;         /* :389 */
        so_iv v;
        v.a1 = hit_r ? theta_r : f->t0;                    /* geometry.py:26 */
        v.a2 = hit_l ? theta_l : f->t1;                    /* geometry.py:27 */
        v.rho = f->rho;                                    /* :413 */
        v.src = k;
        s->iv[L++] = v;
    }
    if (n_intersect) *n_intersect = L;
    qsort(s->iv, (size_t)L, sizeof(so_iv), iv_cmp);        /* :416-417 */

    return occlusion_tail(s, L, theta_r, theta_l, d, beam_div_deg, key, rj, ratio);
}

/* ------------------------------------------------------------------------------------------ */
/* simulation.py:553-569.  The float32 twin reproduces NumPy-2 (NEP 50) scalar promotion: a
 * Python float next to an np.float32 scalar is computed in float32. */
static double xsi_f64(double R)
{
    if (R <= 0.9) return 0.0;
    if (R >= 1.0) return 1.0;
    double m = (1 - 0) / (1.0 - 0.9);
    double b = 0 - (m * 0.9);
    return m * R + b;
}
static double xsi_f32(float R)
{
    if (R <= (float)0.9) return 0.0;
    if (R >= (float)1.0) return 1.0;
    double m = (1 - 0) / (1.0 - 0.9);
    double b = 0 - (m * 0.9);
    float y = (float)m * R;
    y = y + (float)b;
    return (double)y;
}

typedef struct {
    int32_t channel;        /* laser index 0..63 */
    int32_t min_intensity;  /* simulation.py:72 */
    int32_t max_intensity;  /* :123-126 */
    double focal_slope;     /* :75 */
    double focal_offset;    /* :76, computed by the caller in Python (float ** 2) */
} so_laser;

#define SO_M_EXT 1230 /* simulation.py:113 */

/*
 * process_single_channel (simulation.py:50-194) for the rows of ONE channel.
 *   pts_in/pts_out : M x 5 rows (x, y, z, intensity, channel/label), float32 or float64
 *   R              : the 1230-entry range grid (:116), computed by the caller with NumPy
 *   dump_*         : optional flattened occlusion dicts (L3 fixtures): per beam count, then
 *                    (key, r_j, ratio) triples, capacity dump_cap triples
 * Returns 0, or -1 on allocation failure, -2 if a range >= the grid (reference: IndexError :149).
 */
static int process_channel(int is_f32, const void *pts_in_v, int64_t M, const double *table_xyr, int64_t K,
                           double beam_div_deg, const so_laser *las, const double *R,
                           void *pts_out_v, double *diff_sum_out,
                           int64_t *dump_count, int64_t *dump_key, double *dump_rj, double *dump_ratio,
                           int64_t dump_cap, int64_t *dump_used)
{
    so_flake *fl = (so_flake *)malloc(sizeof(so_flake) * (size_t)(K + 1));
    so_scratch s;
    int64_t *key = (int64_t *)malloc(sizeof(int64_t) * (size_t)(K + 2));
    double *rj = (double *)malloc(sizeof(double) * (size_t)(K + 2));
    double *ratio = (double *)malloc(sizeof(double) * (size_t)(K + 2));
    if (!fl || !key || !rj || !ratio || scratch_init(&s, K)) return -1;
    so_prepare_table(table_xyr, K, fl);

    const double c_tau = 299792458.0 * 1e-8;               /* c * tau_h, :109, :113 */
    const double beta_0 = 1 * 1e-6 / SO_PI;                /* :108 (10 ** -6 == 1e-06) */
    const double half = (beam_div_deg / 2) * (SO_PI / 180.0); /* np.radians(beam_divergence / 2), :96 */
    const int max_i = las->max_intensity, min_i = las->min_intensity;
    double diff_sum = 0.0;
    int64_t used = 0;
    int rc = 0;
    double I[SO_M_EXT];

    for (int64_t j = 0; j < M; j++) {
        double xd, yd, zd, d64, theta_c;
        float d32 = 0.f;
        if (is_f32) {
            const float *p = (const float *)pts_in_v + 5 * j;
            float x = p[0], y = p[1], z = p[2];
            d32 = sqrtf((x * x + y * y) + z * z);          /* :89 (float32 throughout) */
            float tc = atan2f(y, x);                       /* :91 */
            if (tc < 0) tc = tc + (float)(2 * SO_PI);      /* :92 (float32 add) */
            theta_c = (double)tc; d64 = (double)d32; xd = x; yd = y; zd = z;
            memcpy((float *)pts_out_v + 5 * j, p, 5 * sizeof(float));
        } else {
            const double *p = (const double *)pts_in_v + 5 * j;
            xd = p[0]; yd = p[1]; zd = p[2];
            d64 = sqrt((xd * xd + yd * yd) + zd * zd);
            theta_c = atan2(yd, xd);
            if (theta_c < 0) theta_c = theta_c + 2 * SO_PI;
            memcpy((double *)pts_out_v + 5 * j, p, 5 * sizeof(double));
        }
        double theta_r = theta_c - half;                   /* :96 */
        double theta_l = theta_c + half;                   /* :97 */
        if (theta_r < 0) theta_r = theta_r + 2 * SO_PI;    /* :100 */
        if (theta_l < 0) theta_l = theta_l + 2 * SO_PI;
        if (theta_r > 2 * SO_PI) theta_r = theta_r - 2 * SO_PI; /* :101 */
        if (theta_l > 2 * SO_PI) theta_l = theta_l - 2 * SO_PI;

        int64_t n = beam_dict(fl, K, theta_r, theta_l, d64, beam_div_deg, &s, key, rj, ratio, NULL);
        if (dump_count) {
            dump_count[j] = n;
            for (int64_t t = 0; t < n && used < dump_cap; t++, used++) {
                dump_key[used] = key[t]; dump_rj[used] = rj[t]; dump_ratio[used] = ratio[t];
            }
        }

        double new_label, new_int = 0, scale = 1.0;
        int touched = 0;
        if (n > 1) {                                       /* :133 */
            memset(I, 0, sizeof(I));                       /* :135 */
            const double i_snow = 0.9 * max_i;             /* :140 */
            const double CA_P0 = i_snow / beta_0;          /* :141; also used for key -1 (Q1) */
            for (int64_t t = 0; t < n; t++) {              /* :137 */
                int64_t k0, k1;
                double A;
                double r64;
                if (key[t] == -1 && is_f32) {              /* hard target keeps the float32 range */
                    float r = d32;
                    k0 = (int64_t)ceilf(r * 10.0f);        /* :145 */
                    float e = r + (float)c_tau;            /* :146 */
                    e = e * 10.0f;
                    e = floorf(e) + 1.0f;
                    k1 = (int64_t)e;
                    float r2 = r * r;                      /* r_j ** 2 in float32, :549 */
                    A = (((CA_P0 * beta_0) * ratio[t]) * xsi_f32(r)) / (double)r2;
                    r64 = (double)r;
                } else {
                    double r = rj[t];
                    k0 = (int64_t)ceil(r * 10);            /* :145 */
                    k1 = (int64_t)(floor((r + c_tau) * 10) + 1); /* :146 */
                    A = (((CA_P0 * beta_0) * ratio[t]) * xsi_f64(r)) / (r * r); /* :549 */
                    r64 = r;
                }
                if (k1 > SO_M_EXT) { rc = -2; k1 = SO_M_EXT; } /* reference: IndexError :149 */
                for (int64_t k = k0; k < k1; k++) {        /* :148-149 */
                    double sn = sin((SO_PI * (R[k] - r64)) / c_tau);
                    I[k] += A * (sn * sn);
                }
            }
            int64_t kmax = 0;                              /* :151 np.argmax: first maximum */
            for (int64_t k = 1; k < SO_M_EXT; k++) if (I[k] > I[kmax]) kmax = k;
            double i_max = I[kmax];                        /* :152 */
            double d_max = ((double)kmax / 10) - (c_tau / 2); /* :153 */
            double t1 = 1 - d_max / 120;
            i_max += max_i * las->focal_slope * fabs(las->focal_offset - t1 * t1); /* :155 */
            if (i_max < min_i) i_max = min_i;              /* :156 */
            if (i_max > max_i) i_max = max_i;
            int64_t new_i = (int64_t)i_max;                /* :162 / :182 int() truncates */
            if (fabs(d_max - d64) < 2 * (1.0 / 10)) {      /* :158 */
                new_label = 1;                             /* :160 */
                diff_sum += i_snow - (double)new_i;        /* :170 (i_orig was overwritten, Q2) */
            } else {
                new_label = 2;                             /* :174 */
                scale = d_max / d64;                       /* :176 */
            }
            if (new_i < min_i) new_i = min_i;              /* :186 */
            if (new_i > max_i) new_i = max_i;
            new_int = (double)new_i;
            touched = 1;
        } else {
            new_label = 0;                                 /* :192 */
        }
        if (is_f32) {
            float *o = (float *)pts_out_v + 5 * j;
            if (touched) {
                if (new_label == 2) {                      /* :178-180 float32 * float64 -> store f32 */
                    o[0] = (float)((double)o[0] * scale);
                    o[1] = (float)((double)o[1] * scale);
                    o[2] = (float)((double)o[2] * scale);
                }
                o[3] = (float)new_int;                     /* :188 */
            }
            o[4] = (float)new_label;
        } else {
            double *o = (double *)pts_out_v + 5 * j;
            if (touched) {
                if (new_label == 2) { o[0] = xd * scale; o[1] = yd * scale; o[2] = zd * scale; }
                o[3] = new_int;
            }
            o[4] = new_label;
        }
    }
    if (diff_sum_out) *diff_sum_out = diff_sum;
    if (dump_used) *dump_used = used;
    scratch_free(&s);
    free(fl); free(key); free(rj); free(ratio);
    return rc;
}

int so_process_channel_f32(const float *pts_in, int64_t M, const double *table_xyr, int64_t K,
                           double beam_div_deg, const so_laser *las, const double *R,
                           float *pts_out, double *diff_sum,
                           int64_t *dump_count, int64_t *dump_key, double *dump_rj, double *dump_ratio,
                           int64_t dump_cap, int64_t *dump_used)
{
    return process_channel(1, pts_in, M, table_xyr, K, beam_div_deg, las, R, pts_out, diff_sum,
                           dump_count, dump_key, dump_rj, dump_ratio, dump_cap, dump_used);
}
int so_process_channel_f64(const double *pts_in, int64_t M, const double *table_xyr, int64_t K,
                           double beam_div_deg, const so_laser *las, const double *R,
                           double *pts_out, double *diff_sum,
                           int64_t *dump_count, int64_t *dump_key, double *dump_rj, double *dump_ratio,
                           int64_t dump_cap, int64_t *dump_used)
{
    return process_channel(0, pts_in, M, table_xyr, K, beam_div_deg, las, R, pts_out, diff_sum,
                           dump_count, dump_key, dump_rj, dump_ratio, dump_cap, dump_used);
}

/* get_occlusions alone (simulation.py:298-424) on explicit beam limits -- L3 fixtures. */
int so_get_occlusions(const double *beam_angles /* M x 2 */, const double *ranges, int64_t M,
                      const double *table_xyr, int64_t K, double beam_div_deg,
                      int64_t *count, int64_t *key_out, double *rj_out, double *ratio_out,
                      int64_t cap, int64_t *used_out, int64_t *n_intersect)
{
    so_flake *fl = (so_flake *)malloc(sizeof(so_flake) * (size_t)(K + 1));
    so_scratch s;
    int64_t *key = (int64_t *)malloc(sizeof(int64_t) * (size_t)(K + 2));
    double *rj = (double *)malloc(sizeof(double) * (size_t)(K + 2));
    double *ratio = (double *)malloc(sizeof(double) * (size_t)(K + 2));
    if (!fl || !key || !rj || !ratio || scratch_init(&s, K)) return -1;
    so_prepare_table(table_xyr, K, fl);
    int64_t used = 0;
    for (int64_t i = 0; i < M; i++) {
        int64_t L = 0;
        int64_t n = beam_dict(fl, K, beam_angles[2 * i], beam_angles[2 * i + 1], ranges[i], beam_div_deg,
                              &s, key, rj, ratio, &L);
        count[i] = n;
        if (n_intersect) n_intersect[i] = L;
        for (int64_t t = 0; t < n && used < cap; t++, used++) {
            key_out[used] = key[t]; rj_out[used] = rj[t]; ratio_out[used] = ratio[t];
        }
    }
    *used_out = used;
    scratch_free(&s);
    free(fl); free(key); free(rj); free(ratio);
    return 0;
}

/* compute_occlusion_dict alone (simulation.py:231-295) on explicit, already sorted intervals. */
int so_occlusion_dict(double right_angle, double left_angle, const double *intervals /* L x 3 */, int64_t L,
                      double current_range, double beam_div_deg,
                      int64_t *key_out, double *rj_out, double *ratio_out)
{
    so_scratch s;
    if (scratch_init(&s, L + 1)) return -1;
    for (int64_t j = 0; j < L; j++) {
        s.iv[j].a1 = intervals[3 * j]; s.iv[j].a2 = intervals[3 * j + 1];
        s.iv[j].rho = intervals[3 * j + 2]; s.iv[j].src = j;
    }
    int64_t n = occlusion_tail(&s, L, right_angle, left_angle, current_range, beam_div_deg,
                               key_out, rj_out, ratio_out);
    scratch_free(&s);
    return (int)n;
}

/* Per-flake table (rho, phi, t0, t1, bad) for L1 geometry fixtures. */
int so_flake_table(const double *xyr, int64_t K, double *out /* K x 5 */)
{
    so_flake *fl = (so_flake *)malloc(sizeof(so_flake) * (size_t)(K + 1));
    if (!fl) return -1;
    so_prepare_table(xyr, K, fl);
    for (int64_t k = 0; k < K; k++) {
        out[5 * k] = fl[k].rho; out[5 * k + 1] = fl[k].phi;
        out[5 * k + 2] = fl[k].t0; out[5 * k + 3] = fl[k].t1; out[5 * k + 4] = (double)fl[k].bad;
    }
    free(fl);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Multi-core driver for the CPU baseline (SURVEY 8 d: "all host cores -- state the count").
 * A work item is a run of consecutive beams of one (frame, channel): exactly process_channel on
 * those rows (beams are independent, simulation.py:118).  Threads take items off a shared counter.
 * The reference's own parallelism is ThreadPool(cpu_count()) over the 64 channels of ONE frame
 * (simulation.py:498); here frames, channels and beam runs are all in flight, which is what a
 * multi-core C port of the same per-beam algorithm would do. */
#include <pthread.h>

typedef struct {
    int32_t is_f32;
    int32_t rc;            /* out: 0, -1, -2 as process_channel */
    const void *pts_in;
    void *pts_out;
    int64_t M;
    const double *table_xyr;
    int64_t K;
    double beam_div_deg;
    so_laser las;
    double diff_sum;       /* out */
} so_item;

typedef struct {
    so_item *items;
    int64_t n;
    int64_t next;
    const double *R;
} so_pool;

static void *so_worker(void *arg)
{
    so_pool *p = (so_pool *)arg;
    for (;;) {
        const int64_t i = __atomic_fetch_add(&p->next, 1, __ATOMIC_RELAXED);
        if (i >= p->n) break;
        so_item *it = &p->items[i];
        it->rc = process_channel(it->is_f32, it->pts_in, it->M, it->table_xyr, it->K, it->beam_div_deg, &it->las, p->R,
                                 it->pts_out, &it->diff_sum, NULL, NULL, NULL, NULL, 0, NULL);
    }
    return NULL;
}

/* Returns the number of threads that ran (>= 1), or -1 on failure. */
int so_process_items_mt(so_item *items, int64_t n, const double *R, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n) n_threads = (int)(n > 0 ? n : 1);
    so_pool pool = {items, n, 0, R};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!th) return -1;
    int started = 0;
    for (int t = 0; t < n_threads - 1; t++) {
        if (pthread_create(&th[started], NULL, so_worker, &pool) != 0) break;
        started++;
    }
    so_worker(&pool);                                      /* the caller is a worker too */
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    return started + 1;
}
