// sg_table_host.h -- filing a particle table on the HOST (snowgpu_upload_table): the bit-exact path for tables that come from the
// reference's .npy files.  Plain C++ (glibc's libm: the flakes' azimuths and tangent angles are the values NumPy computes), no HIP:
// snowgpu_api.cpp uploads the result; tests/host_harness/beam_vs_oracle.cpp scans it with the kernels' own device functions compiled
// for the host.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "sg_common.h"

// Per flake (beam-independent, hoisted out of get_occlusions' beam loop): rho, phi, the two tangent
// angles.  Same operation order as the reference (geometry.py:138-190, :32-80; simulation.py:351-352).
static inline bool sg_host_forward_of(double ray, double centre)
{
    double d = ray - centre;
    return (std::fabs(d) < SG_PI / 2) || (std::fabs(d - SG_TWO_PI) < SG_PI / 2) || (std::fabs(d + SG_TWO_PI) < SG_PI / 2);
}

static inline bool derive_flake(double x, double y, double r, SgEntry *f)
{
    if (!(std::isfinite(x) && std::isfinite(y) && std::isfinite(r)) || !(r > 0)) return false;
    f->x = x; f->y = y; f->r = r;
    f->rho = std::sqrt(x * x + y * y);
    if (!(f->rho > r)) return false;                   // disk contains the origin: sqrt of a negative below
    f->phi = std::atan2(y, x);
    if (f->phi < 0) f->phi = f->phi + SG_TWO_PI;
    double a[2], b[2];
    const double disc = r * std::sqrt(x * x + y * y - r * r);
    if (std::fabs(x) - r == 0) {
        a[0] = 1.0; b[0] = 0.0;
        a[1] = (y * y - x * x) / (2 * x * y); b[1] = -1.0;
    } else {
        a[0] = (-x * y + disc) / (r * r - x * x);
        a[1] = (-x * y - disc) / (r * r - x * x);
        b[0] = b[1] = -1.0;
    }
    double ang[2];
    for (int i = 0; i < 2; ++i) {
        double ray1 = std::atan(-a[i] / b[i]);
        double ray2 = ray1 + SG_PI;
        if (ray1 < 0) ray1 = ray1 + SG_TWO_PI;
        ray1 = std::fabs(ray1);
        if (b[i] == 0) { ray1 = SG_PI / 2; ray2 = 3 * SG_PI / 2; }
        const bool ok1 = sg_host_forward_of(ray1, f->phi), ok2 = sg_host_forward_of(ray2, f->phi);
        if (ok1 == ok2) return false;                  // the reference would raise / mis-align (geometry.py:72)
        ang[i] = ok1 ? ray1 : ray2;
    }
    const double lo = std::min(ang[0], ang[1]), hi = std::max(ang[0], ang[1]);
    if (hi - lo > SG_PI) { f->t0 = hi; f->t1 = lo; } else { f->t0 = lo; f->t1 = hi; }
    return true;
}

static inline int bin_of(double theta, double inv_w, int nb)
{
    theta = std::fmod(theta, SG_TWO_PI);
    if (theta < 0) theta += SG_TWO_PI;
    int b = (int)std::floor(theta * inv_w);
    if (b < 0) b = 0;
    if (b >= nb) b = nb - 1;
    return b;
}

// The table as the kernels read it: every flake's record (derive_flake) filed under every azimuth bin its angular interval
// +- SG_BIN_MARGIN touches (flags bit 0: first bin of the flake), each bin sorted by (range, table row); start[b] .. start[b + 1]
// are bin b's records; one spare record at the end (the scan prefetches entry e + 1).
// Returns 0; 1: row *bad_row is not a disk clear of the origin; 2: flakes so close to the sensor that they cover most azimuths.
static inline int sg_file_table_host(const double *xyr, int64_t k, std::vector<SgEntry> &entries, std::vector<uint32_t> &start,
                                     uint32_t &max_bin, int64_t *bad_row)
{
    const int nb = SG_NBINS;
    const double inv_w = nb / SG_TWO_PI;
    std::vector<SgEntry> fl((size_t)k);
    std::vector<int> b0((size_t)k), span((size_t)k);
    std::vector<uint32_t> count((size_t)nb + 1, 0);
    for (int64_t i = 0; i < k; ++i) {
        if (!derive_flake(xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2], &fl[(size_t)i])) { if (bad_row) *bad_row = i; return 1; }
        SgEntry &f = fl[(size_t)i];
        f.src = (uint32_t)i;
        const double alpha = std::asin(std::min(1.0, f.r / f.rho));
        const double lo = f.phi - alpha - SG_BIN_MARGIN, hi = f.phi + alpha + SG_BIN_MARGIN;
        int s;
        if (hi - lo >= SG_TWO_PI - 2.0 / inv_w) { b0[(size_t)i] = 0; s = nb; }
        else {
            const int bl = bin_of(lo, inv_w, nb), bh = bin_of(hi, inv_w, nb);
            b0[(size_t)i] = bl;
            s = bh - bl;
            if (s < 0) s += nb;
            s += 1;
        }
        span[(size_t)i] = s;
        for (int t = 0; t < s; ++t) count[(size_t)((b0[(size_t)i] + t) % nb)]++;
    }
    start.assign((size_t)nb + 1, 0);
    for (int b = 0; b < nb; ++b) start[(size_t)b + 1] = start[(size_t)b] + count[(size_t)b];
    const size_t n_entries = start[(size_t)nb];
    if (n_entries > (size_t)64 * (size_t)std::max<int64_t>(k, 1) + 4096) return 2;
    entries.assign(n_entries + 1, SgEntry{});      // + one spare record: the scan prefetches entry e + 1
    std::memset(&entries[n_entries], 0, sizeof(SgEntry));
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < k; ++i) {
        for (int t = 0; t < span[(size_t)i]; ++t) {
            const int b = (b0[(size_t)i] + t) % nb;
            SgEntry e = fl[(size_t)i];
            e.flags = (t == 0) ? 1u : 0u;
            entries[fill[(size_t)b]++] = e;
        }
    }
    max_bin = 0;
    for (int b = 0; b < nb; ++b) {
        std::sort(entries.begin() + (ptrdiff_t)start[(size_t)b], entries.begin() + (ptrdiff_t)start[(size_t)b + 1],
                  [](const SgEntry &p, const SgEntry &q) { return p.rho < q.rho || (p.rho == q.rho && p.src < q.src); });
        max_bin = std::max(max_bin, start[(size_t)b + 1] - start[(size_t)b]);
    }
    return 0;
}
