// atan_cr_check.cpp -- sg_atan2_cr / sg_atan_cr against glibc's atan2 / atan on random and structured inputs.
//   g++ -O2 -ffp-contract=off -o /tmp/atan_cr_check scripts/probe/atan_cr_check.cpp -lm && /tmp/atan_cr_check [millions]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../lidar_snow_sim_amd/csrc/sg_atan_cr.h"

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t next() { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static double uni(double lo, double hi) { return lo + (hi - lo) * ((next() >> 11) * 0x1p-53); }

int main(int argc, char **argv)
{
    const long n = (argc > 1 ? atol(argv[1]) : 20) * 1000000L;
    long bad2 = 0, bad1 = 0, shown = 0;
    for (long i = 0; i < n; ++i) {
        double x, y;
        switch (i & 7) {
        case 0: case 1: case 2: x = uni(-120, 120); y = uni(-120, 120); break;                 // lidar points / flake centres
        case 3: x = uni(-120, 120); y = uni(-1e-3, 1e-3); break;                               // near the seam
        case 4: x = uni(-1e-3, 1e-3); y = uni(-120, 120); break;
        case 5: x = (float)uni(-120, 120); y = (float)uni(-120, 120); break;                   // float32 values widened
        case 6: x = ldexp(uni(-1, 1), (int)(next() % 80) - 40); y = ldexp(uni(-1, 1), (int)(next() % 80) - 40); break;
        default: x = uni(-1, 1); y = x * (1.0 + uni(-1e-9, 1e-9)); break;                      // the diagonal
        }
        const double a = atan2(y, x), b = sg_atan2_cr(y, x);
        if (memcmp(&a, &b, 8) != 0) { if (shown++ < 200) printf("M atan2 %a %a %a %a\n", y, x, a, b); ++bad2; }
        const double v = (i & 1) ? y / (x == 0 ? 1 : x) : uni(-4, 4);
        const double c = atan(v), d = sg_atan_cr(v);
        if (memcmp(&c, &d, 8) != 0) { if (shown++ < 200) printf("M atan %a 1 %a %a\n", v, c, d); ++bad1; }
    }
    // special values
    const double sp[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, 5e-324, -5e-324, 1e308, -1e308, 1e-310};
    long bads = 0;
    for (double y : sp) for (double x : sp) { const double a = atan2(y, x), b = sg_atan2_cr(y, x); if (memcmp(&a, &b, 8)) { printf("special atan2(%a, %a): glibc %a ours %a\n", y, x, a, b); ++bads; } }
    printf("%ld inputs: atan2 mismatches %ld, atan mismatches %ld, special-value mismatches %ld\n", n, bad2, bad1, bads);
    return bads ? 1 : 0;
}
