#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <vector>
int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    const int F = 256, N = 131072;
    std::vector<float> in((size_t)F * N * 5, 1.0f), out((size_t)F * N * 5);
    std::vector<uint32_t> meta((size_t)F * N);
    std::vector<float> inten((size_t)F * N, 3.0f);
    for (size_t i = 0; i < meta.size(); ++i) { size_t r = i % N; meta[i] = (uint32_t)(r) | ((i % 97 == 0 ? 1u : 0u) << 30); }
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
            for (int f = t; f < F; f += T) {
                const size_t base = (size_t)f * N;
                const int kept = N * 94 / 100;
                for (int j = 0; j < kept; ++j) {
                    const uint32_t m = meta[base + j];
                    const uint32_t src = m & 0x3fffffffu, lab = m >> 30;
                    const float *ip = &in[(base + src) * 5];
                    float *o = &out[(base + j) * 5];
                    o[0] = ip[0]; o[1] = ip[1]; o[2] = ip[2]; o[3] = inten[base + j]; o[4] = lab == 3 ? ip[4] : (float)lab;
                }
            }
        });
        for (auto &x : th) x.join();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d: %.2f ms for %d frames (%.1f Gpts/s)\n", T, ms, F, F * (double)N / ms / 1e6);
    }
}
