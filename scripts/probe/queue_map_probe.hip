// Which streams share a hardware queue?  S streams are created (optionally each touched once in creation order), then a set of them runs one
// spinning kernel each (8 blocks: no contention for CUs); concurrent streams finish in one kernel time, streams on one queue in the sum.
// hipcc --offload-arch=gfx950 -O2 queue_map_probe.hip -o /tmp/queue_map_probe; GPU_MAX_HW_QUEUES=4 /tmp/queue_map_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 9999) *sink = 1;
}
static double run(const std::vector<hipStream_t> &st, const std::vector<int> &use, long long ticks)
{
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 4; ++r)
        for (int i : use) spin<<<8, 64, 0, st[i]>>>(ticks, nullptr);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 4;
}
int main(int argc, char **argv)
{
    const int S = 16;
    const bool touch = argc > 1 && argv[1][0] == 't';
    std::vector<hipStream_t> st(S);
    for (int i = 0; i < S; ++i) {
        hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        if (touch) { spin<<<1, 64, 0, st[i]>>>(1, nullptr); }
    }
    hipDeviceSynchronize();
    const long long ticks = 200000;                      // 100 MHz wall clock: 2 ms
    printf("touch-at-creation=%d; one kernel = %.2f ms\n", (int)touch, run(st, {0}, ticks));
    const std::vector<std::vector<int>> sets = {{0, 1}, {0, 1, 2}, {0, 1, 2, 3}, {0, 1, 2, 3, 4}, {0, 4}, {0, 4, 8}, {0, 4, 8, 12}, {1, 6, 11}, {0, 2, 4, 6}, {3, 7}, {0, 1, 2, 3, 4, 5, 6, 7}};
    for (const auto &u : sets) {
        printf("streams");
        for (int i : u) printf(" %d", i);
        printf(": %.2f ms per round\n", run(st, u, ticks));
    }
    return 0;
}
