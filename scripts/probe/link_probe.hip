// link_probe.hip -- which engine moves device -> host copies, and what a copy costs the kernels running beside it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/link_probe scripts/probe/link_probe.hip && /tmp/link_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_copy16(uint4 *dst, const uint4 *src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_copy16_nt(uint4 *dst, const uint4 *src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint4 v = src[i];
        __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z); __builtin_nontemporal_store(v.w, &dst[i].w);
    }
}
// ALU-only victim: no memory traffic but one store at the end
__global__ void __launch_bounds__(256) k_alu(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 1e-7f;
    if (a == 123.f) out[0] = a;
}
// HBM-streaming victim
__global__ void __launch_bounds__(256) k_stream(float4 *dst, const float4 *src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { float4 v = src[i]; v.x += 1.f; dst[i] = v; }
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = (size_t)42 << 20, n16 = bytes / 16;
    void *d_a, *d_b, *d_c, *d_d;
    CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes)); CK(hipMalloc(&d_c, (size_t)512 << 20)); CK(hipMalloc(&d_d, (size_t)512 << 20));
    CK(hipMemset(d_a, 1, bytes));
    struct { const char *name; unsigned flags; } kinds[] = {{"default", hipHostMallocDefault}, {"portable", hipHostMallocPortable},
        {"mapped", hipHostMallocMapped}, {"noncoherent", hipHostMallocNonCoherent}, {"writecombined", hipHostMallocWriteCombined}};
    hipStream_t s_copy, s_k;
    CK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    hipEvent_t e0, e1, k0, k1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
    auto victim_alone = [&](int which) {
        float ms = 0;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(k0, s_k));
            if (which == 0) hipLaunchKernelGGL(k_alu, dim3(2048), dim3(256), 0, s_k, (float *)d_c, 20000);
            else hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s_k, (float4 *)d_d, (const float4 *)d_c, ((size_t)512 << 20) / 16);
            CK(hipEventRecord(k1, s_k)); CK(hipEventSynchronize(k1)); CK(hipEventElapsedTime(&ms, k0, k1));
        }
        return ms;
    };
    printf("victims alone: alu %.3f ms, stream(1 GiB moved) %.3f ms\n", victim_alone(0), victim_alone(1));
    for (auto &kd : kinds) {
        void *h = nullptr;
        if (hipHostMalloc(&h, bytes, kd.flags) != hipSuccess) { printf("%s: alloc failed\n", kd.name); (void)hipGetLastError(); continue; }
        std::memset(h, 0, bytes);
        void *h2 = nullptr;
        CK(hipHostMalloc(&h2, bytes, kd.flags));
        std::memset(h2, 0, bytes);
        hipStream_t s_h2;
        CK(hipStreamCreateWithFlags(&s_h2, hipStreamNonBlocking));
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (int mode = 0; mode < 8; ++mode) {
            // 0: hipMemcpyAsync D2H  1: same after a kernel on the stream  2: our kernel 64 blocks  3: 16 blocks  4: 256 blocks  5: H2D hipMemcpyAsync
            for (int victim = -1; victim < 2; ++victim) {
                float best = 1e9f, vbest = 1e9f;
                for (int r = 0; r < 3; ++r) {
                    CK(hipDeviceSynchronize());
                    if (mode == 1) hipLaunchKernelGGL(k_alu, dim3(64), dim3(256), 0, s_copy, (float *)d_b, 10);
                    CK(hipEventRecord(e0, s_copy));
                    if (victim >= 0) CK(hipEventRecord(k0, s_k));
                    if (mode == 6 || mode == 7) {      // an upload in flight on a third stream; 7: the download also waits for an event of the kernel stream
                        CK(hipMemcpyAsync(d_b, h2, bytes, hipMemcpyHostToDevice, s_h2));
                        if (mode == 7) { hipLaunchKernelGGL(k_alu, dim3(64), dim3(256), 0, s_k, (float *)d_c, 10); CK(hipEventRecord(ev, s_k)); CK(hipStreamWaitEvent(s_copy, ev, 0)); }
                        CK(hipMemcpyAsync(h, d_a, bytes, hipMemcpyDeviceToHost, s_copy));
                    } else
                    if (mode <= 1) CK(hipMemcpyAsync(h, d_a, bytes, hipMemcpyDeviceToHost, s_copy));
                    else if (mode == 5) CK(hipMemcpyAsync(d_b, h, bytes, hipMemcpyHostToDevice, s_copy));
                    else hipLaunchKernelGGL(k_copy16, dim3(mode == 2 ? 64 : (mode == 3 ? 16 : 256)), dim3(256), 0, s_copy, (uint4 *)h, (const uint4 *)d_a, n16);
                    CK(hipEventRecord(e1, s_copy));
                    if (victim == 0) hipLaunchKernelGGL(k_alu, dim3(2048), dim3(256), 0, s_k, (float *)d_c, 20000);
                    if (victim == 1) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s_k, (float4 *)d_d, (const float4 *)d_c, ((size_t)512 << 20) / 16);
                    if (victim >= 0) CK(hipEventRecord(k1, s_k));
                    CK(hipDeviceSynchronize());
                    float ms = 0, vms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (victim >= 0) CK(hipEventElapsedTime(&vms, k0, k1));
                    best = ms < best ? ms : best; vbest = vms < vbest ? vms : vbest;
                }
                printf("%-13s mode %d victim %2d: copy %.3f ms = %.1f GB/s%s", kd.name, mode, victim, best, bytes / best / 1e6, victim >= 0 ? "" : "\n");
                if (victim >= 0) printf(", victim %.3f ms\n", vbest);
            }
        }
        CK(hipHostFree(h)); CK(hipHostFree(h2)); CK(hipStreamDestroy(s_h2)); CK(hipEventDestroy(ev));
        if (getenv("PROBE_ONE")) break;
    }
    return 0;
}
