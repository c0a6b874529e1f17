// chain_probe.hip -- what does link traffic cost a CHAIN of small dependent kernels (one stream, 45 launches)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
        const uint4 v = src[i];
        uint32_t *d = (uint32_t *)&dst[i];
        __builtin_nontemporal_store(v.x, d); __builtin_nontemporal_store(v.y, d + 1);
        __builtin_nontemporal_store(v.z, d + 2); __builtin_nontemporal_store(v.w, d + 3);
    }
}
// one link of the chain: touches 2 MB of HBM (like a small per-chunk kernel)
__global__ void __launch_bounds__(256) k_link(float4 *buf, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { float4 v = buf[i]; v.x += 1.f; buf[i] = v; }
}

int main()
{
    const size_t bytes = (size_t)42 << 20, n16 = bytes / 16;
    void *d_a, *d_b, *d_c, *h_up, *h_dn;
    CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes)); CK(hipMalloc(&d_c, (size_t)64 << 20));
    CK(hipHostMalloc(&h_up, bytes, hipHostMallocPortable)); CK(hipHostMalloc(&h_dn, bytes, hipHostMallocPortable));
    memset(h_up, 1, bytes); memset(h_dn, 0, bytes);
    CK(hipMemset(d_a, 1, bytes));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s_chain, s_up, s_dn;
    CK(hipStreamCreateWithFlags(&s_chain, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_up, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&s_dn, hipStreamNonBlocking, least));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"nothing beside it", "SDMA upload", "SDMA download", "kernel download 16 blocks", "kernel download 16 blocks, nontemporal",
                           "kernel download 64 blocks", "SDMA upload + kernel download 16", "SDMA upload + SDMA download", "kernel download 4 blocks", "kernel download 8 blocks"};
    for (int links : {45, 12}) {
        for (int mode = 0; mode < 10; ++mode) {
            float best = 1e9f;
            for (int r = 0; r < 4; ++r) {
                CK(hipDeviceSynchronize());
                // enough traffic to cover the chain: 3 copies of 42 MB
                for (int k = 0; k < 3; ++k) {
                    if (mode == 1 || mode == 6 || mode == 7) CK(hipMemcpyAsync(d_b, h_up, bytes, hipMemcpyHostToDevice, s_up));
                    if (mode == 2 || mode == 7) CK(hipMemcpyAsync(h_dn, d_a, bytes, hipMemcpyDeviceToHost, s_dn));
                    if (mode == 3 || mode == 6) hipLaunchKernelGGL(k_copy16, dim3(16), dim3(256), 0, s_dn, (uint4 *)h_dn, (const uint4 *)d_a, n16);
                    if (mode == 4) hipLaunchKernelGGL(k_copy16_nt, dim3(16), dim3(256), 0, s_dn, (uint4 *)h_dn, (const uint4 *)d_a, n16);
                    if (mode == 5) hipLaunchKernelGGL(k_copy16, dim3(64), dim3(256), 0, s_dn, (uint4 *)h_dn, (const uint4 *)d_a, n16);
                    if (mode == 8) hipLaunchKernelGGL(k_copy16, dim3(4), dim3(256), 0, s_dn, (uint4 *)h_dn, (const uint4 *)d_a, n16);
                    if (mode == 9) hipLaunchKernelGGL(k_copy16, dim3(8), dim3(256), 0, s_dn, (uint4 *)h_dn, (const uint4 *)d_a, n16);
                }
                CK(hipEventRecord(e0, s_chain));
                for (int k = 0; k < links; ++k) hipLaunchKernelGGL(k_link, dim3(128), dim3(256), 0, s_chain, (float4 *)d_c, ((size_t)2 << 20) / 16);
                CK(hipEventRecord(e1, s_chain));
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("%2d-launch chain, %-42s: %.3f ms (%.1f us per launch)\n", links, names[mode], best, best * 1e3 / links);
        }
    }
    return 0;
}
