// pmc_calib.hip -- kernels with KNOWN HBM byte counts, in the access shapes of the engine's kernels, to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream).
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -o c -- ./pmc_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -o c -- ./pmc_calib
// Every kernel touches 640 MiB (32 Mi rows of 20 bytes, the row buffer of a 256-sweep batch), far beyond L2 + MALL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// the channel sort's first pass: one thread per row, reads field 4 of every 20-byte row (every line is fetched), writes 1 byte per row
__global__ void __launch_bounds__(256) cal_rows_field_read(const float *rows, unsigned char *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (unsigned char)rows[5 * i + 4];
}
// a kernel that reads all five fields of every row (scan pass / compaction) and writes the row back elsewhere (compaction's scatter)
__global__ void __launch_bounds__(256) cal_rows_copy(const float *rows, float *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { for (int c = 0; c < 5; ++c) out[5 * i + c] = rows[5 * i + c]; }
}
template <typename V> __global__ void __launch_bounds__(256) cal_stream(const V *src, V *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
template <typename V> __global__ void __launch_bounds__(256) cal_read_only(const V *src, V *sink, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { V v = src[i]; if (*(const unsigned *)&v == 0x12345678u) sink[0] = v; }
}
struct Rec64 { double a[8]; };
__global__ void __launch_bounds__(256) cal_rec64_read(const Rec64 *src, double *sink, size_t n)     // one 64-byte record per lane (table records)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const Rec64 r = src[i]; if (r.a[0] + r.a[7] == 123.456) sink[0] = r.a[3]; }
}

int main()
{
    const size_t rows = (size_t)32 << 20, bytes = rows * 20;
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    auto grid = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(cal_rows_field_read, grid(rows), dim3(256), 0, 0, (const float *)a, (unsigned char *)b, rows);
        hipLaunchKernelGGL(cal_rows_copy, grid(rows), dim3(256), 0, 0, (const float *)a, (float *)b, rows);
        hipLaunchKernelGGL(cal_stream<float>, grid(bytes / 4), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4);
        hipLaunchKernelGGL(cal_stream<double>, grid(bytes / 8), dim3(256), 0, 0, (const double *)a, (double *)b, bytes / 8);
        hipLaunchKernelGGL(cal_stream<float4>, grid(bytes / 16), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, bytes / 16);
        hipLaunchKernelGGL(cal_read_only<float>, grid(bytes / 4), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4);
        hipLaunchKernelGGL(cal_read_only<double>, grid(bytes / 8), dim3(256), 0, 0, (const double *)a, (double *)b, bytes / 8);
        hipLaunchKernelGGL(cal_rec64_read, grid(bytes / 64), dim3(256), 0, 0, (const Rec64 *)a, (double *)b, bytes / 64);
    }
    CK(hipDeviceSynchronize());
    printf("bytes per pass %zu\n", bytes);
    return 0;
}
