// tools/micro_write.hip -- what a pure stream of whole-line stores (and a 1:8 read:write stream, the
// footprint build's ratio) sustains on this GPU.  hipcc --offload-arch=gfx950 -O3 -o tools/bin/micro_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void fill_kernel(uint4 *dst, size_t n16, int per_thread)
{
    // a workgroup writes per_thread KB-sized runs: store k of a wave covers 1 KB of whole lines
    size_t base = ((size_t)blockIdx.x * 256 * per_thread) + threadIdx.x;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (int k = 0; k < per_thread; ++k) {
        const size_t i = base + (size_t)k * 256;
        if (i < n16) dst[i] = v;
    }
}

__global__ __launch_bounds__(256) void expand_kernel(const uint2 *src, uint4 *dst, size_t n_src8)
{
    // 8 bytes read -> 64 bytes written (1:8), whole lines per store instruction
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_src8) return;
    const uint2 a = src[i];
    const size_t wbase = ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63u)) * 4 + (threadIdx.x & 63u);
    for (int k = 0; k < 4; ++k) dst[wbase + (size_t)k * 64] = make_uint4(a.x, a.y, a.x + k, a.y);
}

int main(int argc, char **argv)
{
    const size_t gb = argc > 1 ? (size_t)atoll(argv[1]) : 32;
    const size_t bytes = gb << 30, n16 = bytes / 16;
    uint4 *dst; uint2 *src;
    if (hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&src, bytes / 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, bytes / 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int per : {1, 4, 16}) {
        const size_t blocks = (n16 + 256ull * per - 1) / (256ull * per);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, dst, n16, per);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("fill %zu GiB, %2d x 16 B per thread: %.2f ms  %.2f TB/s\n", gb, per, ms, bytes / ms / 1e9);
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipMemsetAsync(dst, 0, bytes, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("hipMemsetAsync %zu GiB: %.2f ms  %.2f TB/s\n", gb, ms, bytes / ms / 1e9);
    }
    const size_t n8 = bytes / 64;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(expand_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, src, dst, n8);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("expand 1:8, %zu GiB written + %zu GiB read: %.2f ms  %.2f TB/s (read + write)\n", gb, gb / 8, ms, (bytes + bytes / 8) / ms / 1e9);
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
