// tools/micro_occ.hip -- VALU issue rate of a SIMD against the number of resident waves (gfx950):
// 8 independent v_fma_f32 chains per lane, W waves per SIMD (launch: 256 CUs x W blocks of 256 threads),
// wall time by HIP events -> wave-instructions per second for the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 20000
__global__ __launch_bounds__(256) void k_fma(float *out, float a, float b)
{
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = __builtin_fmaf(x[j], a, b); asm volatile("" : "+v"(x[j])); }
    }
    float acc = 0;
    for (int j = 0; j < 8; ++j) acc += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// the same with a dependent chain of 1 (latency-bound per wave): what a wave with little ILP sees
__global__ __launch_bounds__(256) void k_dep(float *out, float a, float b)
{
    float x = threadIdx.x;
    for (int i = 0; i < N_ITER * 8; ++i) { x = __builtin_fmaf(x, a, b); asm volatile("" : "+v"(x)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dep = 0; dep < 2; ++dep)
        for (int w = 1; w <= 8; ++w) {
            // one block of 256 threads = 1 wave per SIMD of a CU
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (dep) hipLaunchKernelGGL(k_dep, dim3(256 * w), dim3(256), 0, 0, out, 0.999f, 0.001f);
                else hipLaunchKernelGGL(k_fma, dim3(256 * w), dim3(256), 0, 0, out, 0.999f, 0.001f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = 256.0 * w * 4 * (double)N_ITER * 8;   // wave-instructions
            printf("%s chains, %d wave(s)/SIMD: %.3f ms, %.3e wave-instr/s (%.2f of 1.2288e12), %.2f cycles/instr/SIMD at 2.4 GHz\n",
                   dep ? "1 dependent" : "8 independent", w, ms, insts / (ms * 1e-3), insts / (ms * 1e-3) / 1.2288e12,
                   2.4e9 / (insts / (ms * 1e-3) / 1024.0));
        }
    return 0;
}
