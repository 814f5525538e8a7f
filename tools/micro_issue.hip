// tools/micro_issue.hip -- lone-wave issue cost of instruction patterns on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/micro_issue.hip -o gpurun_out/micro_issue && ./micro_issue
// One wave per SIMD (256 blocks x 256 threads); each kernel runs N iterations of a pattern and
// reports cycles per iteration (s_memtime), median over waves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define N_ITER 4096

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// A: 16 dependent v_fma per iteration
__global__ void k_fma(float *out, unsigned long long *cyc, float a, float b)
{
    float x = threadIdx.x * 1e-3f;
    unsigned long long t0 = now();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fmaf(x, a, b);
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// B: 8 x (v_cmp -> v_cndmask) dependent chain per iteration (mask through VCC/SGPR, no SALU)
__global__ void k_cmpsel(float *out, unsigned long long *cyc, float a, float b)
{
    float x = threadIdx.x * 1e-3f, y = a;
    unsigned long long t0 = now();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = (x < y) ? x + b : x - b;   // v_cmp + 2 arith + v_cndmask
            asm volatile("" : "+v"(x));
        }
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// C: masks combined on the SALU: 4 x (2 v_cmp -> s_and -> v_cndmask) per iteration
__global__ void k_cmpand(float *out, unsigned long long *cyc, float a, float b)
{
    float x = threadIdx.x * 1e-3f, y = a, z = a * 0.5f;
    unsigned long long t0 = now();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool m = (x <= y) && (x <= z);
            x = m ? x + b : x - b;
            asm volatile("" : "+v"(x));
        }
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// D: wave ballot + scalar branch per iteration (v_cmp -> s_cbranch), 1 fma body
__global__ void k_ballot(float *out, unsigned long long *cyc, float a, float b)
{
    float x = threadIdx.x * 1e-3f;
    int n = 0;
    unsigned long long t0 = now();
    for (int i = 0; i < N_ITER; ++i) {
        if (__ballot(x < a)) { x = __builtin_fmaf(x, 1.0001f, b); ++n; }
        if (__ballot(x > 1e30f)) break;
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + n;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// E: dependent LDS read per iteration (index depends on the previous value)
__global__ void k_lds(float *out, unsigned long long *cyc, float a, float b)
{
    __shared__ unsigned s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (i * 2654435761u) >> 20;
    __syncthreads();
    unsigned idx = threadIdx.x;
    unsigned long long t0 = now();
    for (int i = 0; i < N_ITER; ++i) idx = s[idx & 4095];
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)idx + a + b;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// F: integer ops: v_mad_u64_u32 + v_mul_lo_u32 chain (4 each per iteration)
__global__ void k_imul(float *out, unsigned long long *cyc, float a, float b)
{
    unsigned long long x = threadIdx.x + 1;
    unsigned y = threadIdx.x + 3;
    unsigned long long t0 = now();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x = (unsigned long long)(unsigned)x * 0x9E3779B1ull + y;
            y = y * 0x85EBCA6Bu + (unsigned)x;
        }
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(x + y) + a + b;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename K>
void run(const char *name, K kernel, int per_iter, int blocks)
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * 4 * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.5f, 1e-3f);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2] / N_ITER;
    printf("%-10s blocks=%4d  %8.1f cycles/iteration  (%d ops/iter -> %.1f cycles/op)\n", name, blocks,
           med, per_iter, med / per_iter);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int blocks : {256, 1024}) {
        run("fma16", k_fma, 16, blocks);
        run("cmpsel8", k_cmpsel, 8 * 4, blocks);
        run("cmpand4", k_cmpand, 4 * 6, blocks);
        run("ballot2", k_ballot, 6, blocks);
        run("lds_dep", k_lds, 2, blocks);
        run("imul8", k_imul, 8, blocks);
    }
    return 0;
}
