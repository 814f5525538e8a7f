// tools/micro_rate.hip -- SIMD time per instruction (throughput with 8 independent chains per lane)
// of the instruction kinds the ray-cast kernels are made of, at 1 and 2 waves per SIMD on gfx950.
// Prints cycles per wave-instruction as seen by one wave and the implied SIMD cycles per instruction.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define N_ITER 2048
__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define KERNEL(name, DECL, BODY)                                                                   \
    __global__ void name(unsigned long long *out, unsigned long long *cyc, unsigned a, unsigned b) \
    {                                                                                              \
        DECL;                                                                                      \
        unsigned long long t0 = now();                                                             \
        for (int i = 0; i < N_ITER; ++i) {                                                         \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { BODY; }                                \
        }                                                                                          \
        unsigned long long t1 = now();                                                             \
        unsigned long long acc = 0;                                                                \
        for (int j = 0; j < 8; ++j) acc += (unsigned long long)x[j];                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                          \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }
KERNEL(k_fadd, float x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j; float fa = __uint_as_float(a),
       x[j] = x[j] + fa; asm volatile("" : "+v"(x[j])))
KERNEL(k_u64add, unsigned long long x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 7ull + j; unsigned long long la = ((unsigned long long)a << 20) | b,
       x[j] = x[j] + la; asm volatile("" : "+v"(x[j])))
KERNEL(k_mad64, unsigned long long x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j; ,
       x[j] = (unsigned long long)(unsigned)x[j] * (unsigned long long)a + x[j]; asm volatile("" : "+v"(x[j])))
KERNEL(k_mullo, unsigned x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j; ,
       x[j] = x[j] * a + b; asm volatile("" : "+v"(x[j])))
KERNEL(k_mad24, unsigned x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j; ,
       x[j] = __umul24(x[j], a) + b; asm volatile("" : "+v"(x[j])))
KERNEL(k_dpp, int x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j; ,
       x[j] = __builtin_amdgcn_update_dpp(0, x[j], 0x55, 0xf, 0xf, true); asm volatile("" : "+v"(x[j])))
KERNEL(k_cvt, float x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j; ,
       x[j] = (float)(int)floorf(x[j]) + 0.5f; asm volatile("" : "+v"(x[j])))
KERNEL(k_div, float x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j + 1; float fa = __uint_as_float(a),
       x[j] = x[j] / fa; asm volatile("" : "+v"(x[j])))
KERNEL(k_sel, float x[8]; for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j; float fa = __uint_as_float(a),
       x[j] = x[j] < fa ? x[j] + 1.f : x[j] - 1.f; asm volatile("" : "+v"(x[j])))

typedef float f2v __attribute__((ext_vector_type(2)));
#define KERNEL2(name, BODY)                                                                        \
    __global__ void name(unsigned long long *out, unsigned long long *cyc, unsigned a, unsigned b) \
    {                                                                                              \
        f2v x[8];                                                                                  \
        for (int j = 0; j < 8; ++j) x[j] = (f2v){(float)threadIdx.x + j, (float)j};                \
        const f2v fa = (f2v){__uint_as_float(a), 0.5f}, fb = (f2v){0.25f, (float)b};               \
        unsigned long long t0 = now();                                                             \
        for (int i = 0; i < N_ITER; ++i) {                                                         \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { BODY; asm volatile("" : "+v"(x[j])); } \
        }                                                                                          \
        unsigned long long t1 = now();                                                             \
        float acc = 0;                                                                             \
        for (int j = 0; j < 8; ++j) acc += x[j].x + x[j].y;                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned long long)acc;                      \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }
KERNEL2(k_pkadd, x[j] = x[j] + fa)
KERNEL2(k_pkmul, x[j] = x[j] * fa)
KERNEL2(k_pkfma, x[j] = __builtin_elementwise_fma(x[j], fa, fb))

template <typename K> void run(const char *name, K k, int instr_per_body, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu, threads = 256;
    unsigned long long *out, *cyc;
    hipMalloc(&out, sizeof(unsigned long long) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 0x3f800000u, 3u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[h.size() / 2] / (double)N_ITER / 8.0 / instr_per_body;
    printf("%-10s waves/SIMD=%d  cycles per wave-instruction (one wave's view) %.2f   SIMD cycles per instruction %.2f\n",
           name, blocks_per_cu, per, per / blocks_per_cu);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int b = 1; b <= 2; ++b) {
        run("fadd", k_fadd, 1, b);
        run("u64 add", k_u64add, 1, b);
        run("mad_u64_u32", k_mad64, 1, b);
        run("mul_lo+add", k_mullo, 1, b);
        run("mad_u24", k_mad24, 1, b);
        run("mov_dpp", k_dpp, 1, b);
        run("floor+cvt2+add", k_cvt, 4, b);
        run("fdiv (ieee)", k_div, 1, b);
        run("cmp+2arith+sel", k_sel, 4, b);
        run("pk_add_f32", k_pkadd, 1, b);
        run("pk_mul_f32", k_pkmul, 1, b);
        run("pk_fma_f32", k_pkfma, 1, b);
    }
    return 0;
}
