// tools/micro_occ.hip -- what does a wave64 VALU instruction cost on a gfx950 SIMD, and at which clock?
//
// For v_fma_f32, v_add_f32 and the packed forms v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 operations
// per lane and instruction), 8 independent chains per lane, W = 1..8 waves per SIMD on EVERY CU (launch: 256 CUs x
// W blocks of 256 threads -- an all-CU burn, so the clock is what the part sustains under load), plus one
// dependent chain (what a wave with no instruction-level parallelism sees):
//   * wall time by HIP events -> wave-instructions per second for the whole chip;
//   * inside the kernel, around the loop: s_memtime (the shader-clock counter, clock64()) and s_memrealtime (the
//     constant 100 MHz counter, wall_clock64()) -> the shader clock the loop actually ran at; the SIMD cycles one
//     wave-instruction occupies = 1024 SIMDs x that clock / the chip-wide rate.
// bench.py's roofline_valu_issue.peak is taken from this output (profiles/r4/micro_occ.txt): a wave64 VALU
// instruction -- plain or packed -- occupies a 16-lane SIMD for four cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 4.
// Build on the box: hipcc --offload-arch=gfx950 -O3 tools/micro_occ.hip -o /tmp/micro_occ
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define N_ITER 20000
typedef float f2v __attribute__((ext_vector_type(2)));

struct Clocks { unsigned long long shader, real; };
__device__ __forceinline__ Clocks now()
{
    Clocks c;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(c.shader), "=s"(c.real)::"memory");
    return c;
}
#define TIMED_TAIL(ACC)                                                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (ACC);                                                     \
    if ((threadIdx.x & 63) == 0) {                                                                          \
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                             \
        clk[2 * w] = t1.shader - t0.shader;                                                                 \
        clk[2 * w + 1] = t1.real - t0.real;                                                                 \
    }

__global__ __launch_bounds__(256) void k_fma(float *out, unsigned long long *clk, float a, float b)
{
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
    const Clocks t0 = now();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = __builtin_fmaf(x[j], a, b); asm volatile("" : "+v"(x[j])); }
    }
    const Clocks t1 = now();
    float acc = 0;
    for (int j = 0; j < 8; ++j) acc += x[j];
    TIMED_TAIL(acc)
}
__global__ __launch_bounds__(256) void k_add(float *out, unsigned long long *clk, float a, float b)
{
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j + b;
    const Clocks t0 = now();
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = x[j] + a; asm volatile("" : "+v"(x[j])); }
    }
    const Clocks t1 = now();
    float acc = 0;
    for (int j = 0; j < 8; ++j) acc += x[j];
    TIMED_TAIL(acc)
}
// one dependent chain: what a wave with no instruction-level parallelism sees
__global__ __launch_bounds__(256) void k_dep(float *out, unsigned long long *clk, float a, float b)
{
    float x = threadIdx.x;
    const Clocks t0 = now();
    for (int i = 0; i < N_ITER * 8; ++i) { x = __builtin_fmaf(x, a, b); asm volatile("" : "+v"(x)); }
    const Clocks t1 = now();
    TIMED_TAIL(x)
}
#define PK_KERNEL(name, BODY)                                                                               \
    __global__ __launch_bounds__(256) void name(float *out, unsigned long long *clk, float a, float b)      \
    {                                                                                                       \
        f2v x[8];                                                                                           \
        for (int j = 0; j < 8; ++j) x[j] = (f2v){(float)threadIdx.x + j, (float)j};                         \
        const f2v fa = (f2v){a, a}, fb = (f2v){b, b};                                                       \
        const Clocks t0 = now();                                                                            \
        for (int i = 0; i < N_ITER; ++i) {                                                                  \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { BODY; asm volatile("" : "+v"(x[j])); }          \
        }                                                                                                   \
        const Clocks t1 = now();                                                                            \
        float acc = fb.x * 0.f;                                                                             \
        for (int j = 0; j < 8; ++j) acc += x[j].x + x[j].y;                                                 \
        TIMED_TAIL(acc)                                                                                     \
    }
PK_KERNEL(k_pkfma, x[j] = __builtin_elementwise_fma(x[j], fa, fb))
PK_KERNEL(k_pkmul, x[j] = x[j] * fa)
PK_KERNEL(k_pkadd, x[j] = x[j] + fa)

typedef void (*kern_t)(float *, unsigned long long *, float, float);

int main()
{
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&clk, 256 * 8 * 4 * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { const char *name; kern_t k; int flops_per_lane; } kinds[] = {
        {"v_fma_f32, 8 independent chains", k_fma, 2},        {"v_add_f32, 8 independent chains", k_add, 1},
        {"v_pk_fma_f32, 8 independent chains", k_pkfma, 4},   {"v_pk_mul_f32, 8 independent chains", k_pkmul, 2},
        {"v_pk_add_f32, 8 independent chains", k_pkadd, 2},   {"v_fma_f32, 1 dependent chain", k_dep, 2},
    };
    for (auto &kd : kinds)
        for (int w = 1; w <= 8; ++w) {
            const int blocks = 256 * w;   // one block of 256 threads = one wave on each SIMD of a CU
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kd.k, dim3(blocks), dim3(256), 0, 0, out, clk, 0.999f, 0.001f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(size_t(blocks) * 4 * 2);
            hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::vector<double> sh, mhz;
            for (size_t i = 0; i < h.size() / 2; ++i) {
                sh.push_back((double)h[2 * i]);
                mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);   // s_memrealtime ticks at 100 MHz
            }
            std::sort(sh.begin(), sh.end());
            std::sort(mhz.begin(), mhz.end());
            const double per_wave = (double)N_ITER * 8;                        // instructions one wave issued
            const double insts = 256.0 * w * 4 * per_wave;                     // wave-instructions, whole chip
            const double rate = insts / (ms * 1e-3);
            const double clock_mhz = mhz[mhz.size() / 2];
            // SIMD cycles one wave-instruction occupies, from the chip-wide rate and the clock the waves measured
            // (the waves of a launch do not all run at once -- the dispatcher does not spread W blocks per CU
            // evenly -- so a single wave's own cycle count divided by W would flatter the SIMD)
            const double simd_cycles = 1024.0 * clock_mhz * 1e6 / rate;
            printf("%-36s %d wave(s)/SIMD: %7.3f ms  %.3e wave-instr/s  %6.2f TFLOP/s  shader clock %6.0f MHz  "
                   "SIMD cycles per wave-instruction %.2f  (a wave's own view: %.2f cycles between its instructions)\n",
                   kd.name, w, ms, rate, rate * 64.0 * kd.flops_per_lane / 1e12, clock_mhz, simd_cycles,
                   sh[sh.size() / 2] / per_wave);
        }
    return 0;
}
