// tools/micro_fp.hip -- which part of the footprint build bounds it?  The kernel of vr_bricks.hip on a
// 2048^3 UCHAR volume of arbitrary bytes, with parts switched off.  Built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_fp tools/micro_fp.hip && /tmp/micro_fp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kThreads = 256;
struct VV { const uint8_t *data; uint8_t *fp; int w, h, d; uint32_t fp_nbx, fp_nby, ystride; unsigned long long zstride; };

// MODE 0: full; 1: no loads (constant data); 2: loads, one store per thread; 3: no byte loads; 4: direct stores (no LDS)
template <int MODE, int ROWS>
__global__ __launch_bounds__(kThreads) void fp_kernel(VV vv, int nbz_fp)
{
    constexpr int C = 2;
    __shared__ uint4 s_t[kThreads * C];
    const uint8_t *p = vv.data;
    uint8_t *out = vv.fp;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int bx = (int)(blockIdx.x * (kThreads / 16) + (threadIdx.x >> 4));
    const int wave_bx0 = (int)(blockIdx.x * (kThreads / 16)) + wave * 4;
    const int ly = (int)(threadIdx.x & 3u), lz = (int)((threadIdx.x >> 2) & 3u);
    const int bz = (int)blockIdx.z;
    const int ez = bz * 4 + lz;
    uint4 *s_w = s_t + wave * 64 * C;
    for (int r = 0; r < ROWS; ++r) {
        const int by = (int)blockIdx.y * ROWS + r;
        if (by >= (int)vv.fp_nby) return;
        const int ey = by * 4 + ly;
        uint8_t o[32];
        const bool inside = bx >= 1 && bx * 4 + 3 <= vv.w - 1 && by >= 1 && by * 4 + 3 <= vv.h - 1 && bz >= 1 && bz * 4 + 3 <= vv.d - 1;
        if (inside && MODE != 1) {
#pragma unroll
            for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int y = ey - 1 + dy, z = ez - 1 + dz;
                    const unsigned long long row = (unsigned long long)(z >> 2) * vv.zstride + (unsigned long long)(y >> 2) * vv.ystride +
                                                   (unsigned long long)(((unsigned)bx << 6) + ((z & 3) << 4) + ((y & 3) << 2));
                    uint8_t c[4];
                    *reinterpret_cast<uint32_t *>(c) = *reinterpret_cast<const uint32_t *>(p + row);
                    const uint8_t left = MODE == 3 ? c[3] : p[row - 61ull];
                    const int j = 2 * dy + 4 * dz;
                    o[0 * 8 + j] = left; o[0 * 8 + j + 1] = c[0];
                    o[1 * 8 + j] = c[0]; o[1 * 8 + j + 1] = c[1];
                    o[2 * 8 + j] = c[1]; o[2 * 8 + j + 1] = c[2];
                    o[3 * 8 + j] = c[2]; o[3 * 8 + j + 1] = c[3];
                }
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) o[i] = (uint8_t)(i + lane + r);
        }
        const uint4 *src = reinterpret_cast<const uint4 *>(o);
        const unsigned long long brick0 = ((unsigned long long)bz * vv.fp_nby + (unsigned long long)by) * vv.fp_nbx + (unsigned long long)wave_bx0;
        uint4 *dst = reinterpret_cast<uint4 *>(out + brick0 * 512ull);
        if (MODE == 2) {
            if (wave_bx0 + lane / 16 < (int)vv.fp_nbx) dst[lane] = make_uint4(src[0].x ^ src[1].x, src[0].y ^ src[1].y, src[0].z ^ src[1].z, src[0].w ^ src[1].w);
            continue;
        }
        if (MODE == 4) {
            if (bx < (int)vv.fp_nbx) { dst[lane * 2] = src[0]; dst[lane * 2 + 1] = src[1]; }
            continue;
        }
#pragma unroll
        for (int k = 0; k < C; ++k) s_w[lane * C + k] = src[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int chunk = k * 64 + lane;
            if (wave_bx0 + chunk / (16 * C) < (int)vv.fp_nbx) dst[chunk] = s_w[chunk];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int MODE, int ROWS>
void run(const VV &vv, const char *what)
{
    const int nbz = (vv.d + 4) >> 2;
    dim3 grid((vv.fp_nbx + 15) / 16, (vv.fp_nby + ROWS - 1) / ROWS, nbz), block(kThreads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fp_kernel<MODE, ROWS>), grid, block, 0, 0, vv, nbz);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-58s rows/block %2d: %7.2f ms\n", what, ROWS, best);
}

int main()
{
    const int N = 2048;
    VV vv; vv.w = vv.h = vv.d = N; vv.fp_nbx = vv.fp_nby = (N + 4) / 4; vv.ystride = (N / 4) * 64; vv.zstride = (unsigned long long)(N / 4) * (N / 4) * 64;
    const size_t vol = (size_t)N * N * N, fpb = (size_t)vv.fp_nbx * vv.fp_nby * ((N + 4) / 4) * 512;
    uint8_t *v, *f;
    if (hipMalloc(&v, vol) != hipSuccess || hipMalloc(&f, fpb) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(v, 7, vol);
    vv.data = v; vv.fp = f;
    run<0, 4>(vv, "full");
    run<0, 1>(vv, "full");
    run<0, 16>(vv, "full");
    run<1, 4>(vv, "no loads (constant data), LDS transposition + stores");
    run<2, 4>(vv, "loads + packing, one 16-B store per thread");
    run<3, 4>(vv, "full without the byte loads of the left neighbour");
    run<4, 4>(vv, "full, direct stores (half lines per instruction)");
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
