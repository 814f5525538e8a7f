// vr_cells.hip -- the cell grid behind the two "skip work whose result is known" devices of the
// renderer (CellView, vr_internal.h): the path tracer's opacity bound and the ray caster's
// empty-cell bitmap.  Nothing here has a counterpart in the reference; both devices leave every
// pixel unchanged (tests: test_pathtrace_culling_is_exact, test_empty_skipping_is_exact).
#include "vr_device_math.h"
#include "vr_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kSparseLevels = 13;   // 2^12 = 4096 >= max transfer function entries

// (min, max) of the raw voxel values of every cell with its halo: voxels
// [(c << s) - 1, ((c + 1) << s) + 1] per axis, clamped to the volume.  One wave per cell; the
// lanes sweep the (2^s + 3)^2 voxels of a slice (gathers inside a handful of micro-bricks).
// NaN voxels (FLOAT volumes) mark the cell min > max: never culled, never empty.
template <typename VT>
__global__ __launch_bounds__(kThreads) void vr_cell_minmax_kernel(VolView vv, CellView grid,
                                                                  float2 *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const size_t n_cells = (size_t)grid.cx * grid.cy * grid.cz;
    const size_t cell = (size_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    const int cxi = (int)(cell % (size_t)grid.cx);
    const int cyi = (int)((cell / (size_t)grid.cx) % (size_t)grid.cy);
    const int czi = (int)(cell / ((size_t)grid.cx * grid.cy));
    const int n = (1 << grid.shift) + 3;
    const int x0 = (cxi << grid.shift) - 1, y0 = (cyi << grid.shift) - 1, z0 = (czi << grid.shift) - 1;
    const VT *p = (const VT *)vv.data;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    bool bad = false;
    // the in-slice part of a lane's addresses does not depend on z: kPerLane voxels of a slice per
    // lane and pass (one pass up to shift 4), their (x, y) offsets computed once per pass
    constexpr int kPerLane = 6;
    for (int base = 0; base < n * n; base += 64 * kPerLane) {
        unsigned long long xy[kPerLane];
        bool have[kPerLane];
#pragma unroll
        for (int j = 0; j < kPerLane; ++j) {
            const int i = base + (int)lane + 64 * j;
            have[j] = i < n * n;
            const int dy = have[j] ? i / n : 0, dx = have[j] ? i - dy * n : 0;
            const int x = min(max(x0 + dx, 0), vv.w - 1), y = min(max(y0 + dy, 0), vv.h - 1);
            xy[j] = vr_voxel_index(vv, x, y, 0);
        }
        for (int dz = 0; dz < n; ++dz) {
            const int z = min(max(z0 + dz, 0), vv.d - 1);
            const unsigned long long zo = vr_voxel_index(vv, 0, 0, z);
#pragma unroll
            for (int j = 0; j < kPerLane; ++j) {
                if (have[j]) {
                    const float v = (float)p[zo + xy[j]];
                    bad = bad || !(v == v);
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float omn = __shfl_down(mn, off, 64), omx = __shfl_down(mx, off, 64);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
    }
    const bool any_bad = __ballot(bad) != 0ull;
    if (lane == 0) out[cell] = any_bad ? make_float2(1.f, 0.f) : make_float2(mn, mx);
}

// sparse table of the TF opacity for O(1) range maxima: T[j][i] = max(alpha[i .. i + 2^j - 1])
__global__ __launch_bounds__(kThreads) void vr_cell_sparse_kernel(TfView tf, float *T)
{
    const int n = (int)tf.tff_n;
    for (int i = threadIdx.x; i < n; i += kThreads) T[i] = tf.tff[i].w;
    for (int j = 1; j < kSparseLevels; ++j) {
        __syncthreads();
        const float *prev = T + (size_t)(j - 1) * n;
        float *cur = T + (size_t)j * n;
        const int half = 1 << (j - 1);
        for (int i = threadIdx.x; i < n; i += kThreads) {
            const float a = prev[i], b = prev[min(i + half, n - 1)];
            cur[i] = a < b ? b : a;
        }
    }
}

// Opacity bound and empty bit of every cell.  A trilinear fetch returns a value within
// [min, max] of the voxels it reads up to a few ulps; the TF lookup (tff_linear) interpolates two
// adjacent entries, so its opacity is at most the larger of them up to an ulp, and exactly 0 when
// both are 0.  One extra table entry on either side of the index range and a relative margin of
// 1e-6 (8 ulps) on the bound cover the roundings.
__global__ __launch_bounds__(kThreads) void vr_cell_bounds_kernel(const float2 *minmax,
                                                                  size_t n_cells, float inv_max,
                                                                  int n, const float *T,
                                                                  float *bound, uint32_t *empty)
{
    const size_t c = (size_t)blockIdx.x * kThreads + threadIdx.x;
    float b = 2.0f;   // above every threshold: never cull
    if (c < n_cells) {
        const float2 mm = minmax[c];
        if (mm.x <= mm.y) {
            const float fn = (float)n;
            float flo = floorf((mm.x * inv_max) * fn - 0.5f) - 1.0f;
            float fhi = floorf((mm.y * inv_max) * fn - 0.5f) + 2.0f;
            flo = vclamp(flo, 0.0f, fn - 1.0f);
            fhi = vclamp(fhi, 0.0f, fn - 1.0f);
            const int lo = (int)flo, hi = (int)fhi;
            if (lo <= hi) {
                const int k = 31 - __clz(hi - lo + 1);
                const float a = T[(size_t)k * n + lo], d = T[(size_t)k * n + (hi - (1 << k) + 1)];
                b = (a < d ? d : a) * 1.000001f;
            }
        }
        if (bound) bound[c] = b;
    }
    const unsigned long long m = __ballot(c < n_cells && b == 0.0f);
    if (empty && (threadIdx.x & 63) == 0 && c < n_cells) {
        const size_t w = c >> 5;   // c is a multiple of 64
        empty[w] = (uint32_t)m;
        if (c + 32 < ((n_cells + 31) & ~(size_t)31)) empty[w + 1] = (uint32_t)(m >> 32);
    }
}

// CellView::bmask: one thread per ESS brick.  Sub-blocks that lie outside the volume are never
// looked up (sample coordinates are clamped into the volume) and read 0.
__global__ __launch_bounds__(kThreads) void vr_cell_bmask_kernel(VolView vv, CellView g, int bw, int bh,
                                                                 int bd, unsigned long long *out)
{
    const size_t n = (size_t)bw * bh * bd;
    const size_t b = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= n) return;
    const int bx = (int)(b % (size_t)bw), by = (int)((b / (size_t)bw) % (size_t)bh), bz = (int)(b / ((size_t)bw * bh));
    const int sx = g.bex - 2, sy = g.bey - 2, sz = g.bez - 2;   // log2 of the sub-block edge
    unsigned long long word = 0;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i) {
                const int x0 = (bx << g.bex) + (i << sx), y0 = (by << g.bey) + (j << sy), z0 = (bz << g.bez) + (k << sz);
                if (x0 >= vv.w || y0 >= vv.h || z0 >= vv.d) continue;
                const int x1 = min(x0 + (1 << sx), vv.w) - 1, y1 = min(y0 + (1 << sy), vv.h) - 1;
                const int z1 = min(z0 + (1 << sz), vv.d) - 1;
                bool all = true;
                for (int cz = z0 >> g.shift; cz <= (z1 >> g.shift); ++cz)
                    for (int cy = y0 >> g.shift; cy <= (y1 >> g.shift); ++cy)
                        for (int cx = x0 >> g.shift; cx <= (x1 >> g.shift); ++cx) {
                            const uint32_t idx = ((uint32_t)cz * (uint32_t)g.cy + (uint32_t)cy) * (uint32_t)g.cx + (uint32_t)cx;
                            all = all && ((g.empty[idx >> 5] >> (idx & 31u)) & 1u);
                        }
                if (all) word |= 1ull << (i + 4 * j + 16 * k);
            }
    out[b] = word;
}

} // namespace

hipError_t vr_launch_cell_bmask(const VolView &vol, const CellView &grid, int bw, int bh, int bd,
                                unsigned long long *bmask, hipStream_t stream)
{
    const size_t n = (size_t)bw * bh * bd;
    hipLaunchKernelGGL(vr_cell_bmask_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       stream, vol, grid, bw, bh, bd, bmask);
    return hipGetLastError();
}

hipError_t vr_launch_cell_minmax(const VolView &vol, int format, const CellView &grid,
                                 float2 *minmax, hipStream_t stream)
{
    const size_t n_cells = (size_t)grid.cx * grid.cy * grid.cz;
    const size_t per_block = kThreads / 64;
    dim3 g((unsigned)((n_cells + per_block - 1) / per_block)), block(kThreads);
    switch (format) {
    case VRHIP_UCHAR:
        hipLaunchKernelGGL(vr_cell_minmax_kernel<uint8_t>, g, block, 0, stream, vol, grid, minmax);
        break;
    case VRHIP_USHORT:
        hipLaunchKernelGGL(vr_cell_minmax_kernel<uint16_t>, g, block, 0, stream, vol, grid, minmax);
        break;
    case VRHIP_FLOAT:
        hipLaunchKernelGGL(vr_cell_minmax_kernel<float>, g, block, 0, stream, vol, grid, minmax);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t vr_launch_cell_bounds(const float2 *minmax, const CellView &grid, float inv_max,
                                 const TfView &tf, float *sparse_scratch, float *bound,
                                 uint32_t *empty_bits, hipStream_t stream)
{
    hipLaunchKernelGGL(vr_cell_sparse_kernel, dim3(1), dim3(kThreads), 0, stream, tf, sparse_scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t n_cells = (size_t)grid.cx * grid.cy * grid.cz;
    dim3 g((unsigned)((n_cells + kThreads - 1) / kThreads)), block(kThreads);
    hipLaunchKernelGGL(vr_cell_bounds_kernel, g, block, 0, stream, minmax, n_cells, inv_max,
                       (int)tf.tff_n, (const float *)sparse_scratch, bound, empty_bits);
    return hipGetLastError();
}
