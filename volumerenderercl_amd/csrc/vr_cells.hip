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

// ---- the same (min, max) by separable streaming passes (cells of 8 or 16 voxels: volumes up to 4096^3)
//
// The extrema over the box [(c << s) - 1, ((c + 1) << s) + 1]^3 separate by axis.  Pass XY reads every
// voxel as part of whole 64-byte micro-brick lines: a workgroup takes one row of cells (cy) in one
// slice of micro-bricks (mz) and a thread a micro-brick column.  Over the rows y of the cell's halo'd
// range it reduces, per z slice of the bricks, three x classes of its four voxel columns -- A: all
// four, L: the first two (what the cell on the left takes from this brick), H: the last one (what the
// cell on the right takes) -- into LDS; then a thread per cell combines H of the brick to its left, A
// of its own bricks and L of the brick to its right: the extrema over the x and y ranges of the cell
// for that voxel slice z.  Pass Z reduces the 2^s + 3 slices of each cell.  Exactly the kernel
// above's values (NaN voxels give (-inf, +inf): never culled, never empty).
template <typename VT> struct CellAcc {   // extrema of raw voxel values; integer voxels stay integers until the end
    uint32_t mn = 0xffffffffu, mx = 0u;
    VR_DEV void add(VT v) { mn = min(mn, (uint32_t)v); mx = max(mx, (uint32_t)v); }
    VR_DEV float2 get() const { return mn > mx ? make_float2(__builtin_inff(), -__builtin_inff()) : make_float2((float)mn, (float)mx); }
};
template <> struct CellAcc<float> {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    VR_DEV void add(float f)
    {
        if (!(f == f)) { mn = -__builtin_inff(); mx = __builtin_inff(); }
        mn = f < mn ? f : mn;
        mx = f > mx ? f : mx;
    }
    VR_DEV float2 get() const { return make_float2(mn, mx); }
};
struct CellMerge {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    VR_DEV void merge(float2 q) { mn = q.x < mn ? q.x : mn; mx = q.y > mx ? q.y : mx; }
};

template <typename VT>
__global__ __launch_bounds__(kThreads) void vr_cell_xy_kernel(VolView vv, CellView grid, float2 *xy /* [d][cy][cx] */)
{
    extern __shared__ float2 s_cls[];   // [nbx][4 z slices][3 classes]
    const int cyi = blockIdx.x % grid.cy, mz = blockIdx.x / grid.cy;
    const int s = grid.shift, E = 1 << s, bpc = E >> 2;
    const VT *base = (const VT *)vv.data;
    const int y_a = max((cyi << s) - 1, 0), y_b = min(((cyi + 1) << s) + 1, vv.h - 1);   // halo'd rows, clipped
    constexpr int N16 = 64 * (int)sizeof(VT) / 16;
    for (int mx = threadIdx.x; mx < (int)vv.nbx; mx += kThreads) {
        CellAcc<VT> acc[4][3];
        for (int my = y_a >> 2; my <= (y_b >> 2); ++my) {
            const uint4 *p = reinterpret_cast<const uint4 *>(base + (unsigned long long)mz * vv.zstride +
                                                             (unsigned long long)my * vv.ystride +
                                                             (unsigned long long)mx * 64ull);
            uint4 q[N16];
#pragma unroll
            for (int i = 0; i < N16; ++i) q[i] = p[i];
            VT v[64];
            __builtin_memcpy(v, q, sizeof v);
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const int y = 4 * my + dy;
                if (y < y_a || y > y_b) continue;   // (uniform over the workgroup)
#pragma unroll
                for (int dz = 0; dz < 4; ++dz) {
                    const VT *r = v + dz * 16 + dy * 4;
                    // columns beyond the volume's width hold padding: never part of a class
                    const int xs = 4 * mx;
                    if (xs + 3 < vv.w) {
                        acc[dz][0].add(r[0]); acc[dz][0].add(r[1]); acc[dz][0].add(r[2]); acc[dz][0].add(r[3]);
                        acc[dz][1].add(r[0]); acc[dz][1].add(r[1]);
                        acc[dz][2].add(r[3]);
                    } else {
                        for (int dx = 0; dx < 4; ++dx)
                            if (xs + dx < vv.w) {
                                acc[dz][0].add(r[dx]);
                                if (dx < 2) acc[dz][1].add(r[dx]);
                                if (dx == 3) acc[dz][2].add(r[dx]);
                            }
                    }
                }
            }
        }
#pragma unroll
        for (int dz = 0; dz < 4; ++dz)
#pragma unroll
            for (int k = 0; k < 3; ++k) s_cls[(mx * 4 + dz) * 3 + k] = acc[dz][k].get();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < grid.cx * 4; i += kThreads) {
        const int cxi = i >> 2, dz = i & 3;
        const int z = 4 * mz + dz;
        if (z >= vv.d) continue;
        CellMerge a;
        const int b0 = cxi * bpc;
        if (b0 - 1 >= 0) a.merge(s_cls[((b0 - 1) * 4 + dz) * 3 + 2]);
        for (int b = b0; b < b0 + bpc && b < (int)vv.nbx; ++b) a.merge(s_cls[(b * 4 + dz) * 3 + 0]);
        if (b0 + bpc < (int)vv.nbx) a.merge(s_cls[((b0 + bpc) * 4 + dz) * 3 + 1]);
        xy[((size_t)z * grid.cy + cyi) * grid.cx + cxi] = make_float2(a.mn, a.mx);
    }
}

__global__ __launch_bounds__(kThreads) void vr_cell_z_kernel(CellView grid, int d, const float2 *xy, float2 *out)
{
    const size_t n_cells = (size_t)grid.cx * grid.cy * grid.cz;
    const size_t c = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (c >= n_cells) return;
    const size_t plane = (size_t)grid.cx * grid.cy;
    const int czi = (int)(c / plane);
    const size_t xyi = c % plane;
    const int z_a = max((czi << grid.shift) - 1, 0), z_b = min(((czi + 1) << grid.shift) + 1, d - 1);
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int z = z_a; z <= z_b; ++z) {
        const float2 q = xy[(size_t)z * plane + xyi];
        mn = q.x < mn ? q.x : mn;
        mx = q.y > mx ? q.y : mx;
    }
    out[c] = make_float2(mn, mx);
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

// Coarse (min, max) from the fine grid's: a coarse cell's halo'd extent [E c - 1, E (c + 1) + 1] per
// axis is the union of the extents [e f - 1, e (f + 1) + 1] of the fine cells f inside it.
__global__ __launch_bounds__(kThreads) void vr_cell_reduce_kernel(const float2 *fine, CellView g, float2 *out)
{
    const size_t n = (size_t)g.cx * g.cy * g.cz;
    const size_t c = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (c >= n) return;
    const int cxi = (int)(c % (size_t)g.cx), cyi = (int)((c / (size_t)g.cx) % (size_t)g.cy);
    const int czi = (int)(c / ((size_t)g.cx * g.cy));
    const int ds = g.shift - g.eshift, m = 1 << ds;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int k = czi << ds; k < min((czi << ds) + m, g.ecz); ++k)
        for (int j = cyi << ds; j < min((cyi << ds) + m, g.ecy); ++j)
            for (int i = cxi << ds; i < min((cxi << ds) + m, g.ecx); ++i) {
                const float2 q = fine[((size_t)k * g.ecy + j) * g.ecx + i];
                mn = q.x < mn ? q.x : mn;
                mx = q.y > mx ? q.y : mx;
            }
    out[c] = make_float2(mn, mx);
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
                for (int cz = z0 >> g.eshift; cz <= (z1 >> g.eshift); ++cz)
                    for (int cy = y0 >> g.eshift; cy <= (y1 >> g.eshift); ++cy)
                        for (int cx = x0 >> g.eshift; cx <= (x1 >> g.eshift); ++cx) {
                            const uint32_t idx = ((uint32_t)cz * (uint32_t)g.ecy + (uint32_t)cy) * (uint32_t)g.ecx + (uint32_t)cx;
                            all = all && ((g.empty[idx >> 5] >> (idx & 31u)) & 1u);
                        }
                if (all) word |= 1ull << (i + 4 * j + 16 * k);
            }
    out[b] = word;
}

} // namespace

hipError_t vr_launch_cell_reduce(const float2 *fine, const CellView &grid, float2 *coarse, hipStream_t stream)
{
    const size_t n = (size_t)grid.cx * grid.cy * grid.cz;
    hipLaunchKernelGGL(vr_cell_reduce_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       stream, fine, grid, coarse);
    return hipGetLastError();
}

hipError_t vr_launch_cell_bmask(const VolView &vol, const CellView &grid, int bw, int bh, int bd,
                                unsigned long long *bmask, hipStream_t stream)
{
    const size_t n = (size_t)bw * bh * bd;
    hipLaunchKernelGGL(vr_cell_bmask_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       stream, vol, grid, bw, bh, bd, bmask);
    return hipGetLastError();
}

hipError_t vr_launch_cell_minmax(const VolView &vol, int format, const CellView &grid,
                                 float2 *minmax, hipStream_t stream, float2 *records)
{
    const size_t n_cells = (size_t)grid.cx * grid.cy * grid.cz;
    if (records && grid.shift <= 4) {
        // separable streaming build (records: d * cy * cx float2 of scratch)
        const size_t lds = (size_t)vol.nbx * 4 * 3 * sizeof(float2);
        dim3 g1((unsigned)(grid.cy * (int)vol.nbz)), block(kThreads);
        int nb = 0;
        hipError_t e;
        switch (format) {
        case VRHIP_UCHAR:
            e = vr_prepare_kernel(vr_cell_xy_kernel<uint8_t>, kThreads, lds, &nb, "cell grid xy", 0);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(vr_cell_xy_kernel<uint8_t>, g1, block, lds, stream, vol, grid, records);
            break;
        case VRHIP_USHORT:
            e = vr_prepare_kernel(vr_cell_xy_kernel<uint16_t>, kThreads, lds, &nb, "cell grid xy", 0);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(vr_cell_xy_kernel<uint16_t>, g1, block, lds, stream, vol, grid, records);
            break;
        case VRHIP_FLOAT:
            e = vr_prepare_kernel(vr_cell_xy_kernel<float>, kThreads, lds, &nb, "cell grid xy", 0);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(vr_cell_xy_kernel<float>, g1, block, lds, stream, vol, grid, records);
            break;
        default: return hipErrorInvalidValue;
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(vr_cell_z_kernel, dim3((unsigned)((n_cells + kThreads - 1) / kThreads)), block, 0, stream,
                           grid, vol.d, (const float2 *)records, minmax);
        return hipGetLastError();
    }
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
