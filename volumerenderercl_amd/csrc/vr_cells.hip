// vr_cells.hip -- the cell grid behind the two "skip work whose result is known" devices of the
// renderer (CellView, vr_internal.h): the path tracer's opacity bound and the ray caster's
// empty-cell bitmap.  Nothing here has a counterpart in the reference; both devices leave every
// pixel unchanged (tests: test_pathtrace_culling_is_exact, test_empty_skipping_is_exact).
#include <algorithm>
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
                                                                  float2 *out, size_t cell0)
{
    const uint32_t lane = threadIdx.x & 63u;
    const size_t n_cells = (size_t)grid.cx * grid.cy * grid.cz;
    const size_t cell = cell0 + (size_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
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

// ---- the same (min, max) by separable streaming passes (cells of 4, 8 or 16 voxels)
//
// The extrema over the box [(c << s) - 1, ((c + 1) << s) + 1]^3 separate by axis.  Pass XY
// (vr_cell_xy_kernel, below) reads every voxel as part of whole micro-brick lines and leaves, per
// voxel slice z and cell column (cx, cy), the extrema over the cell's x and y ranges: a record in the
// voxel type.  Pass Z (vr_cell_z_kernel) reduces the 2^s + 3 slices of each cell.  Exactly the kernel
// above's values (NaN voxels give (-inf, +inf): never culled, never empty).
// A record of the passes: extrema in the voxel type, mn > mx = nothing yet; FLOAT volumes flag NaN as (-inf, +inf).
template <typename VT> struct CellRec { VT mn, mx; };
template <typename VT> struct CellAcc {   // extrema of raw voxel values; integer voxels stay integers until the end
    uint32_t mn = 0xffffffffu, mx = 0u;
    VR_DEV void add(VT v) { mn = min(mn, (uint32_t)v); mx = max(mx, (uint32_t)v); }
    VR_DEV CellRec<VT> rec() const
    {
        CellRec<VT> q;
        q.mn = mn > mx ? (VT)~(VT)0 : (VT)mn;
        q.mx = mn > mx ? (VT)0 : (VT)mx;
        return q;
    }
};
template <> struct CellAcc<float> {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    VR_DEV void add(float f)
    {
        if (!(f == f)) { mn = -__builtin_inff(); mx = __builtin_inff(); }
        mn = f < mn ? f : mn;
        mx = f > mx ? f : mx;
    }
    VR_DEV CellRec<float> rec() const { return CellRec<float>{mn, mx}; }
};
template <typename VT> struct CellMerge {
    CellRec<VT> a;
    bool any = false;
    VR_DEV void merge(CellRec<VT> q)
    {
        if (q.mn > q.mx) return;              // nothing in q
        if (!any) { a = q; any = true; return; }
        a.mn = q.mn < a.mn ? q.mn : a.mn;
        a.mx = q.mx > a.mx ? q.mx : a.mx;
    }
    VR_DEV CellRec<VT> rec() const
    {
        if (any) return a;
        CellRec<VT> e;
        e.mn = (VT)1; e.mx = (VT)0;
        return e;
    }
};
template <> struct CellMerge<float> {   // (-inf, +inf) of a NaN survives min / max; (+inf, -inf) is neutral
    float mn = __builtin_inff(), mx = -__builtin_inff();
    VR_DEV void merge(CellRec<float> q) { mn = q.mn < mn ? q.mn : mn; mx = q.mx > mx ? q.mx : mx; }
    VR_DEV CellRec<float> rec() const { return CellRec<float>{mn, mx}; }
};

// Per-column extrema of some voxel rows of one z slice of a micro-brick (4 columns x 4 rows, the
// 16 * sizeof(VT) bytes a lane loads).  The columns stay apart until the end, where the x classes are
// formed from the columns that lie inside the volume (padding columns never enter a class).
template <typename VT> struct CellCols {
    CellAcc<VT> c[4];
    VR_DEV void add_row(const VT *r) { c[0].add(r[0]); c[1].add(r[1]); c[2].add(r[2]); c[3].add(r[3]); }
    VR_DEV void merge(const CellCols &o)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i].mn = o.c[i].mn < c[i].mn ? o.c[i].mn : c[i].mn;
            c[i].mx = o.c[i].mx > c[i].mx ? o.c[i].mx : c[i].mx;
        }
    }
    // x class k (A: all columns, L: 0 and 1, H: 3) over the first `ncols` columns
    VR_DEV CellRec<VT> rec(int k, int ncols) const
    {
        CellAcc<VT> a;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = i < ncols && (k == 0 || (k == 1 && i < 2) || (k == 2 && i == 3));
            if (in) { a.mn = c[i].mn < a.mn ? c[i].mn : a.mn; a.mx = c[i].mx > a.mx ? c[i].mx : a.mx; }
        }
        return a.rec();
    }
};
// UCHAR: the columns as 16-bit lanes of two registers (columns 0|2 and 1|3), packed min / max
template <> struct CellCols<uint8_t> {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    us2 mnA = us2{0x00ff, 0x00ff}, mnB = us2{0x00ff, 0x00ff}, mxA = us2{0, 0}, mxB = us2{0, 0};
    static VR_DEV us2 lanes(uint32_t x) { us2 r; __builtin_memcpy(&r, &x, 4); return r; }
    VR_DEV void add_row(const uint8_t *r)
    {
        uint32_t wd;
        __builtin_memcpy(&wd, r, 4);
        const us2 e = lanes(wd & 0x00ff00ffu), o = lanes((wd >> 8) & 0x00ff00ffu);
        mnA = __builtin_elementwise_min(mnA, e); mxA = __builtin_elementwise_max(mxA, e);
        mnB = __builtin_elementwise_min(mnB, o); mxB = __builtin_elementwise_max(mxB, o);
    }
    VR_DEV void merge(const CellCols &o)
    {
        mnA = __builtin_elementwise_min(mnA, o.mnA); mxA = __builtin_elementwise_max(mxA, o.mxA);
        mnB = __builtin_elementwise_min(mnB, o.mnB); mxB = __builtin_elementwise_max(mxB, o.mxB);
    }
    VR_DEV CellRec<uint8_t> rec(int k, int ncols) const
    {
        const unsigned short mn[4] = {mnA.x, mnB.x, mnA.y, mnB.y}, mx[4] = {mxA.x, mxB.x, mxA.y, mxB.y};
        unsigned short lo = 0x00ff, hi = 0;   // nothing: (255, 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = i < ncols && (k == 0 || (k == 1 && i < 2) || (k == 2 && i == 3));
            if (in) { lo = min(lo, mn[i]); hi = max(hi, mx[i]); }
        }
        CellRec<uint8_t> q;
        q.mn = (uint8_t)lo; q.mx = (uint8_t)hi;
        return q;
    }
};

// R rows of cells of 2^S voxels per workgroup: the R * E + 3 voxel rows they cover between them are read
// once (R * E / 4 + 2 micro-brick rows).  A lane takes one z slice of a micro-brick column -- 16 *
// sizeof(VT) contiguous bytes per micro-brick, a wave's loads cover whole lines -- and reduces each
// slice to three y classes of its four voxel rows, like the x classes: A all four, L rows 0 and 1 (what
// the row of cells below takes from this micro-brick row), H row 3 (what the one above takes).  Which
// class of which micro-brick row goes to which row of cells is known at compile time; only rows beyond
// the volume's height (padding of the last micro-brick row) are tested for.
template <typename VT, int R, int S>
__global__ __launch_bounds__(kThreads) void vr_cell_xy_kernel(VolView vv, CellView grid, CellRec<VT> *xy /* [d][cy][cx] */)
{
    extern __shared__ unsigned char s_raw[];
    CellRec<VT> *s_cls = reinterpret_cast<CellRec<VT> *>(s_raw);   // [R][nbx][4 z slices][3 classes]
    constexpr int BPC = (1 << S) / 4;        // micro-brick rows per row of cells
    constexpr int NMY = R * BPC + 2;         // micro-brick rows a workgroup reads
    constexpr int N16 = (int)sizeof(VT);     // 16-byte loads per slice
    constexpr int kPre = sizeof(VT) == 1 ? 6 : sizeof(VT) == 2 ? 4 : 2;   // slices in flight per lane
    const int nrb = (grid.cy + R - 1) / R;
    const int cy0 = ((int)blockIdx.x % nrb) * R, mz = (int)blockIdx.x / nrb;
    const int nrows = min(R, grid.cy - cy0);
    const VT *base = (const VT *)vv.data;
    const int my_first = cy0 * BPC - 1, my_last = (vv.h - 1) >> 2;
    for (int t = threadIdx.x; t < (int)vv.nbx * 4; t += kThreads) {
        const int mx = t >> 2, dz = t & 3;
        CellCols<VT> acc[R];
        const VT *col = base + (unsigned long long)mz * vv.zstride + (unsigned long long)mx * 64ull +
                        (unsigned long long)dz * 16ull;
#pragma unroll
        for (int j0 = 0; j0 < NMY; j0 += kPre) {
            uint4 q[kPre][N16];
#pragma unroll
            for (int jj = 0; jj < kPre; ++jj) {
                if (j0 + jj >= NMY) continue;
                const int my = min(max(my_first + j0 + jj, 0), my_last);   // (rows outside are loaded and not used)
                const uint4 *p = reinterpret_cast<const uint4 *>(col + (unsigned long long)my * vv.ystride);
#pragma unroll
                for (int i = 0; i < N16; ++i) q[jj][i] = p[i];
            }
#pragma unroll
            for (int jj = 0; jj < kPre; ++jj) {
                constexpr int dummy = 0; (void)dummy;
                const int j = j0 + jj;
                if (j >= NMY) continue;
                const int my = my_first + j;
                if (my < 0 || my > my_last) continue;   // (uniform over the workgroup)
                VT v[16];
                __builtin_memcpy(v, q[jj], sizeof v);
                const int nvalid = min(4, vv.h - 4 * my);   // rows of this micro-brick row inside the volume
                CellCols<VT> A, L, H;
                L.add_row(v);
                if (nvalid > 1) L.add_row(v + 4);
                A = L;
                if (nvalid > 2) A.add_row(v + 8);
                if (nvalid > 3) { H.add_row(v + 12); A.merge(H); }
#pragma unroll
                for (int rl = 0; rl < R; ++rl) {
                    const int jr = 1 + rl * BPC;   // first micro-brick row of this row of cells
                    if (j == jr - 1) acc[rl].merge(H);
                    if (j >= jr && j < jr + BPC) acc[rl].merge(A);
                    if (j == jr + BPC) acc[rl].merge(L);
                }
            }
        }
        const int ncols = min(4, vv.w - 4 * mx);
#pragma unroll
        for (int rl = 0; rl < R; ++rl)
#pragma unroll
            for (int k = 0; k < 3; ++k) s_cls[((rl * (int)vv.nbx + mx) * 4 + dz) * 3 + k] = acc[rl].rec(k, ncols);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nrows * grid.cx * 4; i += kThreads) {
        const int rl = i / (grid.cx * 4), j = i - rl * grid.cx * 4;
        const int dz = j / grid.cx, cxi = j - dz * grid.cx;   // (neighbouring threads write neighbouring records)
        const int z = 4 * mz + dz;
        if (z >= vv.d) continue;
        const CellRec<VT> *cls = s_cls + (size_t)rl * vv.nbx * 12;
        CellMerge<VT> a;
        const int b0 = cxi * BPC;
        if (b0 - 1 >= 0) a.merge(cls[((b0 - 1) * 4 + dz) * 3 + 2]);
        for (int b = b0; b < b0 + BPC && b < (int)vv.nbx; ++b) a.merge(cls[(b * 4 + dz) * 3 + 0]);
        if (b0 + BPC < (int)vv.nbx) a.merge(cls[((b0 + BPC) * 4 + dz) * 3 + 1]);
        xy[((size_t)z * grid.cy + (cy0 + rl)) * grid.cx + cxi] = a.rec();
    }
}

// (a thread takes kZCells cells next to each other in x: the records of integer volumes are 2 or 4 bytes)
constexpr int kZCells = 4;
template <typename VT>
__global__ __launch_bounds__(kThreads) void vr_cell_z_kernel(CellView grid, int d, const CellRec<VT> *xy, float2 *out)
{
    const int gx = (grid.cx + kZCells - 1) / kZCells;
    const size_t n_groups = (size_t)gx * grid.cy * grid.cz;
    const size_t gi = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (gi >= n_groups) return;
    const int cx0 = (int)(gi % (size_t)gx) * kZCells;
    const int cyi = (int)((gi / (size_t)gx) % (size_t)grid.cy), czi = (int)(gi / ((size_t)gx * grid.cy));
    const size_t plane = (size_t)grid.cx * grid.cy;
    const size_t row = (size_t)cyi * grid.cx + cx0;
    const int z_a = max((czi << grid.shift) - 1, 0), z_b = min(((czi + 1) << grid.shift) + 1, d - 1);
    const int n = min(kZCells, grid.cx - cx0);
    CellMerge<VT> a[kZCells];
    for (int z = z_a; z <= z_b; ++z) {
        const CellRec<VT> *p = xy + (size_t)z * plane + row;
#pragma unroll
        for (int i = 0; i < kZCells; ++i)
            if (i < n) a[i].merge(p[i]);
    }
#pragma unroll
    for (int i = 0; i < kZCells; ++i) {
        if (i >= n) continue;
        const CellRec<VT> q = a[i].rec();
        out[(size_t)czi * plane + row + i] =
            q.mn > q.mx ? make_float2(__builtin_inff(), -__builtin_inff()) : make_float2((float)q.mn, (float)q.mx);
    }
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

namespace {
// one thread per macro cell: the maximum of the bounds of its (up to) 4 x 4 x 4 cells
__global__ __launch_bounds__(kThreads) void vr_cell_coarse_bounds_kernel(CellView g, float *cbound)
{
    const size_t n = (size_t)g.ccx * g.ccy * g.ccz;
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int X = (int)(i % (size_t)g.ccx), Y = (int)((i / (size_t)g.ccx) % (size_t)g.ccy), Z = (int)(i / ((size_t)g.ccx * g.ccy));
    constexpr int E = 1 << kLeapShift;
    float m = 0.f;   // (bounds are opacities: >= 0)
    for (int z = Z * E; z < (Z + 1) * E && z < g.cz; ++z)
        for (int y = Y * E; y < (Y + 1) * E && y < g.cy; ++y)
            for (int x = X * E; x < (X + 1) * E && x < g.cx; ++x) {
                const float b = g.bound[((size_t)z * g.cy + y) * g.cx + x];
                m = (b > m || b != b) ? b : m;   // (a NaN bound rules nothing out: it must survive the maximum)
            }
    cbound[i] = m;
}
} // namespace

namespace {
// CellView::cdist, pass 0: 1 = the macro cell is free at level j (bound < j / 8), 0 = it is not
__global__ __launch_bounds__(kThreads) void vr_leap_radius_init_kernel(const float *cbound, size_t n, uint8_t *d)
{
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float tau = (float)(blockIdx.y + 1u) * 0.125f;
    d[(size_t)blockIdx.y * n + i] = cbound[i] < tau ? 1 : 0;   // (a NaN bound is not free)
}
// pass r >= 1: a macro cell whose cube of radius r - 1 is free (d == r) and whose 26 neighbours' cubes of radius
// r - 1 are free as well (d >= r; neighbours outside the grid do not exist) has a free cube of radius r (d = r + 1)
__global__ __launch_bounds__(kThreads) void vr_leap_radius_pass_kernel(const uint8_t *in, uint8_t *out, int cx, int cy, int cz,
                                                                       int r)
{
    const size_t n = (size_t)cx * cy * cz;
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint8_t *src = in + (size_t)blockIdx.y * n;
    uint8_t v = src[i];
    if (v == (uint8_t)r) {
        const int X = (int)(i % (size_t)cx), Y = (int)((i / (size_t)cx) % (size_t)cy), Z = (int)(i / ((size_t)cx * cy));
        bool all = true;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int x = X + dx, y = Y + dy, z = Z + dz;
                    if (x < 0 || y < 0 || z < 0 || x >= cx || y >= cy || z >= cz) continue;
                    all = all && src[((size_t)z * cy + y) * cx + x] >= (uint8_t)r;
                }
        if (all) v = (uint8_t)(r + 1);
    }
    out[(size_t)blockIdx.y * n + i] = v;
}
} // namespace

hipError_t vr_launch_cell_leap_radius(const CellView &grid, const float *cbound, uint8_t *dist, const uint8_t **result,
                                      hipStream_t stream)
{
    const size_t n = (size_t)grid.ccx * grid.ccy * grid.ccz;
    const dim3 g((unsigned)((n + kThreads - 1) / kThreads), kLeapLevels);
    uint8_t *buf[2] = {dist, dist + (size_t)kLeapLevels * n};
    hipLaunchKernelGGL(vr_leap_radius_init_kernel, g, dim3(kThreads), 0, stream, cbound, n, buf[0]);
    int cur = 0;
    for (int r = 1; r <= kLeapRadius; ++r) {
        hipLaunchKernelGGL(vr_leap_radius_pass_kernel, g, dim3(kThreads), 0, stream, (const uint8_t *)buf[cur], buf[cur ^ 1],
                           grid.ccx, grid.ccy, grid.ccz, r);
        cur ^= 1;
    }
    *result = buf[cur];
    return hipGetLastError();
}

hipError_t vr_launch_cell_coarse_bounds(const CellView &grid, float *cbound, hipStream_t stream)
{
    const size_t n = (size_t)grid.ccx * grid.ccy * grid.ccz;
    hipLaunchKernelGGL(vr_cell_coarse_bounds_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       stream, grid, cbound);
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

namespace {
template <typename VT, int R, int S>
hipError_t launch_xy_rs(const VolView &vol, const CellView &grid, hipStream_t stream, CellRec<VT> *rec)
{
    const size_t lds = (size_t)R * vol.nbx * 4 * 3 * sizeof(CellRec<VT>);
    int nb = 0;
    hipError_t e = vr_prepare_kernel(vr_cell_xy_kernel<VT, R, S>, kThreads, lds, &nb, "cell grid xy", 0);
    if (e != hipSuccess) return e;
    const int nrb = (grid.cy + R - 1) / R;
    hipLaunchKernelGGL((vr_cell_xy_kernel<VT, R, S>), dim3((unsigned)(nrb * (int)vol.nbz)), dim3(kThreads), lds, stream,
                       vol, grid, rec);
    return hipGetLastError();
}
template <typename VT, int R>
hipError_t launch_xy(const VolView &vol, const CellView &grid, hipStream_t stream, CellRec<VT> *rec)
{
    switch (grid.shift) {
    case 2: return launch_xy_rs<VT, R, 2>(vol, grid, stream, rec);
    case 3: return launch_xy_rs<VT, R, 3>(vol, grid, stream, rec);
    case 4: return launch_xy_rs<VT, (R > 2 ? 2 : R), 4>(vol, grid, stream, rec);
    default: return hipErrorInvalidValue;
    }
}

template <typename VT>
hipError_t launch_separable(const VolView &vol, const CellView &grid, float2 *minmax, hipStream_t stream, void *records)
{
    CellRec<VT> *rec = static_cast<CellRec<VT> *>(records);
    // rows of cells per workgroup: as many (<= 4) as keep the class table within 64 KiB of LDS
    const size_t per_row = (size_t)vol.nbx * 4 * 3 * sizeof(CellRec<VT>);
    hipError_t e;
    if (4 * per_row <= 65536) e = launch_xy<VT, 4>(vol, grid, stream, rec);
    else if (2 * per_row <= 65536) e = launch_xy<VT, 2>(vol, grid, stream, rec);
    else e = launch_xy<VT, 1>(vol, grid, stream, rec);
    if (e != hipSuccess) return e;
    const size_t n_groups = (size_t)((grid.cx + kZCells - 1) / kZCells) * grid.cy * grid.cz;
    hipLaunchKernelGGL(vr_cell_z_kernel<VT>, dim3((unsigned)((n_groups + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       stream, grid, vol.d, (const CellRec<VT> *)rec, minmax);
    return hipGetLastError();
}
} // namespace

hipError_t vr_launch_cell_minmax(const VolView &vol, int format, const CellView &grid,
                                 float2 *minmax, hipStream_t stream, void *records)
{
    const size_t n_cells = (size_t)grid.cx * grid.cy * grid.cz;
    if (records && grid.shift <= 4) {
        // separable streaming build (records: d * cy * cx CellRec of scratch, vr_cell_record_bytes)
        hipError_t e = hipErrorInvalidValue;
        switch (format) {
        case VRHIP_UCHAR: e = launch_separable<uint8_t>(vol, grid, minmax, stream, records); break;
        case VRHIP_USHORT: e = launch_separable<uint16_t>(vol, grid, minmax, stream, records); break;
        case VRHIP_FLOAT: e = launch_separable<float>(vol, grid, minmax, stream, records); break;
        default: break;
        }
        return e;
    }
    // one wave per cell, in launches of at most 2^24 cells (a launch of more than 2^32 threads does not run)
    const size_t per_block = kThreads / 64, chunk = (size_t)1 << 24;
    for (size_t c0 = 0; c0 < n_cells; c0 += chunk) {
        const size_t n = std::min(chunk, n_cells - c0);
        dim3 g((unsigned)((n + per_block - 1) / per_block)), block(kThreads);
        switch (format) {
        case VRHIP_UCHAR:
            hipLaunchKernelGGL(vr_cell_minmax_kernel<uint8_t>, g, block, 0, stream, vol, grid, minmax, c0);
            break;
        case VRHIP_USHORT:
            hipLaunchKernelGGL(vr_cell_minmax_kernel<uint16_t>, g, block, 0, stream, vol, grid, minmax, c0);
            break;
        case VRHIP_FLOAT:
            hipLaunchKernelGGL(vr_cell_minmax_kernel<float>, g, block, 0, stream, vol, grid, minmax, c0);
            break;
        default: return hipErrorInvalidValue;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
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
