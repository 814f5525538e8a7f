// vr_sampling.h -- device-side building blocks of the ray-cast kernels (single-TU header,
// included by vr_raycast.hip only): volume / transfer-function / prefix reads restated from
// the OpenCL 1.2 image rules, the ESS skip bitmap kernel, ray set-up
// (/root/reference/src/kernel/volumeraycast.cl:605-683) and the diagnostic phase stamps.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "vr_device_math.h"
#include "vr_internal.h"

namespace {

#ifndef VR_BLOCK_DIM
#define VR_BLOCK_DIM 256
#endif
constexpr int kBlockDim = VR_BLOCK_DIM;   // 4 waves
#ifndef VR_MAXBRICK
#define VR_MAXBRICK 4
#endif
constexpr int kMaxBrickSteps = VR_MAXBRICK;   // DDA steps per round while other lanes wait to sample
#ifndef VR_BATCH
#define VR_BATCH 4
#endif
constexpr int kBatch = VR_BATCH;     // consecutive samples of a ray evaluated per round

enum : int { S_DONE = 0, S_BRICK = 1, S_SAMPLE = 2 };

// Diagnostic build only (-DVR_STAMPS, tools/stamps.sh): per-phase shader-clock totals summed
// over all waves.  Every stamp drains the memory queues, so only the SHARES are meaningful;
// the stamp values leave the kernel through g_stamps alone and feed no output.
#ifdef VR_STAMPS
__device__ unsigned long long g_stamps[32];   // [0,16) phase 1, [16,32) phase 2
__device__ unsigned long long g_wave_span[2][2][8192];   // [phase][start|end][wave]: s_memtime of every wave
#define VR_STAMP_DECL unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = vr_stamp(), st_first = st_last
#define VR_STAMP(i) do { unsigned long long n_ = vr_stamp(); st_acc[i] += n_ - st_last; st_last = n_; } while (0)
#define VR_COUNT(i) st_acc[i] += 1
#define VR_STAMP_FLUSH_AT(b) do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_stamps[(b) + i_], st_acc[i_]); atomicAdd(&g_stamps[(b) + 12], vr_stamp() - st_first); unsigned w_ = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 8191u; g_wave_span[(b) ? 1 : 0][0][w_] = st_first; g_wave_span[(b) ? 1 : 0][1][w_] = vr_stamp(); } } while (0)
__device__ __forceinline__ unsigned long long vr_stamp()
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#else
#define VR_STAMP_DECL
#define VR_STAMP(i)
#define VR_COUNT(i)
#define VR_STAMP_FLUSH_AT(b)
#endif

// ------------------------------------------------------------------ volume reads

// One entry of the footprint volume (VolView::fp): the 2x2x2 voxels of a trilinear fetch, value
// j = dx + 2 dy + 4 dz, loaded with one 8 / 16 / 2x16-byte access.
template <typename VT> struct FpEntry;
template <> struct FpEntry<uint8_t> {
    uint2 r;
    template <int J> VR_DEV float v() const { return (float)(((J < 4 ? r.x : r.y) >> (8 * (J & 3))) & 0xffu); }
};
template <> struct FpEntry<uint16_t> {
    uint4 r;
    template <int J> VR_DEV float v() const
    {
        const uint32_t w = (J >> 1) == 0 ? r.x : (J >> 1) == 1 ? r.y : (J >> 1) == 2 ? r.z : r.w;
        return (float)((J & 1) ? (w >> 16) : (w & 0xffffu));
    }
};
template <> struct FpEntry<float> {
    float4 a, b;
    template <int J> VR_DEV float v() const
    {
        return J == 0 ? a.x : J == 1 ? a.y : J == 2 ? a.z : J == 3 ? a.w
             : J == 4 ? b.x : J == 5 ? b.y : J == 6 ? b.z : b.w;
    }
};

// LS (experiment, vr_raycast_staged_kernel): a box of kStageEdge^3 voxels around the wave's rays is
// staged in LDS (x fastest, origin (ox, oy, oz), a multiple of 4) and fetches whose voxels all lie
// inside it are served from there; the others go to HBM / L2 as usual.  Same voxels, same blend.
constexpr int kStageEdge = 20;

template <typename VT, int INSTR, bool FP = false, bool LS = false>
struct Vol {
    const VT *p;
    const __attribute__((address_space(3))) VT *lds = nullptr;   // LS: this wave's box in LDS
    bool staged = false;            // LS: the box holds the voxels [o, o + kStageEdge)^3
    int ox = 0, oy = 0, oz = 0;
    int w1, h1, d1;   // res - 1
    float fw, fh, fd;
    float inv_max;
    uint32_t nbx, nby, ystride;     // micro-brick layout (vr_internal.h)
    uint32_t zstride;               // fits 32 bits (<= 2048 * 2048 * 64): one 32x32->64 multiply per z
    uint32_t *touched;
    const VT *pc[3];                // channels 1..3 of CL_RG / CL_RGBA volumes (XS variants only)
    int channels;
    const FpEntry<VT> *fp;          // FP: footprint volume (VolView::fp)
    uint32_t fp_ystride, fp_zstride;   // entries per brick row / brick slice of it

    // FP: the entry at entry coordinates (ex, ey, ez) = low-corner texel + 1, already in [0, res]
    VR_DEV FpEntry<VT> entry_at(uint32_t ex, uint32_t ey, uint32_t ez) const   // entry coordinates, already in range
    {
        const uint32_t in = __umul24(ey >> 2, fp_ystride) + ((ex >> 2) << 6) + ((ez & 3u) << 4) +
                            ((ey & 3u) << 2) + (ex & 3u);
        return fp[(unsigned long long)(ez >> 2) * (unsigned long long)fp_zstride + (unsigned long long)in];
    }
    // The entry coordinate of low-corner texel fx + off (fx = floor(..), a whole number as a float; off a small whole
    // number): clamp(ix + off, -1, res - 1) + 1 taken in the float domain -- one v_med3_f32 and the conversion instead
    // of a conversion and an integer max, min and add (the compiler cannot prove -1 <= res - 1 for a v_med3_i32).
    // Whole numbers far below 2^24: every step is exact, the index is the same.
    VR_DEV uint32_t ecoord(float fx, float off1, float fres) const   // off1 = off + 1, fres = (float)res
    {
        return (uint32_t)(int)__builtin_amdgcn_fmed3f(fx + off1, 0.f, fres);
    }

    // this volume's channel ch >= 1 as a single-channel volume
    VR_DEV Vol channel(int ch) const
    {
        Vol v = *this;
        v.p = pc[ch - 1];
        return v;
    }

    // per-axis parts of the element index of voxel (x, y, z) in the 4x4x4 micro-brick layout
    VR_DEV uint32_t xoff(int x) const { return ((uint32_t)(x >> 2) << 6) + (uint32_t)(x & 3); }
    VR_DEV uint32_t yoff(int y) const
    {
        return __umul24((uint32_t)(y >> 2), ystride) + ((uint32_t)(y & 3) << 2);
    }
    VR_DEV unsigned long long zoff(int z) const
    {
        return (unsigned long long)(uint32_t)(z >> 2) * (unsigned long long)zstride +
               (unsigned long long)((z & 3) << 4);
    }

    VR_DEV void touch(int x, int y, int z) const
    {
        if (INSTR == 2) {
            unsigned long long b = ((unsigned long long)(z >> 2) * (unsigned long long)nby +
                                    (unsigned long long)(y >> 2)) * (unsigned long long)nbx +
                                   (unsigned long long)(x >> 2);
            uint32_t bit = 1u << (uint32_t)(b & 31);
            uint32_t *wp = touched + (b >> 5);
            if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit))
                atomicOr(wp, bit);
        }
    }
    VR_DEV float raw(uint32_t xo, uint32_t yo, unsigned long long zo, int x, int y, int z) const
    {
        touch(x, y, z);
        return (float)p[zo + (unsigned long long)(yo + xo)];
    }

    // read_imagef(vol, linearSmp, pos).x -- normalised, CLAMP_TO_EDGE, LINEAR
    VR_DEV float linear(float px, float py, float pz) const
    {
        float u = px * fw, v = py * fh, s = pz * fd;
        float ub = u - 0.5f, vb = v - 0.5f, sb = s - 0.5f;
        float fx = floorf(ub), fy = floorf(vb), fz = floorf(sb);
        float a = ub - fx, b = vb - fy, c = sb - fz;
        if (FP) {   // the same eight voxels and the same blend, from one load
            const FpEntry<VT> e = entry_at(ecoord(fx, 1.f, fw), ecoord(fy, 1.f, fh), ecoord(fz, 1.f, fd));
            float c00 = lerpf(e.template v<0>(), e.template v<1>(), a);
            float c10 = lerpf(e.template v<2>(), e.template v<3>(), a);
            float c01 = lerpf(e.template v<4>(), e.template v<5>(), a);
            float c11 = lerpf(e.template v<6>(), e.template v<7>(), a);
            float c0 = lerpf(c00, c10, b);
            float c1 = lerpf(c01, c11, b);
            return lerpf(c0, c1, c) * inv_max;
        }
        int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        int x0 = iclamp(ix, 0, w1), x1 = iclamp(ix + 1, 0, w1);
        int y0 = iclamp(iy, 0, h1), y1 = iclamp(iy + 1, 0, h1);
        int z0 = iclamp(iz, 0, d1), z1 = iclamp(iz + 1, 0, d1);
        if (LS && staged && (uint32_t)(x0 - ox) < (uint32_t)kStageEdge && (uint32_t)(x1 - ox) < (uint32_t)kStageEdge &&
            (uint32_t)(y0 - oy) < (uint32_t)kStageEdge && (uint32_t)(y1 - oy) < (uint32_t)kStageEdge &&
            (uint32_t)(z0 - oz) < (uint32_t)kStageEdge && (uint32_t)(z1 - oz) < (uint32_t)kStageEdge) {
            const int bx0 = x0 - ox, bx1 = x1 - ox, by0 = (y0 - oy) * kStageEdge, by1 = (y1 - oy) * kStageEdge;
            const int bz0 = (z0 - oz) * (kStageEdge * kStageEdge), bz1 = (z1 - oz) * (kStageEdge * kStageEdge);
            float c00 = lerpf((float)lds[bz0 + by0 + bx0], (float)lds[bz0 + by0 + bx1], a);
            float c10 = lerpf((float)lds[bz0 + by1 + bx0], (float)lds[bz0 + by1 + bx1], a);
            float c01 = lerpf((float)lds[bz1 + by0 + bx0], (float)lds[bz1 + by0 + bx1], a);
            float c11 = lerpf((float)lds[bz1 + by1 + bx0], (float)lds[bz1 + by1 + bx1], a);
            return lerpf(lerpf(c00, c10, b), lerpf(c01, c11, b), c) * inv_max;
        }
        const uint32_t xo0 = xoff(x0), xo1 = xoff(x1), yo0 = yoff(y0), yo1 = yoff(y1);
        const unsigned long long zo0 = zoff(z0), zo1 = zoff(z1);
        float v000 = raw(xo0, yo0, zo0, x0, y0, z0), v100 = raw(xo1, yo0, zo0, x1, y0, z0);
        float v010 = raw(xo0, yo1, zo0, x0, y1, z0), v110 = raw(xo1, yo1, zo0, x1, y1, z0);
        float v001 = raw(xo0, yo0, zo1, x0, y0, z1), v101 = raw(xo1, yo0, zo1, x1, y0, z1);
        float v011 = raw(xo0, yo1, zo1, x0, y1, z1), v111 = raw(xo1, yo1, zo1, x1, y1, z1);
        float c00 = lerpf(v000, v100, a);
        float c10 = lerpf(v010, v110, a);
        float c01 = lerpf(v001, v101, a);
        float c11 = lerpf(v011, v111, a);
        float c0 = lerpf(c00, c10, b);
        float c1 = lerpf(c01, c11, b);
        return lerpf(c0, c1, c) * inv_max;
    }

    // index of the majorant-grid cell holding the low corner texel (x0, y0, z0) of linear()'s
    // footprint: the fetch uses voxels x0..x0+1 etc., all inside that cell's halo'd extent
    VR_DEV uint32_t cell_index(float px, float py, float pz, const CellView &g) const
    {
        const int x0 = iclamp((int)floorf(px * fw - 0.5f), 0, w1);
        const int y0 = iclamp((int)floorf(py * fh - 0.5f), 0, h1);
        const int z0 = iclamp((int)floorf(pz * fd - 0.5f), 0, d1);
        return ((uint32_t)(z0 >> g.shift) * (uint32_t)g.cy + (uint32_t)(y0 >> g.shift)) *
                   (uint32_t)g.cx + (uint32_t)(x0 >> g.shift);
    }

    // the same from texel-space coordinates (u = px * fw - 0.5 ...) that need only be good to a
    // texel: the cells' extents carry a one-texel halo (CellView)
    VR_DEV uint32_t cell_index_texel(float u, float v, float s, const CellView &g) const
    {
        const int x0 = iclamp((int)floorf(u), 0, w1), y0 = iclamp((int)floorf(v), 0, h1);
        const int z0 = iclamp((int)floorf(s), 0, d1);
        return ((uint32_t)(z0 >> g.shift) * (uint32_t)g.cy + (uint32_t)(y0 >> g.shift)) *
                   (uint32_t)g.cx + (uint32_t)(x0 >> g.shift);
    }

    // -gradientCentralDiff(vol, pos).xyz (volumeraycast.cl:159-178, :814).  The six taps sit
    // exactly one texel from the centre sample (offset = 1/volRes, :162): they are evaluated
    // in texel space -- the centre's filter weights with indices shifted by -+1 and clamped to
    // the edge -- so the 4x4x4 neighbourhood is loaded once: 32 voxel loads and one set of
    // coordinate arithmetic instead of 6 x (8 loads + coordinates).  DESIGN.md "Numerics".
    VR_DEV f3 neg_gradient(float px, float py, float pz) const
    {
        float ub = px * fw - 0.5f, vb = py * fh - 0.5f, sb = pz * fd - 0.5f;
        float fx = floorf(ub), fy = floorf(vb), fz = floorf(sb);
        float a = ub - fx, b = vb - fy, c = sb - fz;
        if (FP) return neg_gradient_fp(fx, fy, fz, a, b, c);
        int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        int X[4], Y[4], Z[4];
        uint32_t xo[4], yo[4];
        unsigned long long zo[4];
        bool in_box = LS && staged;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            X[k] = iclamp(ix - 1 + k, 0, w1);
            Y[k] = iclamp(iy - 1 + k, 0, h1);
            Z[k] = iclamp(iz - 1 + k, 0, d1);
            if (LS)
                in_box = in_box && (uint32_t)(X[k] - ox) < (uint32_t)kStageEdge && (uint32_t)(Y[k] - oy) < (uint32_t)kStageEdge &&
                         (uint32_t)(Z[k] - oz) < (uint32_t)kStageEdge;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // (staged box: plain offsets into the wave's LDS copy; else the micro-brick layout)
            xo[k] = in_box ? (uint32_t)(X[k] - ox) : xoff(X[k]);
            yo[k] = in_box ? (uint32_t)((Y[k] - oy) * kStageEdge) : yoff(Y[k]);
            zo[k] = in_box ? (unsigned long long)((Z[k] - oz) * (kStageEdge * kStageEdge)) : zoff(Z[k]);
        }
#define VR_L(xi, yi, zi) (in_box ? (float)lds[(uint32_t)zo[zi] + yo[yi] + xo[xi]] : raw(xo[xi], yo[yi], zo[zi], X[xi], Y[yi], Z[zi]))
#define VR_R(yi, zi) lerpf(VR_L(1, yi, zi), VR_L(2, yi, zi), a)   /* texels (x0, x1)   */
#define VR_M(yi, zi) lerpf(VR_L(0, yi, zi), VR_L(1, yi, zi), a)   /* texels (x0-1, x0) */
#define VR_P(yi, zi) lerpf(VR_L(2, yi, zi), VR_L(3, yi, zi), a)   /* texels (x1, x1+1) */
        const float r01 = VR_R(0, 1), r11 = VR_R(1, 1), r21 = VR_R(2, 1), r31 = VR_R(3, 1);
        const float r02 = VR_R(0, 2), r12 = VR_R(1, 2), r22 = VR_R(2, 2), r32 = VR_R(3, 2);
        const float r10 = VR_R(1, 0), r20 = VR_R(2, 0), r13 = VR_R(1, 3), r23 = VR_R(2, 3);
        f3 s1, s2;
        s1.x = lerpf(lerpf(VR_M(1, 1), VR_M(2, 1), b), lerpf(VR_M(1, 2), VR_M(2, 2), b), c) * inv_max;
        s2.x = lerpf(lerpf(VR_P(1, 1), VR_P(2, 1), b), lerpf(VR_P(1, 2), VR_P(2, 2), b), c) * inv_max;
        s1.y = lerpf(lerpf(r01, r11, b), lerpf(r02, r12, b), c) * inv_max;
        s2.y = lerpf(lerpf(r21, r31, b), lerpf(r22, r32, b), c) * inv_max;
        s1.z = lerpf(lerpf(r10, r20, b), lerpf(r11, r21, b), c) * inv_max;
        s2.z = lerpf(lerpf(r12, r22, b), lerpf(r13, r23, b), c) * inv_max;
#undef VR_L
#undef VR_R
#undef VR_M
#undef VR_P
        f3 g = sub3(s2, s1);
        f3 n = normalize3(g);
        if (dot3(g, g) == 0.0f) n = mk3(0.57735f, 0.57735f, 0.57735f);
        return neg3(n);
    }

    // neg_gradient() on the footprint volume: the 4x4x4 neighbourhood is the eight entries at
    // (ix - 1 + 2i, iy - 1 + 2j, iz - 1 + 2k); texel (xi, yi, zi) of it is value
    // (xi & 1) + 2 (yi & 1) + 4 (zi & 1) of entry (xi >> 1, yi >> 1, zi >> 1).  Same taps, same blends.
    VR_DEV f3 neg_gradient_fp(float fx, float fy, float fz, float a, float b, float c) const
    {
        // (entry coordinates of the low-corner texels ix - 1 and ix + 1 per axis: ecoord)
        const uint32_t xm = ecoord(fx, 0.f, fw), xp = ecoord(fx, 2.f, fw);
        const uint32_t ym = ecoord(fy, 0.f, fh), yp = ecoord(fy, 2.f, fh);
        const uint32_t zm = ecoord(fz, 0.f, fd), zp = ecoord(fz, 2.f, fd);
        const FpEntry<VT> e000 = entry_at(xm, ym, zm), e100 = entry_at(xp, ym, zm);
        const FpEntry<VT> e010 = entry_at(xm, yp, zm), e110 = entry_at(xp, yp, zm);
        const FpEntry<VT> e001 = entry_at(xm, ym, zp), e101 = entry_at(xp, ym, zp);
        const FpEntry<VT> e011 = entry_at(xm, yp, zp), e111 = entry_at(xp, yp, zp);
#define VR_E(xi, yi, zi)                                                                             \
    (((zi) >> 1) == 0 ? (((yi) >> 1) == 0 ? (((xi) >> 1) == 0 ? e000 : e100)                        \
                                          : (((xi) >> 1) == 0 ? e010 : e110))                        \
                      : (((yi) >> 1) == 0 ? (((xi) >> 1) == 0 ? e001 : e101)                        \
                                          : (((xi) >> 1) == 0 ? e011 : e111)))
#define VR_L(xi, yi, zi) VR_E(xi, yi, zi).template v<((xi) & 1) + 2 * ((yi) & 1) + 4 * ((zi) & 1)>()
#define VR_R(yi, zi) lerpf(VR_L(1, yi, zi), VR_L(2, yi, zi), a)   /* texels (x0, x1)   */
#define VR_M(yi, zi) lerpf(VR_L(0, yi, zi), VR_L(1, yi, zi), a)   /* texels (x0-1, x0) */
#define VR_P(yi, zi) lerpf(VR_L(2, yi, zi), VR_L(3, yi, zi), a)   /* texels (x1, x1+1) */
        const float r01 = VR_R(0, 1), r11 = VR_R(1, 1), r21 = VR_R(2, 1), r31 = VR_R(3, 1);
        const float r02 = VR_R(0, 2), r12 = VR_R(1, 2), r22 = VR_R(2, 2), r32 = VR_R(3, 2);
        const float r10 = VR_R(1, 0), r20 = VR_R(2, 0), r13 = VR_R(1, 3), r23 = VR_R(2, 3);
        f3 s1, s2;
        s1.x = lerpf(lerpf(VR_M(1, 1), VR_M(2, 1), b), lerpf(VR_M(1, 2), VR_M(2, 2), b), c) * inv_max;
        s2.x = lerpf(lerpf(VR_P(1, 1), VR_P(2, 1), b), lerpf(VR_P(1, 2), VR_P(2, 2), b), c) * inv_max;
        s1.y = lerpf(lerpf(r01, r11, b), lerpf(r02, r12, b), c) * inv_max;
        s2.y = lerpf(lerpf(r21, r31, b), lerpf(r22, r32, b), c) * inv_max;
        s1.z = lerpf(lerpf(r10, r20, b), lerpf(r11, r21, b), c) * inv_max;
        s2.z = lerpf(lerpf(r12, r22, b), lerpf(r13, r23, b), c) * inv_max;
#undef VR_E
#undef VR_L
#undef VR_R
#undef VR_M
#undef VR_P
        f3 g = sub3(s2, s1);
        f3 n = normalize3(g);
        if (dot3(g, g) == 0.0f) n = mk3(0.57735f, 0.57735f, 0.57735f);
        return neg3(n);
    }

    // fast_length(s2 - s1) of gradientCentralDiff (:177) with the same texel-space taps: what
    // illumType 4 feeds to the transfer function (:796-799)
    VR_DEV float gradient_len(float px, float py, float pz) const
    {
        float ub = px * fw - 0.5f, vb = py * fh - 0.5f, sb = pz * fd - 0.5f;
        float fx = floorf(ub), fy = floorf(vb), fz = floorf(sb);
        float a = ub - fx, b = vb - fy, c = sb - fz;
        int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        f3 s1, s2;
        s1.x = tri_at(ix - 1, iy, iz, a, b, c);
        s2.x = tri_at(ix + 1, iy, iz, a, b, c);
        s1.y = tri_at(ix, iy - 1, iz, a, b, c);
        s2.y = tri_at(ix, iy + 1, iz, a, b, c);
        s1.z = tri_at(ix, iy, iz - 1, a, b, c);
        s2.z = tri_at(ix, iy, iz + 1, a, b, c);
        return len3(sub3(s2, s1));
    }

    // trilinear blend with weights (a, b, c) of the 2x2x2 texels whose low corner is (x, y, z),
    // indices clamped to the edge: a fetch a whole number of texels away from the centre sample
    VR_DEV float tri_at(int x, int y, int z, float a, float b, float c) const
    {
        const int x0 = iclamp(x, 0, w1), x1 = iclamp(x + 1, 0, w1);
        const int y0 = iclamp(y, 0, h1), y1 = iclamp(y + 1, 0, h1);
        const int z0 = iclamp(z, 0, d1), z1 = iclamp(z + 1, 0, d1);
        const uint32_t xo0 = xoff(x0), xo1 = xoff(x1), yo0 = yoff(y0), yo1 = yoff(y1);
        const unsigned long long zo0 = zoff(z0), zo1 = zoff(z1);
        float c00 = lerpf(raw(xo0, yo0, zo0, x0, y0, z0), raw(xo1, yo0, zo0, x1, y0, z0), a);
        float c10 = lerpf(raw(xo0, yo1, zo0, x0, y1, z0), raw(xo1, yo1, zo0, x1, y1, z0), a);
        float c01 = lerpf(raw(xo0, yo0, zo1, x0, y0, z1), raw(xo1, yo0, zo1, x1, y0, z1), a);
        float c11 = lerpf(raw(xo0, yo1, zo1, x0, y1, z1), raw(xo1, yo1, zo1, x1, y1, z1), a);
        return lerpf(lerpf(c00, c10, b), lerpf(c01, c11, b), c) * inv_max;
    }

    // -gradientSobel(vol, pos).xyz (volumeraycast.cl:217-277, :824): 27 taps whole texels from the
    // centre sample, the reference's weights (d = (-1,0,1), s = (1,2,1); x: d[i] s[j] s[k], ...)
    // and loop order.  A rolled loop: the mode is rare and must not cost the others registers.
    VR_DEV f3 neg_sobel(float px, float py, float pz) const
    {
        float ub = px * fw - 0.5f, vb = py * fh - 0.5f, sb = pz * fd - 0.5f;
        float fx = floorf(ub), fy = floorf(vb), fz = floorf(sb);
        float a = ub - fx, b = vb - fy, c = sb - fz;
        int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll 1
        for (int t = 0; t < 27; ++t) {
            const int i = t / 9, j = (t / 3) % 3, k = t % 3;
            const float di = (float)(i - 1), dj = (float)(j - 1), dk = (float)(k - 1);
            const float si = i == 1 ? 2.f : 1.f, sj = j == 1 ? 2.f : 1.f, sk = k == 1 ? 2.f : 1.f;
            const float smp = tri_at(ix - 1 + i, iy - 1 + j, iz - 1 + k, a, b, c);
            gx = gx + ((di * sj) * sk) * smp;
            gy = gy + ((si * dj) * sk) * smp;
            gz = gz + ((si * sj) * dk) * smp;
        }
        f3 g = mk3(gx / 27.f, gy / 27.f, gz / 27.f);
        if (len3(g) == 0.f) g = mk3(1.f, 1.f, 1.f);
        return neg3(normalize3(g));
    }

    // read_imagef(vol, nearestSmp, pos).x -- normalised, CLAMP (border 0), NEAREST
    VR_DEV float nearest(float px, float py, float pz) const
    {
        float fx = floorf(px * fw), fy = floorf(py * fh), fz = floorf(pz * fd);
        if (!(fx >= 0.0f && fx <= (float)w1 && fy >= 0.0f && fy <= (float)h1 && fz >= 0.0f &&
              fz <= (float)d1))
            return 0.0f;
        int x = (int)fx, y = (int)fy, z = (int)fz;
        return raw(xoff(x), yoff(y), zoff(z), x, y, z) * inv_max;
    }
};

// read_imagef(tffData, linearSmp, x) on the float4 table
VR_DEV float4 tff_linear(const float4 *tff, int n, float x)
{
    float ub = x * (float)n - 0.5f;
    float fl = floorf(ub);
    float a = ub - fl;
    int i = (int)fl;
    int i0 = iclamp(i, 0, n - 1), i1 = iclamp(i + 1, 0, n - 1);
    float4 t0 = tff[i0], t1 = tff[i1];
    float4 r;
    r.x = lerpf(t0.x, t1.x, a);
    r.y = lerpf(t0.y, t1.y, a);
    r.z = lerpf(t0.z, t1.z, a);
    r.w = lerpf(t0.w, t1.w, a);
    return r;
}
VR_DEV float tff_linear_alpha(const float4 *tff, int n, float x)
{
    float ub = x * (float)n - 0.5f;
    float fl = floorf(ub);
    float a = ub - fl;
    int i = (int)fl;
    int i0 = iclamp(i, 0, n - 1), i1 = iclamp(i + 1, 0, n - 1);
    return lerpf(tff[i0].w, tff[i1].w, a);
}

// gradientCentralDiffTff (:181-206), un-negated: xyz = normalised difference of the TF opacities
// one texel either side, w = its length.  (illumType 2 and the path tracer.)
template <typename VT, int INSTR, typename V>
VR_DEV float4 gradient_tff(const V &vol, const float4 *s_tff, int tffn, f3 p)
{
    const f3 off = mk3(1.0f / vol.fw, 1.0f / vol.fh, 1.0f / vol.fd);
    f3 s1, s2;
    s1.x = tff_linear_alpha(s_tff, tffn, vol.linear(p.x + (-off.x), p.y + 0.0f, p.z + 0.0f));
    s1.y = tff_linear_alpha(s_tff, tffn, vol.linear(p.x + 0.0f, p.y + (-off.y), p.z + 0.0f));
    s1.z = tff_linear_alpha(s_tff, tffn, vol.linear(p.x + 0.0f, p.y + 0.0f, p.z + (-off.z)));
    s2.x = tff_linear_alpha(s_tff, tffn, vol.linear(p.x + off.x, p.y + 0.0f, p.z + 0.0f));
    s2.y = tff_linear_alpha(s_tff, tffn, vol.linear(p.x + 0.0f, p.y + off.y, p.z + 0.0f));
    s2.z = tff_linear_alpha(s_tff, tffn, vol.linear(p.x + 0.0f, p.y + 0.0f, p.z + off.z));
    const f3 g = sub3(s2, s1);
    f3 n = normalize3(g);
    if (dot3(g, g) == 0.0f) n = mk3(0.57735f, 0.57735f, 0.57735f);
    return make_float4(n.x, n.y, n.z, len3(g));
}

// read_imageui(tffPrefix, nearestSmp, x).x -- border 0 outside [0, n-1]
VR_DEV uint32_t prefix_nearest(const uint32_t *prefix, uint32_t n, float x)
{
    float fi = floorf(x * (float)n);
    if (!(fi >= 0.0f && fi <= (float)(n - 1))) return 0u;
    return prefix[(int)fi];
}

// The reference's per-brick skip test (volumeraycast.cl:777-787) on one (min,max) pair.
VR_DEV bool skip_test(const TfView &tf, float mn, float mx)
{
    float alphaMax = tff_linear_alpha(tf.tff, (int)tf.tff_n, mx);
    if (!(alphaMax < 1e-6f)) return false;
    return prefix_nearest(tf.prefix, tf.prefix_n, mn) == prefix_nearest(tf.prefix, tf.prefix_n, mx);
}

// One bit per brick + one trailing word for out-of-range cells, which the reference reads
// with undefined result and SURVEY A.6 defines as (min,max) = (0,0).
template <typename VT>
__global__ __launch_bounds__(kBlockDim) void vr_skipmap_kernel(BrickView b, float inv_max,
                                                               TfView tf, uint32_t *bits,
                                                               uint32_t n_words)
{
    const size_t n = (size_t)b.bw * b.bh * b.bd;
    const size_t i = (size_t)blockIdx.x * kBlockDim + threadIdx.x;
    bool s = false;
    if (i < n) {
        const VT *p = (const VT *)b.data;
        s = skip_test(tf, (float)p[2 * i] * inv_max, (float)p[2 * i + 1] * inv_max);
    }
    unsigned long long m = __ballot(s);
    if ((threadIdx.x & 63) == 0) {
        size_t w = (i >> 6) * 2;
        if (w < n_words) bits[w] = (uint32_t)m;
        if (w + 1 < n_words) bits[w + 1] = (uint32_t)(m >> 32);
    }
    if (i == 0) bits[n_words] = skip_test(tf, 0.0f, 0.0f) ? 0xffffffffu : 0u;   // every bit
}

// ------------------------------------------------------------------ ray set-up

struct Ray {
    f3 cam, dir;
    float env[4];
    float rnd;
    float tnear, tfar;
    bool hit;
};

// volumeraycast.cl:605-683: RNG jitter, padded-grid NDC, view transform, background, bbox
VR_DEV Ray make_ray(uint32_t gx, uint32_t gy, const FrameView &fr, const vrhip_camera_params &cam,
                    const vrhip_rendering_params &rp, uint32_t seed)
{
    Ray r;
    const float *V = cam.viewMat;
    const f3 ms = mk3(rp.modelScale[0], rp.modelScale[1], rp.modelScale[2]);
    r.rnd = (float)parallel_rng3(gx, gy, seed) / 4294967296.0f;

    const float aspect = fr.ray_aspect;   // min(gsy / gsx, gsx / gsy), from the host (FrameView)
    int maxImg = (int)(fr.gsx > fr.gsy ? fr.gsx : fr.gsy);
    float icx = ((float)(int)gx / (float)maxImg) * 2.f;
    float icy = ((float)(int)gy / (float)maxImg) * 2.f;
    if (fr.gsx > fr.gsy) { icx -= 1.0f; icy -= aspect; }
    else { icx -= aspect; icy -= 1.0f; }
    icy *= -1.f;
    const float psx = fr.ray_psx, psy = fr.ray_psy;   // 2 / gsx, 2 / gsy
    float rnd2 = (float)parallel_rng3(gy, gx, 2u * seed) / 4294967296.0f;
    icx += rnd2 * psx;
    icy += (-r.rnd) * psy;

    f3 npp = mk3(icx, icy, -1.0f);
    f3 rayDir = mk3(dot3(mk3(V[0], V[1], V[2]), npp), dot3(mk3(V[4], V[5], V[6]), npp),
                    dot3(mk3(V[8], V[9], V[10]), npp));
    f3 camPos = mul3(mk3(V[3], V[7], V[11]), ms);
    if (cam.ortho) {
        camPos = mk3(V[3], V[7], V[11]);
        f3 vpx = mk3(V[0], V[4], V[8]);
        f3 vpy = mk3(V[1], V[5], V[9]);
        f3 vpz = mk3(V[2], V[6], V[10]);
        rayDir = neg3(vpz);
        npp = add3(add3(camPos, scale3(vpx, icx)), scale3(vpy, icy));
        npp = scale3(npp, len3(camPos));
        camPos = mul3(npp, ms);
    }
    rayDir = normalize3(mul3(rayDir, ms));
    r.cam = camPos;
    r.dir = rayDir;

    float bgf = rp.useGradient ? (0.7f + 0.5f * rayDir.y) : 1.f;
    for (int i = 0; i < 4; ++i) r.env[i] = rp.backgroundColor[i] * bgf;
    if (fr.env) {
        // environment map (:506-510, :655-656): float RGBA image, linearSmp = normalised
        // coordinates, CLAMP_TO_EDGE, LINEAR; replaces the background colour, alpha included
        const float es = vr_atan2f(rayDir.z, rayDir.x) * (float)(0.5 / 3.14159265358979323846) + 0.5f;
        const float et = vr_acosf(vmax(vmin(rayDir.y, 1.0f), -1.0f)) * (float)(1.0 / 3.14159265358979323846);
        const int ew = (int)fr.env_w, eh = (int)fr.env_h;
        const float ub = es * (float)ew - 0.5f, vb = et * (float)eh - 0.5f;
        const float fx = floorf(ub), fy = floorf(vb);
        const float ea = ub - fx, eb = vb - fy;
        const int x0 = iclamp((int)fx, 0, ew - 1), x1 = iclamp((int)fx + 1, 0, ew - 1);
        const int y0 = iclamp((int)fy, 0, eh - 1), y1 = iclamp((int)fy + 1, 0, eh - 1);
        const float4 t00 = fr.env[(size_t)y0 * ew + x0], t10 = fr.env[(size_t)y0 * ew + x1];
        const float4 t01 = fr.env[(size_t)y1 * ew + x0], t11 = fr.env[(size_t)y1 * ew + x1];
        r.env[0] = lerpf(lerpf(t00.x, t10.x, ea), lerpf(t01.x, t11.x, ea), eb);
        r.env[1] = lerpf(lerpf(t00.y, t10.y, ea), lerpf(t01.y, t11.y, ea), eb);
        r.env[2] = lerpf(lerpf(t00.z, t10.z, ea), lerpf(t01.z, t11.z, ea), eb);
        r.env[3] = lerpf(lerpf(t00.w, t10.w, ea), lerpf(t01.w, t11.w, ea), eb);
    }

    // intersectBBox, volumeraycast.cl:122-142
    float o[3] = {camPos.x, camPos.y, camPos.z}, d[3] = {rayDir.x, rayDir.y, rayDir.z};
    float tmin[3], tmax[3];
    for (int i = 0; i < 3; ++i) {
        float inv = 1.0f / d[i];
        float tbot = inv * (cam.bbox_bl[i] - o[i]);
        float ttop = inv * (cam.bbox_tr[i] - o[i]);
        tmin[i] = vmin(ttop, tbot);
        tmax[i] = vmax(ttop, tbot);
    }
    r.tnear = vmax(vmax(tmin[0], tmin[1]), vmax(tmin[0], tmin[2]));
    r.tfar = vmin(vmin(tmax[0], tmax[1]), vmin(tmax[0], tmax[2]));
    r.hit = (r.tfar > r.tnear) && !(r.tfar < 0);
    return r;
}

VR_DEV unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}


} // namespace
