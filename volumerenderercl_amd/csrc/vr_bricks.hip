// vr_bricks.hip -- everything that streams the whole volume: the min/max brick grid for
// object-order empty-space skipping, the dense <-> micro-brick re-tiling used by upload /
// download, and the on-GPU synthetic volume generator.
//
// vr_build_bricks replaces the reference's `generateBricks` kernel
// (/root/reference/src/kernel/volumeraycast.cl:932-961): per brick, min and max of the
// voxel values over [lo, min(lo + vpc, res - 1)) on each axis -- the last voxel plane is
// never included.  The reference launches one work-item per brick,
// each walking its voxels with strided reads; here the volume is streamed once, fully
// coalesced: one workgroup per (brick-row y, brick-row z) pair sweeps the micro-bricks of
// its x-row (64 contiguous voxels per lane and load group), accumulates in registers and
// merges per ESS brick through LDS atomics.
#include <algorithm>

#include "vr_internal.h"

namespace {

constexpr int kThreads = 256;

template <typename VT> struct Key;   // order-preserving map VT -> uint32
template <> struct Key<uint8_t> {
    static __device__ uint32_t enc(uint8_t v) { return v; }
    static __device__ uint8_t dec(uint32_t k) { return (uint8_t)k; }
    static __device__ uint8_t top() { return 255; }          // 1.0 as UNORM8
};
template <> struct Key<uint16_t> {
    static __device__ uint32_t enc(uint16_t v) { return v; }
    static __device__ uint16_t dec(uint32_t k) { return (uint16_t)k; }
    static __device__ uint16_t top() { return 65535; }
};
template <> struct Key<float> {
    static __device__ uint32_t enc(float v)
    {
        uint32_t b = __float_as_uint(v);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    static __device__ float dec(uint32_t k)
    {
        return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
    }
    static __device__ float top() { return 1.0f; }
};

template <typename VT>
__global__ __launch_bounds__(kThreads) void vr_build_bricks_kernel(VolView vol, int tex_x, int tex_y,
                                                                   int vpc_x, int vpc_y, int vpc_z,
                                                                   VT *__restrict__ out)
{
    extern __shared__ uint32_t s_keys[];   // [tex_x] min keys, then [tex_x] max keys
    uint32_t *s_min = s_keys, *s_max = s_keys + tex_x;
    const int cy = blockIdx.x % tex_y, cz = blockIdx.x / tex_y;
    for (int i = threadIdx.x; i < tex_x; i += kThreads) {
        s_min[i] = Key<VT>::enc(Key<VT>::top());   // minVal = 1.f (:946)
        s_max[i] = Key<VT>::enc((VT)0);            // maxVal = 0.f (:945)
    }
    __syncthreads();

    const int w = vol.w, h = vol.h, d = vol.d;
    const int ylo = vpc_y * cy, yhi = min(ylo + vpc_y, h - 1);   // [lo, hi): last plane excluded
    const int zlo = vpc_z * cz, zhi = min(zlo + vpc_z, d - 1);
    const VT *base = (const VT *)vol.data;
    if (yhi > ylo && zhi > zlo) {
        const int my0 = ylo >> 2, my1 = (yhi - 1) >> 2, mz0 = zlo >> 2, mz1 = (zhi - 1) >> 2;
        constexpr int N16 = 64 * (int)sizeof(VT) / 16;   // 16-byte loads per micro-brick
        for (int mx = threadIdx.x; mx < (int)vol.nbx; mx += kThreads) {
            uint32_t kmin[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            uint32_t kmax[4] = {0u, 0u, 0u, 0u};
            for (int mz = mz0; mz <= mz1; ++mz)
                for (int my = my0; my <= my1; ++my) {
                    const uint4 *p = reinterpret_cast<const uint4 *>(
                        base + (unsigned long long)mz * vol.zstride +
                        (unsigned long long)my * vol.ystride + (unsigned long long)mx * 64ull);
                    uint4 q[N16];
#pragma unroll
                    for (int i = 0; i < N16; ++i) q[i] = p[i];
                    VT v[64];
                    __builtin_memcpy(v, q, sizeof v);
#pragma unroll
                    for (int dz = 0; dz < 4; ++dz) {
                        const int z = 4 * mz + dz;
                        if (z < zlo || z >= zhi) continue;   // block-uniform
#pragma unroll
                        for (int dy = 0; dy < 4; ++dy) {
                            const int y = 4 * my + dy;
                            if (y < ylo || y >= yhi) continue;   // block-uniform
#pragma unroll
                            for (int dx = 0; dx < 4; ++dx) {
                                uint32_t k = Key<VT>::enc(v[dz * 16 + dy * 4 + dx]);
                                kmin[dx] = min(kmin[dx], k);
                                kmax[dx] = max(kmax[dx], k);
                            }
                        }
                    }
                }
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int x = 4 * mx + dx;
                const int cx = x / vpc_x;
                const int xhi = min(vpc_x * cx + vpc_x, w - 1);
                if (x < xhi && cx < tex_x && kmin[dx] <= kmax[dx]) {
                    atomicMin(&s_min[cx], kmin[dx]);
                    atomicMax(&s_max[cx], kmax[dx]);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tex_x; i += kThreads) {
        size_t o = 2 * (((size_t)cz * tex_y + cy) * tex_x + i);
        out[o] = Key<VT>::dec(s_min[i]);
        out[o + 1] = Key<VT>::dec(s_max[i]);
    }
}

template <typename VT>
hipError_t build_typed(const VolView &vol, const uint32_t tex[3], void *out, hipStream_t stream)
{
    // voxPerCell = ceil(volDim / brickDim) in fp32 (:940-941)
    int vpc[3];
    const int dim[3] = {vol.w, vol.h, vol.d};
    for (int i = 0; i < 3; ++i) vpc[i] = (int)ceilf((float)dim[i] / (float)tex[i]);
    dim3 grid(tex[1] * tex[2]), block(kThreads);
    size_t lds = 2 * (size_t)tex[0] * sizeof(uint32_t);
    hipLaunchKernelGGL(vr_build_bricks_kernel<VT>, grid, block, lds, stream, vol, (int)tex[0],
                       (int)tex[1], vpc[0], vpc[1], vpc[2], (VT *)out);
    return hipGetLastError();
}

// One thread per micro-brick of the slab [z0, z0 + nz): gathers its 16 rows of 4 voxels from
// the dense array (each wave-load covers 64 lanes x 4 voxels of one row: coalesced) and writes
// the 64 voxels contiguously -- or the inverse.  Voxels outside the volume are written as 0 and
// never read back (texel indices are clamped to the edge).
template <typename VT>
__global__ __launch_bounds__(kThreads) void vr_retile_kernel(VolView vol, VT *dense, int z0, int nz,
                                                             bool to_bricks)
{
    const int mz0 = z0 >> 2, mz_n = ((z0 + nz - 1) >> 2) - mz0 + 1;
    const size_t n = (size_t)vol.nbx * vol.nby * (size_t)mz_n;
    VT *bricks = (VT *)const_cast<void *>(vol.data);
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kThreads) {
        const int mx = (int)(i % vol.nbx);
        const size_t r = i / vol.nbx;
        const int my = (int)(r % vol.nby), mz = mz0 + (int)(r / vol.nby);
        VT *b = bricks + (unsigned long long)mz * vol.zstride + (unsigned long long)my * vol.ystride +
                (unsigned long long)mx * 64ull;
#pragma unroll
        for (int dz = 0; dz < 4; ++dz)
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const int x = 4 * mx + dx, y = 4 * my + dy, z = 4 * mz + dz;
                    const bool in = x < vol.w && y < vol.h && z >= z0 && z < z0 + nz;
                    const size_t di = ((size_t)(z - z0) * vol.h + y) * vol.w + x;
                    if (to_bricks) {
                        if (in) b[dz * 16 + dy * 4 + dx] = dense[di];
                        else if (z >= vol.d || y >= vol.h || x >= vol.w) b[dz * 16 + dy * 4 + dx] = (VT)0;
                    } else if (in) {
                        dense[di] = b[dz * 16 + dy * 4 + dx];
                    }
                }
    }
}

// SURVEY 8(d) synthetic fields: p = 2(i+0.5)/N - 1; sphere d = max(0, 1-|p|/0.9);
// shells = d*(0.5+0.5cos(24 pi |p|)) with values < 0.35 zeroed.  One thread per micro-brick.
template <typename VT>
__global__ __launch_bounds__(kThreads) void vr_synth_kernel(VolView vol, int kind)
{
    const size_t n = (size_t)vol.nbx * vol.nby * vol.nbz;
    VT *bricks = (VT *)const_cast<void *>(vol.data);
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kThreads) {
        const int mx = (int)(i % vol.nbx);
        const size_t r = i / vol.nbx;
        const int my = (int)(r % vol.nby), mz = (int)(r / vol.nby);
        VT v[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int x = 4 * mx + (j & 3), y = 4 * my + ((j >> 2) & 3), z = 4 * mz + (j >> 4);
            double px = 2.0 * (x + 0.5) / vol.w - 1.0;
            double py = 2.0 * (y + 0.5) / vol.h - 1.0;
            double pz = 2.0 * (z + 0.5) / vol.d - 1.0;
            double rr = sqrt(px * px + py * py + pz * pz);
            double dv = 1.0 - rr / 0.9;
            if (dv < 0.0) dv = 0.0;
            if (kind == 1) {
                dv = dv * (0.5 + 0.5 * cos(24.0 * 3.14159265358979323846 * rr));
                if (dv < 0.35) dv = 0.0;
            }
            if (x >= vol.w || y >= vol.h || z >= vol.d) dv = 0.0;
            if (sizeof(VT) == 1) v[j] = (VT)llround(255.0 * dv);
            else if (sizeof(VT) == 2) v[j] = (VT)llround(65535.0 * dv);
            else v[j] = (VT)dv;
        }
        uint4 q[64 * sizeof(VT) / 16];
        __builtin_memcpy(q, v, sizeof v);
        uint4 *dst = reinterpret_cast<uint4 *>(bricks + i * 64ull);
#pragma unroll
        for (size_t k = 0; k < 64 * sizeof(VT) / 16; ++k) dst[k] = q[k];
    }
}

} // namespace

// downsampling (volumeraycast.cl:966-994): one thread per low-res voxel, the box summed in the
// reference's k, j, i order; dense x-fastest output in the volume's type.
template <typename VT>
__global__ __launch_bounds__(kThreads) void vr_downsample_kernel(VolView vol, int lx, int ly, int lz,
                                                                 int vx, int vy, int vz, VT *out)
{
    const size_t n = (size_t)lx * ly * lz;
    // grid-stride: a launch cannot have more than 2^32 threads
    for (size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x; o < n; o += (size_t)gridDim.x * kThreads) {
    const int cx = (int)(o % (size_t)lx), cy = (int)((o / (size_t)lx) % (size_t)ly);
    const int cz = (int)(o / ((size_t)lx * ly));
    const int x0 = vx * cx, y0 = vy * cy, z0 = vz * cz;
    const int x1 = min(x0 + vx, vol.w), y1 = min(y0 + vy, vol.h), z1 = min(z0 + vz, vol.d);
    const VT *p = (const VT *)vol.data;
    float value = 0.f;
    for (int k = z0; k < z1; ++k)
        for (int j = y0; j < y1; ++j)
            for (int i = x0; i < x1; ++i) value += (float)p[vr_voxel_index(vol, i, j, k)] * vol.inv_max;
    value /= (float)(vx * vy * vz);
    if (sizeof(VT) == 4) {
        reinterpret_cast<float *>(out)[o] = value;
    } else {
        const float top = sizeof(VT) == 1 ? 255.0f : 65535.0f;
        const float q = rintf(fminf(fmaxf(value * top, 0.f), top));   // convert_*_sat_rte
        out[o] = (VT)q;
    }
    }
}

// Footprint volume: entries in memory order (coalesced 8/16/32-byte stores), grid-stride (a launch
// is limited to 2^32 threads; 2048^3 has 8.6e9 entries); the eight source voxels come from at
// most eight micro-bricks of the volume (cached gathers).
template <typename VT>
__global__ __launch_bounds__(kThreads) void vr_build_footprint_kernel(VolView vv)
{
    const unsigned long long n = (unsigned long long)vv.fp_nbx * vv.fp_nby *
                                 (unsigned long long)((vv.d + 4) >> 2) * 64ull;
    const unsigned long long stride = (unsigned long long)gridDim.x * kThreads;
    const VT *p = (const VT *)vv.data;
    VT *out = (VT *)const_cast<void *>(vv.fp);
    for (unsigned long long i = (unsigned long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const unsigned long long brick = i >> 6;
        const int in = (int)(i & 63ull);
        const int bx = (int)(brick % vv.fp_nbx), by = (int)((brick / vv.fp_nbx) % vv.fp_nby);
        const int bz = (int)(brick / ((unsigned long long)vv.fp_nbx * vv.fp_nby));
        const int ex = bx * 4 + (in & 3), ey = by * 4 + ((in >> 2) & 3), ez = bz * 4 + (in >> 4);
        VT e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = min(max(ex - 1 + (j & 1), 0), vv.w - 1);
            const int y = min(max(ey - 1 + ((j >> 1) & 1), 0), vv.h - 1);
            const int z = min(max(ez - 1 + (j >> 2), 0), vv.d - 1);
            e[j] = p[vr_voxel_index(vv, x, y, z)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) out[i * 8ull + j] = e[j];
    }
}

hipError_t vr_launch_build_footprint(const VolView &vol, int format, hipStream_t stream)
{
    const unsigned long long n = (unsigned long long)vol.fp_nbx * vol.fp_nby *
                                 (unsigned long long)((vol.d + 4) >> 2) * 64ull;
    dim3 grid((unsigned)std::min<unsigned long long>((n + kThreads - 1) / kThreads, 1ull << 22)), block(kThreads);
    switch (format) {
    case VRHIP_UCHAR:
        hipLaunchKernelGGL(vr_build_footprint_kernel<uint8_t>, grid, block, 0, stream, vol);
        break;
    case VRHIP_USHORT:
        hipLaunchKernelGGL(vr_build_footprint_kernel<uint16_t>, grid, block, 0, stream, vol);
        break;
    case VRHIP_FLOAT:
        hipLaunchKernelGGL(vr_build_footprint_kernel<float>, grid, block, 0, stream, vol);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t vr_launch_downsample(const VolView &vol, int format, const int lo[3], const int vpc[3],
                                void *out, hipStream_t stream)
{
    const size_t n = (size_t)lo[0] * lo[1] * lo[2];
    dim3 grid((unsigned)std::min<size_t>((n + kThreads - 1) / kThreads, (size_t)1 << 22)), block(kThreads);
    switch (format) {
    case VRHIP_UCHAR:
        hipLaunchKernelGGL(vr_downsample_kernel<uint8_t>, grid, block, 0, stream, vol, lo[0], lo[1], lo[2],
                           vpc[0], vpc[1], vpc[2], (uint8_t *)out);
        break;
    case VRHIP_USHORT:
        hipLaunchKernelGGL(vr_downsample_kernel<uint16_t>, grid, block, 0, stream, vol, lo[0], lo[1], lo[2],
                           vpc[0], vpc[1], vpc[2], (uint16_t *)out);
        break;
    case VRHIP_FLOAT:
        hipLaunchKernelGGL(vr_downsample_kernel<float>, grid, block, 0, stream, vol, lo[0], lo[1], lo[2],
                           vpc[0], vpc[1], vpc[2], (float *)out);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t vr_launch_build_bricks(const VolView &vol, int format, const uint32_t tex[3],
                                  void *bricks_out, hipStream_t stream)
{
    switch (format) {
    case VRHIP_UCHAR: return build_typed<uint8_t>(vol, tex, bricks_out, stream);
    case VRHIP_USHORT: return build_typed<uint16_t>(vol, tex, bricks_out, stream);
    case VRHIP_FLOAT: return build_typed<float>(vol, tex, bricks_out, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t vr_launch_synth(int kind, const VolView &vol, int format, hipStream_t stream)
{
    dim3 grid(256 * 16), block(kThreads);
    switch (format) {
    case VRHIP_UCHAR: hipLaunchKernelGGL(vr_synth_kernel<uint8_t>, grid, block, 0, stream, vol, kind); break;
    case VRHIP_USHORT: hipLaunchKernelGGL(vr_synth_kernel<uint16_t>, grid, block, 0, stream, vol, kind); break;
    case VRHIP_FLOAT: hipLaunchKernelGGL(vr_synth_kernel<float>, grid, block, 0, stream, vol, kind); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t vr_launch_retile(const VolView &vol, int format, const void *dense, int z0, int nz,
                            bool to_bricks, hipStream_t stream)
{
    if (nz <= 0) return hipSuccess;
    const size_t n = (size_t)vol.nbx * vol.nby * (size_t)(((z0 + nz - 1) >> 2) - (z0 >> 2) + 1);
    unsigned blocks = (unsigned)std::min<size_t>((n + kThreads - 1) / kThreads, 256 * 32);
    dim3 grid(blocks), block(kThreads);
    void *dn = const_cast<void *>(dense);
    switch (format) {
    case VRHIP_UCHAR:
        hipLaunchKernelGGL(vr_retile_kernel<uint8_t>, grid, block, 0, stream, vol, (uint8_t *)dn, z0, nz, to_bricks);
        break;
    case VRHIP_USHORT:
        hipLaunchKernelGGL(vr_retile_kernel<uint16_t>, grid, block, 0, stream, vol, (uint16_t *)dn, z0, nz, to_bricks);
        break;
    case VRHIP_FLOAT:
        hipLaunchKernelGGL(vr_retile_kernel<float>, grid, block, 0, stream, vol, (float *)dn, z0, nz, to_bricks);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
