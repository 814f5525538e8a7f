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

// Footprint volume (VolView::fp): entry (ex, ey, ez) = the 8 voxels (ex - 1 + dx, ey - 1 + dy, ez - 1 + dz),
// edge-clamped, value j = dx + 2 dy + 4 dz; entries in 4x4x4 micro-bricks like the voxels.  A pure stream:
// b N^3 bytes read, 8 b N^3 written (69 GB at 2048^3 UCHAR), so the kernel must be bound by HBM, not by
// address arithmetic or by the load path.
//
// A thread makes one z slice of an output micro-brick: 16 entries = 128 b contiguous bytes, from two voxel
// slices (z - 1, z) of 5 x 5 voxels each.  Per slice it loads the source micro-brick's slice whole (16
// voxels, one 16 b-byte load; four lanes cover a 64 b-byte brick), the slice of the brick on its left (for
// its last column), row 3 of the brick below in y and one voxel of the diagonal one: 8 loads for 16 entries
// where an entry per thread needed 8 byte gathers each (and was bound by instruction issue: 40.6 ms at
// 2048^3, 24 % of the HBM peak).  Clamping is a fix-up of the loaded slices -- columns, rows and slices
// beyond the volume take the last valid one's values -- so there is no slow path.  The wave's 64 x 128 b
// bytes are contiguous (16 bricks along x); they go through LDS once so that every store instruction writes
// 1 KB of whole lines (half-written lines cost a read-modify-write in the L2: +20 %).
constexpr int kFpThreads = 64;   // one wave per workgroup: its LDS staging is its own, no barriers
template <typename VT> struct Vox4;   // four voxels as one load
template <> struct Vox4<uint8_t> { using type = uint32_t; };
template <> struct Vox4<uint16_t> { using type = uint2; };
template <> struct Vox4<float> { using type = uint4; };

template <typename VT>
__global__ __launch_bounds__(kFpThreads) void vr_build_footprint_kernel(VolView vv, int nbz_fp)
{
    constexpr int SV = (int)(16 * sizeof(VT) / 16);   // uint4 per 16-voxel slice: 1 / 2 / 4
    constexpr int C = 8 * (int)sizeof(VT);            // 16-byte chunks a thread makes: 8 / 16 / 32
    __shared__ uint4 s_w[kFpThreads * C];
    const VT *p = (const VT *)vv.data;
    VT *out = (VT *)const_cast<void *>(vv.fp);
    const int lane = (int)threadIdx.x;
    const int lz = lane & 3;
    const int wave_bx0 = (int)blockIdx.x * 16;
    const int bxo = wave_bx0 + (lane >> 2);           // output brick (entries 4 bxo .. 4 bxo + 3)
    const int by = (int)blockIdx.y, bz = (int)blockIdx.z;
    const int nbx = (int)vv.nbx, nby = (int)vv.nby;
    // source bricks, clamped into the volume (a clamped brick's values are replaced by the fix-ups below)
    const int bx = min(bxo, nbx - 1), bxl = max(min(bxo - 1, nbx - 1), 0);
    const int byc = min(by, nby - 1), byl = max(min(by - 1, nby - 1), 0);
    // valid columns / rows of the CURRENT source brick as seen from this output brick: voxel 4 bxo + x
    // exists for x < mx (0: the output brick lies beyond the volume, everything clamps to the left one)
    const int mx = min(max(vv.w - 4 * bxo, 0), 4), my = min(max(vv.h - 4 * by, 0), 4);
    uint4 src[C];   // the thread's 16 entries [ly][lx][j], 128 b bytes
    // UCHAR bricks that need no clamping (all but the volume's faces): the rows stay packed -- a row of a
    // slice IS a dword of the 16-byte load -- and v_perm_b32 cuts the byte pairs out of them: 52 permutes
    // per thread where the compiler's byte-by-byte packing of the generic code below takes ~250 instructions
    const bool plain = sizeof(VT) == 1 && bxo >= 1 && mx == 4 && by >= 1 && my == 4;
    if (sizeof(VT) == 1 && plain) {
        uint32_t o32[32];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
            const int z = min(max(bz * 4 + lz - 1 + dz, 0), vv.d - 1);
            const unsigned long long zoff = (unsigned long long)(z >> 2) * vv.zstride + (unsigned long long)((z & 3) << 4);
            const unsigned long long a_cur = zoff + (unsigned long long)by * vv.ystride + ((unsigned long long)bxo << 6);
            const unsigned long long a_down = a_cur - vv.ystride;
            const uint4 A4 = *reinterpret_cast<const uint4 *>(p + a_cur);
            const uint4 L4 = *reinterpret_cast<const uint4 *>(p + a_cur - 64);
            const uint32_t D1 = *reinterpret_cast<const uint32_t *>(p + a_down + 12);
            const uint32_t dg1 = (uint32_t)p[a_down - 64 + 15];
            const uint32_t Aw[5] = {D1, A4.x, A4.y, A4.z, A4.w};
            const uint32_t Lw[5] = {dg1, L4.x, L4.y, L4.z, L4.w};
            uint32_t Q0[5], Q1[5];   // byte pairs (c0 c1)(c1 c2) and (c2 c3)(c3 c4) of the five-voxel rows
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                Q0[r] = __builtin_amdgcn_perm(Aw[r], Lw[r], r == 0 ? 0x05040400u : 0x05040403u);
                Q1[r] = __builtin_amdgcn_perm(Aw[r], Aw[r], 0x03020201u);
            }
#pragma unroll
            for (int ly = 0; ly < 4; ++ly)
#pragma unroll
                for (int lx = 0; lx < 4; ++lx) {
                    const uint32_t lo = lx < 2 ? Q0[ly] : Q1[ly], hi = lx < 2 ? Q0[ly + 1] : Q1[ly + 1];
                    o32[(ly * 4 + lx) * 2 + dz] = __builtin_amdgcn_perm(hi, lo, (lx & 1) ? 0x07060302u : 0x05040100u);
                }
        }
        __builtin_memcpy(src, o32, sizeof src);
    } else {
    VT o[128];   // [ly][lx][j]
#pragma unroll
    for (int dz = 0; dz < 2; ++dz) {
        const int z = min(max(bz * 4 + lz - 1 + dz, 0), vv.d - 1);   // the slice, clamped in z
        const unsigned long long zoff = (unsigned long long)(z >> 2) * vv.zstride + (unsigned long long)((z & 3) << 4);
        const unsigned long long a_cur = zoff + (unsigned long long)byc * vv.ystride + ((unsigned long long)bx << 6);
        const unsigned long long a_left = zoff + (unsigned long long)byc * vv.ystride + ((unsigned long long)bxl << 6);
        const unsigned long long a_down = zoff + (unsigned long long)byl * vv.ystride + ((unsigned long long)bx << 6);
        const unsigned long long a_diag = zoff + (unsigned long long)byl * vv.ystride + ((unsigned long long)bxl << 6);
        VT A[16], L[16], D[4];   // current slice [y][x]; left brick's slice; row 3 of the brick below
        {
            uint4 t[SV];
#pragma unroll
            for (int q = 0; q < SV; ++q) t[q] = reinterpret_cast<const uint4 *>(p + a_cur)[q];
            __builtin_memcpy(A, t, sizeof A);
#pragma unroll
            for (int q = 0; q < SV; ++q) t[q] = reinterpret_cast<const uint4 *>(p + a_left)[q];
            __builtin_memcpy(L, t, sizeof L);
        }
        {
            const typename Vox4<VT>::type t = *reinterpret_cast<const typename Vox4<VT>::type *>(p + a_down + 12);
            __builtin_memcpy(D, &t, sizeof D);
        }
        const VT dg = p[a_diag + 15];
        // rows y = -1 .. 3 of five voxels x = -1 .. 3 (relative to the output brick's first voxel)
        VT R[5][5];
#pragma unroll
        for (int x = 0; x < 4; ++x) R[0][x + 1] = D[x];
        R[0][0] = dg;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            R[y + 1][0] = L[y * 4 + 3];
#pragma unroll
            for (int x = 0; x < 4; ++x) R[y + 1][x + 1] = A[y * 4 + x];
        }
        // ---- clamping as fix-ups (uniform per brick, rare): what lies beyond the volume repeats the last
        // voxel inside.  x: the left column is voxel 4 bxo - 1 (or voxel 0 for the first brick: clamp of -1)
        // (voxel 4 bxo - 1 itself always exists: entries stop at ex = w, i.e. bxo <= ceil((w + 1) / 4) - 1)
        if (bxo == 0) {
#pragma unroll
            for (int y = 0; y < 5; ++y) R[y][0] = R[y][1];
        }
        if (mx < 4) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
                if (x >= mx) {
#pragma unroll
                    for (int y = 0; y < 5; ++y) R[y][x + 1] = R[y][x];   // (x ascending: the last valid value runs on)
                }
        }
        // y: row -1 of the first brick row clamps to row 0; rows beyond the volume repeat the last one inside
        if (by == 0) {
#pragma unroll
            for (int x = 0; x < 5; ++x) R[0][x] = R[1][x];
        }
        if (my < 4) {
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (y >= my) {
#pragma unroll
                    for (int x = 0; x < 5; ++x) R[y + 1][x] = R[y][x];
                }
        }
#pragma unroll
        for (int ly = 0; ly < 4; ++ly)
#pragma unroll
            for (int lx = 0; lx < 4; ++lx)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) o[(ly * 4 + lx) * 8 + dx + 2 * dy + 4 * dz] = R[ly + dy][lx + dx];
    }
    __builtin_memcpy(src, o, sizeof src);
    }
    // thread t of the wave owns chunks [t C, t C + C) of the wave's 64 C contiguous chunks; chunk k 64 + lane
    // goes out with store k
#pragma unroll
    for (int k = 0; k < C; ++k) s_w[lane * C + k] = src[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (one wave: LDS operations execute in order)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long brick0 = ((unsigned long long)bz * vv.fp_nby + (unsigned long long)by) * vv.fp_nbx +
                                      (unsigned long long)wave_bx0;
    uint4 *dst = reinterpret_cast<uint4 *>(out + brick0 * 512ull);   // 64 entries x 8 values per brick
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int chunk = k * 64 + lane;                  // 4 C chunks per brick
        if (wave_bx0 + chunk / (4 * C) < (int)vv.fp_nbx) dst[chunk] = s_w[chunk];
    }
}

hipError_t vr_launch_build_footprint(const VolView &vol, int format, hipStream_t stream)
{
    const int nbz = (vol.d + 4) >> 2;
    dim3 grid((vol.fp_nbx + 15) / 16, vol.fp_nby, (unsigned)nbz), block(kFpThreads);
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;   // (8192^3 voxels: 2049 bricks per axis)
    switch (format) {
    case VRHIP_UCHAR:
        hipLaunchKernelGGL(vr_build_footprint_kernel<uint8_t>, grid, block, 0, stream, vol, nbz);
        break;
    case VRHIP_USHORT:
        hipLaunchKernelGGL(vr_build_footprint_kernel<uint16_t>, grid, block, 0, stream, vol, nbz);
        break;
    case VRHIP_FLOAT:
        hipLaunchKernelGGL(vr_build_footprint_kernel<float>, grid, block, 0, stream, vol, nbz);
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
