// vr_raycast.hip -- the per-pixel front-to-back ray march for gfx950 (CDNA4).
//
// Replaces the reference's OpenCL kernel `volumeRender`
// (/root/reference/src/kernel/volumeraycast.cl:589-926).  CDNA has no image/sampler
// hardware (__HIP_NO_IMAGE_SUPPORT), so every read_imagef of the reference is restated
// as explicit address arithmetic + loads following the OpenCL 1.2 image rules
// (SURVEY.md App. B).
//
// Execution design (DESIGN.md "Kernels"):
//  * one lane per pixel, one wave64 per 8x8-pixel patch (the reference's work-group);
//  * PERSISTENT workgroups (4 waves), sized to fill the 256 CUs once; every wave pulls
//    8x8 patches from a global queue ordered centre-first, so the expensive rays start
//    first and the tail is filled with cheap ones (dynamic load balance);
//  * the transfer function (float4 table) and the ESS skip bitmap (1 bit per brick,
//    precomputed from bricks + TF + prefix sum) live in LDS: a DDA step over an empty
//    brick touches no global memory at all;
//  * the reference's nested loops (DDA over bricks / samples inside a brick) are flattened
//    into a per-lane state machine driven by wave ballots: in every round all lanes that
//    have a sample to take take it together; lanes that need brick steps take a bounded
//    number of them first.  The per-ray sequence of t values, and therefore the image, is
//    exactly the reference's.
#include <cstdio>
#include <cstdlib>

#include "vr_device_math.h"
#include "vr_internal.h"

namespace {

constexpr int kBlockDim = 256;       // 4 waves
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
__device__ unsigned long long g_stamps[16];
#define VR_STAMP_DECL unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = vr_stamp(), st_first = st_last
#define VR_STAMP(i) do { unsigned long long n_ = vr_stamp(); st_acc[i] += n_ - st_last; st_last = n_; } while (0)
#define VR_COUNT(i) st_acc[i] += 1
#define VR_STAMP_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_stamps[i_], st_acc[i_]); atomicAdd(&g_stamps[12], vr_stamp() - st_first); } } while (0)
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
#define VR_STAMP_FLUSH
#endif

// ------------------------------------------------------------------ volume reads

template <typename VT, int INSTR>
struct Vol {
    const VT *p;
    int w1, h1, d1;   // res - 1
    float fw, fh, fd;
    float inv_max;
    uint32_t nbx, nby, ystride;     // micro-brick layout (vr_internal.h)
    unsigned long long zstride;
    uint32_t *touched;

    // per-axis parts of the element index of voxel (x, y, z) in the 4x4x4 micro-brick layout
    VR_DEV uint32_t xoff(int x) const { return ((uint32_t)(x >> 2) << 6) + (uint32_t)(x & 3); }
    VR_DEV uint32_t yoff(int y) const
    {
        return __umul24((uint32_t)(y >> 2), ystride) + ((uint32_t)(y & 3) << 2);
    }
    VR_DEV unsigned long long zoff(int z) const
    {
        return (unsigned long long)(uint32_t)(z >> 2) * zstride + (unsigned long long)((z & 3) << 4);
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
        int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        int x0 = iclamp(ix, 0, w1), x1 = iclamp(ix + 1, 0, w1);
        int y0 = iclamp(iy, 0, h1), y1 = iclamp(iy + 1, 0, h1);
        int z0 = iclamp(iz, 0, d1), z1 = iclamp(iz + 1, 0, d1);
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
        int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        int X[4], Y[4], Z[4];
        uint32_t xo[4], yo[4];
        unsigned long long zo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            X[k] = iclamp(ix - 1 + k, 0, w1);
            Y[k] = iclamp(iy - 1 + k, 0, h1);
            Z[k] = iclamp(iz - 1 + k, 0, d1);
            xo[k] = xoff(X[k]);
            yo[k] = yoff(Y[k]);
            zo[k] = zoff(Z[k]);
        }
#define VR_L(xi, yi, zi) raw(xo[xi], yo[yi], zo[zi], X[xi], Y[yi], Z[zi])
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
                    const vrhip_rendering_params &rp)
{
    Ray r;
    const float *V = cam.viewMat;
    const f3 ms = mk3(rp.modelScale[0], rp.modelScale[1], rp.modelScale[2]);
    r.rnd = (float)parallel_rng3(gx, gy, rp.seed) / 4294967296.0f;

    float gsx = (float)fr.gsx, gsy = (float)fr.gsy;
    float aspect = gsy / gsx;
    aspect = vmin(aspect, gsx / gsy);
    int maxImg = (int)(fr.gsx > fr.gsy ? fr.gsx : fr.gsy);
    float icx = ((float)(int)gx / (float)maxImg) * 2.f;
    float icy = ((float)(int)gy / (float)maxImg) * 2.f;
    if (fr.gsx > fr.gsy) { icx -= 1.0f; icy -= aspect; }
    else { icx -= aspect; icy -= 1.0f; }
    icy *= -1.f;
    float psx = 2.f / gsx, psy = 2.f / gsy;
    float rnd2 = (float)parallel_rng3(gy, gx, 2u * rp.seed) / 4294967296.0f;
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

// ------------------------------------------------------------------ ray cast

template <typename VT, bool ESS, int INSTR, bool SKIP_LDS>
__global__ __launch_bounds__(kBlockDim) void vr_raycast_kernel(
    VolView vv, BrickView bricks, TfView tf, SkipView skip, FrameView fr, vrhip_camera_params cam,
    vrhip_rendering_params rp, vrhip_raycast_params rc, DevStats *stats, uint32_t *touched)
{
    // LDS: [tff_n float4][skip words + 1]
    extern __shared__ float4 s_mem[];
    VR_STAMP_DECL;
    float4 *s_tff = s_mem;
    uint32_t *s_skip = reinterpret_cast<uint32_t *>(s_mem + tf.tff_n);
    for (uint32_t i = threadIdx.x; i < tf.tff_n; i += kBlockDim) s_tff[i] = tf.tff[i];
    if (ESS && SKIP_LDS)
        for (uint32_t i = threadIdx.x; i <= skip.n_words; i += kBlockDim) s_skip[i] = skip.bits[i];
    __syncthreads();
    VR_STAMP(8);

    const uint32_t lane = threadIdx.x & 63u;
    const int tffn = (int)tf.tff_n;
    unsigned long long c_taken = 0, c_nominal = 0, c_shaded = 0, c_bricks = 0, c_skipped = 0,
                       c_hit = 0;

    Vol<VT, INSTR> vol;
    vol.p = (const VT *)vv.data;
    vol.w1 = vv.w - 1; vol.h1 = vv.h - 1; vol.d1 = vv.d - 1;
    vol.fw = vv.fw; vol.fh = vv.fh; vol.fd = vv.fd;
    vol.inv_max = vv.inv_max;
    vol.nbx = vv.nbx; vol.nby = vv.nby;
    vol.ystride = vv.ystride; vol.zstride = vv.zstride;
    vol.touched = touched;
    const f3 voxLen = mk3(1.f / vol.fw, 1.f / vol.fh, 1.f / vol.fd);
    const float refInterval = 1.f / rc.samplingRate;
    const int bw = bricks.bw, bh = bricks.bh, bd = bricks.bd;
    float brickLen[3] = {0.f, 0.f, 0.f}, brickDia = 0.f;
    if (ESS) {
        for (int i = 0; i < 3; ++i) brickLen[i] = 1.f / rc.brickRes[i];
        brickDia = sqrtf(((brickLen[0] * brickLen[0]) + (brickLen[1] * brickLen[1])) +
                         (brickLen[2] * brickLen[2])) * 2.f;
    }

    // every wave pulls 8x8 patches until the queue is drained (exit condition reached by
    // every wave: the head only grows)
    // The next ticket is drawn while the current patch is marched, so the ~microseconds of the
    // contended atomic are hidden; every wave draws exactly one ticket past the end.
    uint32_t q_next = 0;
    if (lane == 0) q_next = atomicAdd(fr.queue_head, 1u);
    for (;;) {
        const uint32_t q = __builtin_amdgcn_readfirstlane(q_next);
        if (q >= fr.n_wave_tiles) break;
        const WaveTile wt = fr.queue[q];
        if (lane == 0) q_next = atomicAdd(fr.queue_head, 1u);
        VR_STAMP(0);
        VR_COUNT(11);
        const uint32_t lx = lane & 7u, ly = lane >> 3;
        const uint32_t gx = (uint32_t)wt.tx8 * 8u + lx, gy = (uint32_t)wt.ty8 * 8u + ly;
        const bool inside = gx < fr.W && gy < fr.H;

        const Ray ray = make_ray(gx, gy, fr, cam, rp);
        float res0 = ray.env[0], res1 = ray.env[1], res2 = ray.env[2], alpha = 0.f;
        float tnear = ray.tnear;
        const float tfar = ray.tfar;
        const float sampleDist = tfar - tnear;
        const f3 camPos = ray.cam, rayDir = ray.dir;

        int state = S_DONE;
        float t = 0.f, t_exit = tfar, stepSize = 0.f, offset = 0.f;
        int stepv[3] = {0, 0, 0}, cell[3] = {0, 0, 0}, exitc[3] = {0, 0, 0};
        float tv[3] = {0, 0, 0}, deltaT[3] = {0, 0, 0};
        uint32_t cidx = 0, skw = 0;   // linear index of the current cell and its bitmap word
        // per-ray invariants of illumination()/specularBlinnPhong() (:280-303)
        const f3 toLight = neg3(rayDir);
        const f3 lgt = normalize3(toLight);
        f3 hv = add3(toLight, lgt);
        const bool hvalid = !(dot3(hv, hv) < 1.e-6f);
        hv = normalize3(hv);

        if (inside && ray.hit && sampleDist > 0.f) {
            if (INSTR) c_hit++;
            // volumeraycast.cl:709-733
            f3 resf = mk3(vol.fw, vol.fh, vol.fd);
            stepSize = vmin(sampleDist, sampleDist / (rc.samplingRate *
                                                      len3(mul3(scale3(rayDir, sampleDist), resf))));
            float samples = ceilf(sampleDist / stepSize);
            stepSize = sampleDist / samples;
            if (INSTR) c_nominal += (unsigned long long)samples;
            tnear = vmax(0.f, tnear);
            t = tnear;
            offset = (len3(voxLen) * ray.rnd) * 2.0f;
            state = ESS ? S_BRICK : S_SAMPLE;
            if (ESS) {   // 3-D DDA set-up (:737-760)
                const int bres[3] = {bw, bh, bd};
                const float dirv[3] = {rayDir.x, rayDir.y, rayDir.z};
                const float camv[3] = {camPos.x, camPos.y, camPos.z};
                for (int i = 0; i < 3; ++i) {
                    float invRay = 1.f / dirv[i];
                    stepv[i] = dirv[i] > 0.f ? 1 : (dirv[i] < 0.f ? -1 : 0);
                    deltaT[i] = (float)stepv[i] * ((brickLen[i] * 2.f) * invRay);
                    float roc = (camv[i] + dirv[i] * tnear) - (-1.f);
                    cell[i] = iclamp((int)floorf(roc / (2.f * brickLen[i])), 0, bres[i] - 1);
                    int cadj = cell[i] - (dirv[i] >= 0.f ? -1 : 0);
                    tv[i] = tnear + ((float)cadj * (2.f * brickLen[i]) - roc) * invRay;
                    exitc[i] = stepv[i] * bres[i];
                    if (exitc[i] < 0) exitc[i] = -1;
                }
                // bitmap word of the start cell (always inside the grid after the clamp)
                cidx = __umul24(__umul24((uint32_t)cell[2], (uint32_t)bh) + (uint32_t)cell[1],
                                (uint32_t)bw) + (uint32_t)cell[0];
                skw = (SKIP_LDS ? s_skip : skip.bits)[cidx >> 5];
            }
        }

        VR_STAMP(1);
        // ---- flattened DDA / sample state machine
        for (;;) {
            VR_COUNT(10);
            if (ESS) {
                for (int it = 0;; ++it) {
                    const bool inB = state == S_BRICK;
                    if (!__ballot(inB)) break;
                    if (it >= kMaxBrickSteps && __ballot(state == S_SAMPLE)) break;
                    VR_COUNT(9);
                    // One DDA step (:763-787) as BRANCH-FREE predicated code: a lone wave pays
                    // an instruction-buffer refill per taken branch, and this loop used to be
                    // mostly branches.  Every lane computes the step; `go` (lane is in S_BRICK
                    // and passes the outer loop condition t < tfar) gates what is committed.
                    const bool go = inB && (t < tfar);
                    // decision for the current cell: its bitmap word was fetched one step
                    // ahead (skw), so the LDS latency overlaps the step arithmetic
                    const bool skp = (skw >> (cidx & 31u)) & 1u;
                    const bool m0 = (tv[0] <= tv[1]) && (tv[0] <= tv[2]);
                    const bool m1 = (tv[1] <= tv[0]) && (tv[1] <= tv[2]);
                    const bool m2 = (tv[2] <= tv[0]) && (tv[2] <= tv[1]);
                    const float inc0 = m0 ? 1.f : 0.f, inc1 = m1 ? 1.f : 0.f, inc2 = m2 ? 1.f : 0.f;
                    float te = ((tv[0] * inc0) + (tv[1] * inc1)) + (tv[2] * inc2);
                    te = vclamp(te, t + stepSize, t + brickDia);
                    cell[0] += (go && m0) ? stepv[0] : 0;
                    cell[1] += (go && m1) ? stepv[1] : 0;
                    cell[2] += (go && m2) ? stepv[2] : 0;
                    tv[0] = go ? tv[0] + inc0 * deltaT[0] : tv[0];
                    tv[1] = go ? tv[1] + inc1 * deltaT[1] : tv[1];
                    tv[2] = go ? tv[2] + inc2 * deltaT[2] : tv[2];
                    t_exit = go ? te : t_exit;
                    // fetch the word of the cell just entered (out-of-range cells read the
                    // trailing word, which holds the (0,0) decision in every bit)
                    {
                        const uint32_t *sb = SKIP_LDS ? s_skip : skip.bits;
                        const bool oob = (uint32_t)cell[0] >= (uint32_t)bw ||
                                         (uint32_t)cell[1] >= (uint32_t)bh ||
                                         (uint32_t)cell[2] >= (uint32_t)bd;
                        cidx = __umul24(__umul24((uint32_t)cell[2], (uint32_t)bh) +
                                            (uint32_t)cell[1], (uint32_t)bw) + (uint32_t)cell[0];
                        skw = sb[oob ? skip.n_words : (cidx >> 5)];
                    }
                    if (INSTR) { c_bricks += go ? 1 : 0; c_skipped += (go && skp) ? 1 : 0; }
                    t = (go && skp) ? te : t;   // :784-785 `continue`
                    state = inB ? (go ? (skp ? S_BRICK : S_SAMPLE) : S_DONE) : state;
                }
            }
            VR_STAMP(2);
            if (state == S_SAMPLE) {
                // ---- up to kBatch consecutive samples of this ray per round (inner loop,
                // :790-880).  Colour and opacity of a sample do not depend on the running
                // alpha, so the batch is evaluated as independent straight-line code (loads of
                // all its fetches in flight together) and only the cheap front-to-back
                // compositing below is sequential.  Samples past ERT / t_exit are speculative:
                // fetched from clamped (always valid) addresses and never composited.
                float tk[kBatch];
                bool vk[kBatch], litk[kBatch];
                tk[0] = t;
                vk[0] = t < t_exit;   // inner loop condition (:790)
#pragma unroll
                for (int k = 1; k < kBatch; ++k) {
                    tk[k] = tk[k - 1] + stepSize;                                   // :879
                    vk[k] = vk[k - 1] && !(tk[k - 1] >= tfar) && (tk[k] < t_exit);  // :868, :790
                    // the traffic-instrumented variant must not touch speculative voxels
                    if (INSTR == 2) vk[k] = false;
                }
                f3 pk[kBatch];
                float dens[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    f3 pos = add3(camPos, scale3(rayDir, tk[k] - offset));
                    pk[k] = mk3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f);
                    dens[k] = 0.f;
                }
                if (rp.useLinear) {
#pragma unroll
                    for (int k = 0; k < kBatch; ++k)
                        if (INSTR != 2 || vk[k]) dens[k] = vol.linear(pk[k].x, pk[k].y, pk[k].z);
                } else {
#pragma unroll
                    for (int k = 0; k < kBatch; ++k)
                        if (INSTR != 2 || vk[k]) dens[k] = vol.nearest(pk[k].x, pk[k].y, pk[k].z);
                }
                VR_STAMP(3);
                float4 tfc[kBatch];
                float opk[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k) tfc[k] = tff_linear(s_tff, tffn, dens[k]);
                VR_STAMP(4);
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    f3 grad = mk3(0.f, 0.f, 0.f);
                    const bool lit = vk[k] && tfc[k].w > 0.1f;
                    litk[k] = lit && rp.illumType == 1;
                    if (lit && (rp.illumType == 1 || (rc.contours && !rp.illumType)))
                        grad = vol.neg_gradient(pk[k].x, pk[k].y, pk[k].z);
                    if (lit && rp.illumType == 1) {
                        // illumination (:294-303)
                        float ndl = vmax(0.f, dot3(grad, lgt));
                        float sp = hvalid ? vr_powr(vmax(dot3(grad, hv), 0.f), 40.f) : 0.0f;
                        sp = sp * 0.15f;
                        tfc[k].x = ((tfc[k].x * 0.15f) + ((tfc[k].x * ndl) * 0.7f)) + sp;
                        tfc[k].y = ((tfc[k].y * 0.15f) + ((tfc[k].y * ndl) * 0.7f)) + sp;
                        tfc[k].z = ((tfc[k].z * 0.15f) + ((tfc[k].z * ndl) * 0.7f)) + sp;
                    }
                    if (lit && rc.contours) {
                        float e = fabsf(dot3(rayDir, grad));
                        tfc[k].x *= e; tfc[k].y *= e; tfc[k].z *= e;
                    }
                    tfc[k].x = ray.env[0] - tfc[k].x;
                    tfc[k].y = ray.env[1] - tfc[k].y;
                    tfc[k].z = ray.env[2] - tfc[k].z;
                    if (rc.aerial) {
                        float depthCue = 1.f - (tk[k] - tnear) / sampleDist;
                        tfc[k].w *= depthCue;
                    }
                }
                VR_STAMP(5);
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    // opacity correction (:864).  alpha == 0 gives 1 - powr(1, y) == 0 exactly,
                    // so the whole wave skips the powr when no lane has a non-zero alpha.
                    opk[k] = 0.f;
                    if (__ballot(vk[k] && tfc[k].w != 0.f))
                        opk[k] = 1.f - vr_powr(1.f - tfc[k].w, refInterval);
                }
                // sequential front-to-back compositing (:865-879)
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    if (vk[k] && state == S_SAMPLE) {
                        if (INSTR) { c_taken++; if (litk[k]) c_shaded++; }
                        float oma = 1.f - alpha;
                        res0 = res0 - (tfc[k].x * opk[k]) * oma;
                        res1 = res1 - (tfc[k].y * opk[k]) * oma;
                        res2 = res2 - (tfc[k].z * opk[k]) * oma;
                        alpha = alpha + opk[k] * oma;
                        // (double)alpha > 0.98 <=> alpha >= 0.98f (ERT_THRESHOLD, :28)
                        if (tk[k] >= tfar || alpha >= 0.98f) state = S_DONE;   // break; :882 breaks
                        else t = tk[k] + stepSize;
                    }
                }
                if (state == S_SAMPLE && !(t < t_exit)) {   // inner loop left by its condition
                    if (!ESS) state = S_DONE;
                    else if (t >= tfar || alpha >= 0.98f) state = S_DONE;                  // :882
                    else if (cell[0] == exitc[0] || cell[1] == exitc[1] || cell[2] == exitc[2])
                        state = S_DONE;                                                     // :883
                    else { t = t_exit; state = S_BRICK; }                                   // :884
                }
                VR_STAMP(6);
            }
            if (!__ballot(state != S_DONE)) break;
        }

        if (inside) {
            // running mean over iterations (:898-909), fp32 accumulate buffer
            const size_t fi = (size_t)gy * fr.W + gx;
            if (rp.iteration != 0 && ray.hit && sampleDist > 0.f) {
                float4 prev = fr.fb[fi];
                float it1 = (float)(rp.iteration + 1u);
                res0 = prev.x + (res0 - prev.x) / it1;
                res1 = prev.y + (res1 - prev.y) / it1;
                res2 = prev.z + (res2 - prev.z) / it1;
            }
            float4 o = make_float4(res0, res1, res2, (ray.hit && sampleDist > 0.f) ? alpha : ray.env[3]);
            fr.fb[fi] = o;
            if (fr.out) fr.out[(size_t)wt.out_base + (size_t)ly * fr.out_stride + lx] = o;
        }
        VR_STAMP(7);
    }
    VR_STAMP_FLUSH;

    if (INSTR) {
        unsigned long long c[6] = {c_taken, c_nominal, c_shaded, c_bricks, c_skipped, c_hit};
        for (int i = 0; i < 6; ++i) {
            unsigned long long s = wave_sum(c[i]);
            if (lane == 0 && s) atomicAdd(&stats->v[i], s);
        }
    }
}

template <typename K>
int blocks_per_cu(K kernel, size_t lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlockDim, lds) != hipSuccess ||
        nb < 1)
        nb = 1;
    return nb;
}

template <typename VT, bool ESS, int INSTR, bool SKIP_LDS>
hipError_t launch_variant(const RaycastLaunch &a, hipStream_t stream)
{
    auto kernel = vr_raycast_kernel<VT, ESS, INSTR, SKIP_LDS>;
    size_t lds = (size_t)a.tf.tff_n * sizeof(float4);
    if (ESS && SKIP_LDS) lds += ((size_t)a.skip.n_words + 1) * sizeof(uint32_t);
    static int cached_nb = 0;
    static size_t cached_lds = ~(size_t)0;
    if (cached_lds != lds) {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        cached_nb = blocks_per_cu(kernel, lds);
        if (const char *e = getenv("VRHIP_BLOCKS_PER_CU")) {   // tuning / experiments
            int v = atoi(e);
            if (v > 0) cached_nb = v;
        }
        if (getenv("VRHIP_DEBUG"))
            fprintf(stderr, "[vrhip] raycast variant: lds=%zu B, blocks/CU=%d, CUs=%d\n", lds,
                    cached_nb, a.num_cus);
        cached_lds = lds;
    }
    uint32_t want = (a.frame.n_wave_tiles + 3u) / 4u;
    uint32_t cap = (uint32_t)(a.num_cus > 0 ? a.num_cus : 256) * (uint32_t)cached_nb;
    dim3 grid(want < cap ? want : cap), block(kBlockDim);
    if (grid.x == 0) return hipSuccess;
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, a.vol, a.bricks, a.tf, a.skip, a.frame,
                       a.cam, a.render, a.raycast, a.stats, a.touched);
    return hipGetLastError();
}

template <typename VT>
hipError_t launch_typed(const RaycastLaunch &a, hipStream_t stream)
{
    const bool lds = a.skip.in_lds != 0;
    if (a.use_ess) {
        if (lds) {
            if (a.instr == 0) return launch_variant<VT, true, 0, true>(a, stream);
            if (a.instr == 1) return launch_variant<VT, true, 1, true>(a, stream);
            return launch_variant<VT, true, 2, true>(a, stream);
        }
        if (a.instr == 0) return launch_variant<VT, true, 0, false>(a, stream);
        if (a.instr == 1) return launch_variant<VT, true, 1, false>(a, stream);
        return launch_variant<VT, true, 2, false>(a, stream);
    }
    if (a.instr == 0) return launch_variant<VT, false, 0, false>(a, stream);
    if (a.instr == 1) return launch_variant<VT, false, 1, false>(a, stream);
    return launch_variant<VT, false, 2, false>(a, stream);
}

} // namespace

#ifdef VR_STAMPS
// diagnostic builds only: read (and optionally clear) the per-phase cycle totals
extern "C" int vrhip_debug_stamps(unsigned long long out[16], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

hipError_t vr_launch_raycast(const RaycastLaunch &a, hipStream_t stream)
{
    switch (a.format) {
    case VRHIP_UCHAR: return launch_typed<uint8_t>(a, stream);
    case VRHIP_USHORT: return launch_typed<uint16_t>(a, stream);
    case VRHIP_FLOAT: return launch_typed<float>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t vr_launch_skipmap(const BrickView &bricks, int format, float inv_max, const TfView &tf,
                             uint32_t *bits, uint32_t n_words, hipStream_t stream)
{
    const size_t n = (size_t)bricks.bw * bricks.bh * bricks.bd;
    dim3 grid((unsigned)((n + kBlockDim - 1) / kBlockDim)), block(kBlockDim);
    switch (format) {
    case VRHIP_UCHAR:
        hipLaunchKernelGGL(vr_skipmap_kernel<uint8_t>, grid, block, 0, stream, bricks, inv_max, tf,
                           bits, n_words);
        break;
    case VRHIP_USHORT:
        hipLaunchKernelGGL(vr_skipmap_kernel<uint16_t>, grid, block, 0, stream, bricks, inv_max, tf,
                           bits, n_words);
        break;
    case VRHIP_FLOAT:
        hipLaunchKernelGGL(vr_skipmap_kernel<float>, grid, block, 0, stream, bricks, inv_max, tf,
                           bits, n_words);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
