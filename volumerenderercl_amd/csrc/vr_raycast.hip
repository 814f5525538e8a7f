// vr_raycast.hip -- the per-pixel front-to-back ray march for gfx950 (CDNA4).
//
// Replaces the reference's OpenCL kernel `volumeRender`
// (/root/reference/src/kernel/volumeraycast.cl:589-926).  CDNA has no image/sampler
// hardware (__HIP_NO_IMAGE_SUPPORT), so every read_imagef of the reference is restated
// as explicit address arithmetic + loads following the OpenCL 1.2 image rules
// (SURVEY.md App. B).  Work decomposition: one lane per pixel, one wave64 per 8x8-pixel
// patch (the reference's work-group), four waves (16x16 pixels) per workgroup sharing
// the transfer function in LDS.
#include "vr_device_math.h"
#include "vr_internal.h"

namespace {

constexpr int kBlockDim = 256;   // 4 waves, 16x16 pixels
constexpr int kBlockPix = 16;

// ------------------------------------------------------------------ volume reads

template <typename VT, int INSTR>
struct Vol {
    const VT *p;
    int w1, h1, d1;   // res - 1
    float fw, fh, fd;
    float inv_max;
    unsigned long long row, slice;
    int mbx, mby;
    uint32_t *touched;

    VR_DEV void touch(int x, int y, int z) const
    {
        if (INSTR == 2) {
            unsigned long long b = ((unsigned long long)(z >> 2) * (unsigned long long)mby +
                                    (unsigned long long)(y >> 2)) * (unsigned long long)mbx +
                                   (unsigned long long)(x >> 2);
            uint32_t bit = 1u << (uint32_t)(b & 31);
            uint32_t *wp = touched + (b >> 5);
            if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit))
                atomicOr(wp, bit);
        }
    }
    VR_DEV float raw(int x, int y, unsigned long long zoff, int z) const
    {
        touch(x, y, z);
        return (float)p[zoff + (unsigned long long)y * row + (unsigned long long)x];
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
        unsigned long long zo0 = (unsigned long long)z0 * slice;
        unsigned long long zo1 = (unsigned long long)z1 * slice;
        float v000 = raw(x0, y0, zo0, z0), v100 = raw(x1, y0, zo0, z0);
        float v010 = raw(x0, y1, zo0, z0), v110 = raw(x1, y1, zo0, z0);
        float v001 = raw(x0, y0, zo1, z1), v101 = raw(x1, y0, zo1, z1);
        float v011 = raw(x0, y1, zo1, z1), v111 = raw(x1, y1, zo1, z1);
        float c00 = lerpf(v000, v100, a);
        float c10 = lerpf(v010, v110, a);
        float c01 = lerpf(v001, v101, a);
        float c11 = lerpf(v011, v111, a);
        float c0 = lerpf(c00, c10, b);
        float c1 = lerpf(c01, c11, b);
        return lerpf(c0, c1, c) * inv_max;
    }

    // read_imagef(vol, nearestSmp, pos).x -- normalised, CLAMP (border 0), NEAREST
    VR_DEV float nearest(float px, float py, float pz) const
    {
        float fx = floorf(px * fw), fy = floorf(py * fh), fz = floorf(pz * fd);
        if (!(fx >= 0.0f && fx <= (float)w1 && fy >= 0.0f && fy <= (float)h1 && fz >= 0.0f &&
              fz <= (float)d1))
            return 0.0f;
        int z = (int)fz;
        return raw((int)fx, (int)fy, (unsigned long long)z * slice, z) * inv_max;
    }
};

// read_imagef(tffData, linearSmp, x) on the LDS copy of the float4 table
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

template <typename VT>
VR_DEV void brick_minmax(const BrickView &b, float inv_max, int cx, int cy, int cz, float *mn,
                         float *mx)
{
    if (cx < 0 || cy < 0 || cz < 0 || cx >= b.bw || cy >= b.bh || cz >= b.bd) {
        *mn = 0.0f;
        *mx = 0.0f;
        return;
    }
    size_t i = 2 * (((size_t)cz * (size_t)b.bh + (size_t)cy) * (size_t)b.bw + (size_t)cx);
    const VT *p = (const VT *)b.data;
    *mn = (float)p[i] * inv_max;
    *mx = (float)p[i + 1] * inv_max;
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

// pixel owned by this lane + where it lands in fb / out
struct PixelMap {
    uint32_t gx, gy;
    size_t out_index;
    bool inside;
};

VR_DEV PixelMap map_pixel(const FrameView &fr)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t lx = (lane & 7u) + 8u * (wave & 1u);
    const uint32_t ly = (lane >> 3) + 8u * (wave >> 1);
    PixelMap m;
    if (fr.tile_ids) {
        uint32_t b = blockIdx.x;
        uint32_t ti = b / fr.bpt, sub = b - ti * fr.bpt;
        uint32_t tile = fr.tile_ids[ti];
        uint32_t tx = tile % fr.tiles_x, ty = tile / fr.tiles_x;
        uint32_t px = (sub % fr.bpt_x) * kBlockPix + lx, py = (sub / fr.bpt_x) * kBlockPix + ly;
        m.gx = tx * fr.tile_w + px;
        m.gy = ty * fr.tile_h + py;
        m.out_index = ((size_t)ti * fr.tile_h + py) * fr.tile_w + px;
    } else {
        uint32_t b = blockIdx.x;
        uint32_t by = b / fr.blocks_x, bx = b - by * fr.blocks_x;
        m.gx = bx * kBlockPix + lx;
        m.gy = by * kBlockPix + ly;
        m.out_index = (size_t)m.gy * fr.W + m.gx;
    }
    m.inside = m.gx < fr.W && m.gy < fr.H;
    return m;
}

VR_DEV unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ------------------------------------------------------------------ ray cast

template <typename VT, bool ESS, int INSTR>
__global__ __launch_bounds__(kBlockDim) void vr_raycast_kernel(
    VolView vv, BrickView bricks, TfView tf, FrameView fr, vrhip_camera_params cam,
    vrhip_rendering_params rp, vrhip_raycast_params rc, DevStats *stats, uint32_t *touched)
{
    extern __shared__ float4 s_tff[];
    for (uint32_t i = threadIdx.x; i < tf.tff_n; i += kBlockDim) s_tff[i] = tf.tff[i];
    __syncthreads();

    const PixelMap pm = map_pixel(fr);
    unsigned long long c_taken = 0, c_nominal = 0, c_shaded = 0, c_bricks = 0, c_skipped = 0,
                       c_hit = 0;

    if (pm.inside) {
        Vol<VT, INSTR> vol;
        vol.p = (const VT *)vv.data;
        vol.w1 = vv.w - 1; vol.h1 = vv.h - 1; vol.d1 = vv.d - 1;
        vol.fw = vv.fw; vol.fh = vv.fh; vol.fd = vv.fd;
        vol.inv_max = vv.inv_max;
        vol.row = vv.row; vol.slice = vv.slice;
        vol.mbx = vv.mbx; vol.mby = vv.mby;
        vol.touched = touched;
        const int tffn = (int)tf.tff_n;

        const Ray ray = make_ray(pm.gx, pm.gy, fr, cam, rp);
        float result[4] = {ray.env[0], ray.env[1], ray.env[2], ray.env[3]};
        float tnear = ray.tnear;
        const float tfar = ray.tfar;
        const float sampleDist = tfar - tnear;
        if (ray.hit && sampleDist > 0.f) {
            c_hit = 1;
            const f3 camPos = ray.cam, rayDir = ray.dir;
            // volumeraycast.cl:709-733
            f3 resf = mk3(vol.fw, vol.fh, vol.fd);
            float stepSize = vmin(sampleDist,
                                  sampleDist / (rc.samplingRate *
                                                len3(mul3(scale3(rayDir, sampleDist), resf))));
            float samples = ceilf(sampleDist / stepSize);
            stepSize = sampleDist / samples;
            c_nominal = (unsigned long long)samples;

            tnear = vmax(0.f, tnear);
            float alpha = 0.f;
            float t = tnear;
            f3 voxLen = mk3(1.f / vol.fw, 1.f / vol.fh, 1.f / vol.fd);
            const float refInterval = 1.f / rc.samplingRate;
            float t_exit = tfar;
            const float offset = (len3(voxLen) * ray.rnd) * 2.0f;

            // per-ray invariants of illumination()/specularBlinnPhong() (:280-303):
            // l = fast_normalize(-rayDir), h = normalize(-rayDir + l)
            const f3 toLight = neg3(rayDir);
            const f3 lgt = normalize3(toLight);
            f3 hv = add3(toLight, lgt);
            const bool hvalid = !(dot3(hv, hv) < 1.e-6f);
            hv = normalize3(hv);
            const f3 goff = voxLen;   // gradientCentralDiff: offset = 1/volRes (:162)

            // 3-D DDA set-up (:737-760)
            int stepv[3] = {0, 0, 0}, cell[3] = {0, 0, 0}, exitc[3] = {0, 0, 0};
            float tv[3] = {0, 0, 0}, deltaT[3] = {0, 0, 0}, brickDia = 0.f;
            if (ESS) {
                const int bres[3] = {bricks.bw, bricks.bh, bricks.bd};
                float brickLen[3];
                const float dirv[3] = {rayDir.x, rayDir.y, rayDir.z};
                const float camv[3] = {camPos.x, camPos.y, camPos.z};
                for (int i = 0; i < 3; ++i) {
                    brickLen[i] = 1.f / rc.brickRes[i];
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
                brickDia = sqrtf(((brickLen[0] * brickLen[0]) + (brickLen[1] * brickLen[1])) +
                                 (brickLen[2] * brickLen[2])) * 2.f;
            }

            bool first = true;
            while (ESS ? (t < tfar) : first) {
                first = false;
                if (ESS) {
                    float mn, mx;
                    brick_minmax<VT>(bricks, vol.inv_max, cell[0], cell[1], cell[2], &mn, &mx);
                    if (INSTR) c_bricks++;
                    float inc0 = (tv[0] <= tv[1]) && (tv[0] <= tv[2]) ? 1.f : 0.f;
                    float inc1 = (tv[1] <= tv[0]) && (tv[1] <= tv[2]) ? 1.f : 0.f;
                    float inc2 = (tv[2] <= tv[0]) && (tv[2] <= tv[1]) ? 1.f : 0.f;
                    cell[0] += (int)inc0 * stepv[0];
                    cell[1] += (int)inc1 * stepv[1];
                    cell[2] += (int)inc2 * stepv[2];
                    t_exit = ((tv[0] * inc0) + (tv[1] * inc1)) + (tv[2] * inc2);
                    t_exit = vclamp(t_exit, t + stepSize, t + brickDia);
                    tv[0] += inc0 * deltaT[0];
                    tv[1] += inc1 * deltaT[1];
                    tv[2] += inc2 * deltaT[2];
                    float alphaMax = tff_linear_alpha(s_tff, tffn, mx);
                    if (alphaMax < 1e-6f) {
                        uint32_t pmin = prefix_nearest(tf.prefix, tf.prefix_n, mn);
                        uint32_t pmax = prefix_nearest(tf.prefix, tf.prefix_n, mx);
                        if (pmin == pmax) {
                            if (INSTR) c_skipped++;
                            t = t_exit;
                            continue;
                        }
                    }
                }
                // inner sample loop (:790-880)
                while (t < t_exit) {
                    if (INSTR) c_taken++;
                    f3 pos = add3(camPos, scale3(rayDir, t - offset));
                    pos = mk3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f);
                    float density = rp.useLinear ? vol.linear(pos.x, pos.y, pos.z)
                                                 : vol.nearest(pos.x, pos.y, pos.z);
                    float4 tfc = tff_linear(s_tff, tffn, density);
                    f3 grad = mk3(0.f, 0.f, 0.f);
                    const bool lit = tfc.w > 0.1f;
                    if (lit && (rp.illumType == 1 || (rc.contours && !rp.illumType))) {
                        // -gradientCentralDiff (:159-178, :814)
                        f3 s1, s2;
                        s1.x = vol.linear(pos.x + (-goff.x), pos.y + 0.0f, pos.z + 0.0f);
                        s1.y = vol.linear(pos.x + 0.0f, pos.y + (-goff.y), pos.z + 0.0f);
                        s1.z = vol.linear(pos.x + 0.0f, pos.y + 0.0f, pos.z + (-goff.z));
                        s2.x = vol.linear(pos.x + goff.x, pos.y + 0.0f, pos.z + 0.0f);
                        s2.y = vol.linear(pos.x + 0.0f, pos.y + goff.y, pos.z + 0.0f);
                        s2.z = vol.linear(pos.x + 0.0f, pos.y + 0.0f, pos.z + goff.z);
                        f3 g = sub3(s2, s1);
                        f3 n = normalize3(g);
                        if (dot3(g, g) == 0.0f) n = mk3(0.57735f, 0.57735f, 0.57735f);
                        grad = neg3(n);
                    }
                    if (lit && rp.illumType == 1) {
                        if (INSTR) c_shaded++;
                        // illumination (:294-303)
                        float ndl = vmax(0.f, dot3(grad, lgt));
                        float sp = hvalid ? vr_powr(vmax(dot3(grad, hv), 0.f), 40.f) : 0.0f;
                        sp = sp * 0.15f;
                        tfc.x = ((tfc.x * 0.15f) + ((tfc.x * ndl) * 0.7f)) + sp;
                        tfc.y = ((tfc.y * 0.15f) + ((tfc.y * ndl) * 0.7f)) + sp;
                        tfc.z = ((tfc.z * 0.15f) + ((tfc.z * ndl) * 0.7f)) + sp;
                    }
                    if (lit && rc.contours) {
                        float e = fabsf(dot3(rayDir, grad));
                        tfc.x *= e; tfc.y *= e; tfc.z *= e;
                    }
                    tfc.x = ray.env[0] - tfc.x;
                    tfc.y = ray.env[1] - tfc.y;
                    tfc.z = ray.env[2] - tfc.z;
                    if (rc.aerial) {
                        float depthCue = 1.f - (t - tnear) / sampleDist;
                        tfc.w *= depthCue;
                    }
                    float opacity = 1.f - vr_powr(1.f - tfc.w, refInterval);
                    float oma = 1.f - alpha;
                    result[0] = result[0] - (tfc.x * opacity) * oma;
                    result[1] = result[1] - (tfc.y * opacity) * oma;
                    result[2] = result[2] - (tfc.z * opacity) * oma;
                    alpha = alpha + opacity * oma;
                    if (t >= tfar) break;
                    if (alpha >= 0.98f) break;   // (double)alpha > 0.98, ERT_THRESHOLD (:28,:869)
                    t += stepSize;
                }
                if (!ESS) break;
                if (t >= tfar || alpha >= 0.98f) break;
                if (cell[0] == exitc[0] || cell[1] == exitc[1] || cell[2] == exitc[2]) break;
                t = t_exit;
            }
            result[3] = alpha;
            // running mean over iterations (:898-909), fp32 accumulate buffer
            if (rp.iteration != 0) {
                float4 prev = fr.fb[(size_t)pm.gy * fr.W + pm.gx];
                float it1 = (float)(rp.iteration + 1u);
                result[0] = prev.x + (result[0] - prev.x) / it1;
                result[1] = prev.y + (result[1] - prev.y) / it1;
                result[2] = prev.z + (result[2] - prev.z) / it1;
            }
        }
        float4 o = make_float4(result[0], result[1], result[2], result[3]);
        fr.fb[(size_t)pm.gy * fr.W + pm.gx] = o;
        if (fr.out) fr.out[pm.out_index] = o;
    }

    if (INSTR) {
        unsigned long long c[6] = {c_taken, c_nominal, c_shaded, c_bricks, c_skipped, c_hit};
        for (int i = 0; i < 6; ++i) {
            unsigned long long s = wave_sum(c[i]);
            if ((threadIdx.x & 63) == 0 && s) atomicAdd(&stats->v[i], s);
        }
    }
}

template <typename VT>
hipError_t launch_typed(const RaycastLaunch &a, hipStream_t stream)
{
    dim3 grid(a.n_blocks), block(kBlockDim);
    size_t lds = (size_t)a.tf.tff_n * sizeof(float4);
#define VR_LAUNCH(ESS, INSTR)                                                                 \
    hipLaunchKernelGGL((vr_raycast_kernel<VT, ESS, INSTR>), grid, block, lds, stream, a.vol,  \
                       a.bricks, a.tf, a.frame, a.cam, a.render, a.raycast, a.stats, a.touched)
    if (a.use_ess) {
        if (a.instr == 0) VR_LAUNCH(true, 0);
        else if (a.instr == 1) VR_LAUNCH(true, 1);
        else VR_LAUNCH(true, 2);
    } else {
        if (a.instr == 0) VR_LAUNCH(false, 0);
        else if (a.instr == 1) VR_LAUNCH(false, 1);
        else VR_LAUNCH(false, 2);
    }
#undef VR_LAUNCH
    return hipGetLastError();
}

} // namespace

hipError_t vr_launch_raycast(const RaycastLaunch &a, hipStream_t stream)
{
    switch (a.format) {
    case VRHIP_UCHAR: return launch_typed<uint8_t>(a, stream);
    case VRHIP_USHORT: return launch_typed<uint16_t>(a, stream);
    case VRHIP_FLOAT: return launch_typed<float>(a, stream);
    default: return hipErrorInvalidValue;
    }
}
