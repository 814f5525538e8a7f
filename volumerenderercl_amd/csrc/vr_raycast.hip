// vr_raycast.hip -- the per-pixel front-to-back ray march for gfx950 (CDNA4).
//
// Replaces the reference's OpenCL kernel `volumeRender`
// (/root/reference/src/kernel/volumeraycast.cl:589-926).  CDNA has no image/sampler
// hardware (__HIP_NO_IMAGE_SUPPORT), so every read_imagef of the reference is restated
// as explicit address arithmetic + loads following the OpenCL 1.2 image rules
// (vr_sampling.h).
//
// Execution design, four launches per frame on one stream:
//  * PRE-PASS (vr_dda_prepass_kernel, with ESS): one wave per 8x8-pixel patch (the reference's
//    work-group) at high occupancy: ray set-up and the reference's DDA up to the first brick the
//    ESS bitmap does not skip.  Rays that never reach one get their background pixel here; the
//    others go to a ray list with the DDA state they have reached.
//  * PHASE 1 (vr_raycast_rays_kernel on that list; vr_raycast_kernel on 8x8 patches for the
//    instrumented / XS variants and without ESS): one lane per ray in PERSISTENT waves.  The
//    transfer function (float4 table) and the ESS skip bitmap (1 bit per brick, precomputed from
//    bricks + TF + prefix sum) live in LDS: a DDA step touches no global memory.  The reference's
//    nested loops (DDA over bricks / samples inside a brick) are flattened into a per-lane state
//    machine driven by wave ballots; each sample round evaluates up to kBatch consecutive samples
//    of every ray as independent straight-line code.
//  * The frame time of a ray caster on a machine this wide is set by its LONGEST rays: their
//    samples form a serial chain.  So phase 1 marches a ray for at most `round_budget` sample
//    rounds; rays still alive are SUSPENDED (13 words of state), counting-sorted by the rounds
//    their pixel needed in the previous frame (longest first), and
//  * PHASE 2 (vr_raycast_split_kernel) resumes them with kSplit = 4 lanes per ray: each lane
//    evaluates 4 of the ray's next 16 consecutive samples, then the 16 contributions are
//    composited in ray order (in-quad DPP broadcasts).  The chain of a long ray shrinks 4x; the
//    16 ray slots of a wave draw their rays one by one from the sorted list.
//  Several independent frames (jitter seeds) can share one set of these launches: the work items
//  carry a frame index (vrhip_render_batch).
//  The per-ray sequence of t values and of fp32 operations is exactly the reference's in every
//  kernel, so the image is bit-identical whatever the schedule (budget, refill, batch, lists).
#include <algorithm>

#include "vr_sampling.h"

// A/B builds (tools/mkvariant.sh NAME -DVR_EXPERIMENTS [-DVR_LEAP_STEPPING]) add the kernels that lost their
// A/B in round 2 -- vr_march_kernel (decoupled march), vr_raycast_staged_kernel (LDS brick staging), leap
// stepping inside the two-phase kernels -- from vr_experiments_*.inc; the product library has none of them.
#if defined(VR_EXPERIMENTS) && defined(VR_LEAP_STEPPING)
#define VR_LEAP 1
#endif
#ifdef VR_EXPERIMENTS
#include "vr_leap.h"
#endif

namespace {

constexpr int kSplit = 4;            // lanes per ray in phase 2 (x kBatch samples per lane)

#ifdef VR_MARCH_STATS   // diagnostic build: what the waves of the marching kernels spend their rounds on
__device__ unsigned long long g_march_stats[32];   // [0, 16) march kernel, [16, 32) phase 1 on the ray list
#define VR_MS(i, v) ms_acc[i] += (unsigned long long)(v)
#else
#define VR_MS(i, v)
#endif
#ifdef VR_ISA_MARKS   // diagnostic: comments in the -S output that delimit the stages (tools/isa_marks.py)
#define VR_MARK(x) asm volatile("; VRMARK " x ::: "memory")
#else
#define VR_MARK(x)
#endif

// Occupancy experiments: -DVR_WAVES_PER_EU=N asks the compiler to fit N waves per SIMD
#ifdef VR_WAVES_PER_EU
#define VR_OCC __attribute__((amdgpu_waves_per_eu(VR_WAVES_PER_EU, VR_WAVES_PER_EU)))
#else
#define VR_OCC
#endif

struct RayCtx {   // per-ray invariants, recomputable from the pixel
    f3 cam, dir;
    float env0, env1, env2, env3;
    float tnear;      // clamped to >= 0 (:719)
    float tfar, sampleDist, stepSize, offset;
    f3 lgt, hv;       // illumination invariants (:280-303)
    bool hvalid;
    int stepv0, stepv1, stepv2, exit0, exit1, exit2;
    float dT0, dT1, dT2;
    bool valid;       // hits the clip box with sampleDist > 0
    bool miss;        // inside the image, misses the clip box (image-order ESS bookkeeping)
    float nominal;    // ceil(sampleDist / stepSize)
};

struct RayDyn {   // marching state
    int state;
    float t, t_exit, alpha, r0, r1, r2;
    int c0, c1, c2;
    float tv0, tv1, tv2;
    uint32_t cidx, skw;   // linear index of the current brick cell and its bitmap word
    float t_last;         // XS variants: ray parameter of the last sample taken, < 0 = none (showEss)
    float t_ert;          // ray parameter of the sample that triggered early ray termination
    bool ert;             // (ambient occlusion is applied there, :870-876)
#ifdef VR_RAYLEN          // diagnostic build: samples taken by the ray, written to the alpha channel
    uint32_t nsmp;
#endif
};
#ifdef VR_RAYLEN
#define VR_RAYLEN_INC(d) ((d).nsmp++)
#else
#define VR_RAYLEN_INC(d)
#endif

struct Grid {     // wave-uniform brick-grid constants
    int bw, bh, bd;
    float bl0, bl1, bl2, brickDia;
    uint32_t oob_word;
};

// volumeraycast.cl:605-683 for one pixel: ray, background, clip.  The first half of setup_ray: all the
// pre-pass's patch culling needs (and all a ray that is never marched needs: write_pixel reads the
// background and the clip result).  SHADE: the illumination invariants (:280-303).
template <bool SHADE>
VR_DEV void setup_ray_head(uint32_t gx, uint32_t gy, bool inside, const FrameView &fr,
                           const vrhip_camera_params &cam, const vrhip_rendering_params &rp, RayCtx &c,
                           RayDyn &d, uint32_t seed, float &rnd)
{
    const Ray ray = make_ray(gx, gy, fr, cam, rp, seed);
    rnd = ray.rnd;
    c.cam = ray.cam;
    c.dir = ray.dir;
    c.env0 = ray.env[0]; c.env1 = ray.env[1]; c.env2 = ray.env[2]; c.env3 = ray.env[3];
    c.tfar = ray.tfar;
    c.sampleDist = ray.tfar - ray.tnear;
    c.valid = inside && ray.hit && c.sampleDist > 0.f;
    c.miss = inside && !ray.hit;
    c.tnear = ray.tnear;
    c.stepSize = 0.f; c.offset = 0.f; c.nominal = 0.f;
    c.stepv0 = c.stepv1 = c.stepv2 = 0;
    c.exit0 = c.exit1 = c.exit2 = 0;
    c.dT0 = c.dT1 = c.dT2 = 0.f;
    if (SHADE) {
        const f3 toLight = neg3(ray.dir);
        c.lgt = normalize3(toLight);
        f3 hv = add3(toLight, c.lgt);
        c.hvalid = !(dot3(hv, hv) < 1.e-6f);
        c.hv = normalize3(hv);
    } else {
        c.lgt = c.hv = mk3(0.f, 0.f, 0.f);
        c.hvalid = false;
    }

    d.state = S_DONE;
    d.t = 0.f; d.t_exit = ray.tfar; d.alpha = 0.f;
    d.r0 = c.env0; d.r1 = c.env1; d.r2 = c.env2;
    d.c0 = d.c1 = d.c2 = 0;
    d.tv0 = d.tv1 = d.tv2 = 0.f;
    d.cidx = 0; d.skw = 0;
    d.t_ert = 0.f; d.ert = false;
    d.t_last = -1.f;
#ifdef VR_RAYLEN
    d.nsmp = 0;
#endif
    if (c.valid) c.tnear = vmax(0.f, ray.tnear);   // :719 (the unclamped value is not needed again)
}

// volumeraycast.cl:709-760: step size, jitter offset, DDA set-up -- for rays that will be marched.
template <bool ESS>
VR_DEV void setup_ray_tail(const vrhip_raycast_params &rcp, f3 resf, f3 voxLen, const Grid &g, RayCtx &c,
                           RayDyn &d, float rnd)
{
    if (c.valid) {
        // volumeraycast.cl:709-733
        float stepSize = vmin(c.sampleDist,
                              c.sampleDist / (rcp.samplingRate *
                                              len3(mul3(scale3(c.dir, c.sampleDist), resf))));
        c.nominal = ceilf(c.sampleDist / stepSize);
        c.stepSize = c.sampleDist / c.nominal;
        d.t = c.tnear;
        c.offset = (len3(voxLen) * rnd) * 2.0f;
        d.state = ESS ? S_BRICK : S_SAMPLE;
        if (ESS) {   // 3-D DDA set-up (:737-760)
            const int bres[3] = {g.bw, g.bh, g.bd};
            const float bl[3] = {g.bl0, g.bl1, g.bl2};
            const float dirv[3] = {c.dir.x, c.dir.y, c.dir.z};
            const float camv[3] = {c.cam.x, c.cam.y, c.cam.z};
            int stepv[3], cell[3], exitc[3];
            float tv[3], dT[3];
            for (int i = 0; i < 3; ++i) {
                float invRay = 1.f / dirv[i];
                stepv[i] = dirv[i] > 0.f ? 1 : (dirv[i] < 0.f ? -1 : 0);
                dT[i] = (float)stepv[i] * ((bl[i] * 2.f) * invRay);
                float roc = (camv[i] + dirv[i] * c.tnear) - (-1.f);
                cell[i] = iclamp((int)floorf(roc / (2.f * bl[i])), 0, bres[i] - 1);
                int cadj = cell[i] - (dirv[i] >= 0.f ? -1 : 0);
                tv[i] = c.tnear + ((float)cadj * (2.f * bl[i]) - roc) * invRay;
                exitc[i] = stepv[i] * bres[i];
                if (exitc[i] < 0) exitc[i] = -1;
            }
            c.stepv0 = stepv[0]; c.stepv1 = stepv[1]; c.stepv2 = stepv[2];
            c.exit0 = exitc[0]; c.exit1 = exitc[1]; c.exit2 = exitc[2];
            c.dT0 = dT[0]; c.dT1 = dT[1]; c.dT2 = dT[2];
            d.c0 = cell[0]; d.c1 = cell[1]; d.c2 = cell[2];
            d.tv0 = tv[0]; d.tv1 = tv[1]; d.tv2 = tv[2];
        }
    }
}

// volumeraycast.cl:605-760 for one pixel: ray, background, clip, step size, DDA set-up.
template <bool ESS>
VR_DEV void setup_ray(uint32_t gx, uint32_t gy, bool inside, const FrameView &fr,
                      const vrhip_camera_params &cam, const vrhip_rendering_params &rp,
                      const vrhip_raycast_params &rcp, f3 resf, f3 voxLen, const Grid &g, RayCtx &c,
                      RayDyn &d, uint32_t seed)
{
    float rnd;
    setup_ray_head<true>(gx, gy, inside, fr, cam, rp, c, d, seed, rnd);
    setup_ray_tail<ESS>(rcp, resf, voxLen, g, c, d, rnd);
}

// bitmap word of the cell the ray is in (out-of-range cells read the trailing word, which
// holds the (0,0) decision in every bit)
VR_DEV void fetch_skip_word(const uint32_t *sb, const Grid &g, RayDyn &d)
{
    const bool oob = (uint32_t)d.c0 >= (uint32_t)g.bw || (uint32_t)d.c1 >= (uint32_t)g.bh ||
                     (uint32_t)d.c2 >= (uint32_t)g.bd;
    d.cidx = __umul24(__umul24((uint32_t)d.c2, (uint32_t)g.bh) + (uint32_t)d.c1, (uint32_t)g.bw) +
             (uint32_t)d.c0;
    d.skw = sb[oob ? g.oob_word : (d.cidx >> 5)];
}

// One DDA step (:763-787) as branch-free predicated code (a lone wave pays ~60 cycles per
// ballot + scalar branch; tools/micro_issue.hip).  Every lane computes the step; `go` (lane is
// in S_BRICK and passes the outer loop condition t < tfar) gates what is committed.  The
// decision for the current cell comes from the bitmap word fetched one step ahead.
template <int INSTR>
VR_DEV void dda_step(const uint32_t *sb, const Grid &g, const RayCtx &c, RayDyn &d,
                     unsigned long long &c_bricks, unsigned long long &c_skipped)
{
    const bool inB = d.state == S_BRICK;
    const bool go = inB && (d.t < c.tfar);
    const bool skp = (d.skw >> (d.cidx & 31u)) & 1u;
    const bool m0 = (d.tv0 <= d.tv1) && (d.tv0 <= d.tv2);
    const bool m1 = (d.tv1 <= d.tv0) && (d.tv1 <= d.tv2);
    const bool m2 = (d.tv2 <= d.tv0) && (d.tv2 <= d.tv1);
    const float inc0 = m0 ? 1.f : 0.f, inc1 = m1 ? 1.f : 0.f, inc2 = m2 ? 1.f : 0.f;
    float te = ((d.tv0 * inc0) + (d.tv1 * inc1)) + (d.tv2 * inc2);
    te = vclamp(te, d.t + c.stepSize, d.t + g.brickDia);
    d.c0 += (go && m0) ? c.stepv0 : 0;
    d.c1 += (go && m1) ? c.stepv1 : 0;
    d.c2 += (go && m2) ? c.stepv2 : 0;
    d.tv0 = go ? d.tv0 + inc0 * c.dT0 : d.tv0;
    d.tv1 = go ? d.tv1 + inc1 * c.dT1 : d.tv1;
    d.tv2 = go ? d.tv2 + inc2 * c.dT2 : d.tv2;
    d.t_exit = go ? te : d.t_exit;
    fetch_skip_word(sb, g, d);
    if (INSTR) { c_bricks += go ? 1 : 0; c_skipped += (go && skp) ? 1 : 0; }
    d.t = (go && skp) ? te : d.t;   // :784-785 `continue`
    d.state = inB ? (go ? (skp ? S_BRICK : S_SAMPLE) : S_DONE) : d.state;
}

// The inner loop was left by its condition (:790): the checks after it (:882-884).
template <bool ESS>
VR_DEV void after_segment(const RayCtx &c, RayDyn &d)
{
    if (d.state == S_SAMPLE && !(d.t < d.t_exit)) {
        if (!ESS) d.state = S_DONE;
        else if (d.t >= c.tfar || d.alpha >= 0.98f) d.state = S_DONE;                       // :882
        else if (d.c0 == c.exit0 || d.c1 == c.exit1 || d.c2 == c.exit2) d.state = S_DONE;  // :883
        else { d.t = d.t_exit; d.state = S_BRICK; }                                         // :884
    }
}

// LDS staging of the gathered sample evaluation (eval_batch): one slot per sample of a wave's
// round (64 lanes x kBatch samples), for each of the 4 waves of a workgroup
constexpr int kSlotFloats = 5;   // in: pos.xyz, opacity, owner|flags   out: ndl, spec, contour, op
constexpr int kStageFloatsPerWave = 64 * kBatch * kSlotFloats;
constexpr int kStageF4 = (kBlockDim / 64) * kStageFloatsPerWave / 4;   // float4 units, whole workgroup

// Up to kBatch consecutive samples of one ray (inner loop, :790-864): for each, the colour
// already multiplied by the sample's opacity and the opacity.  Neither depends on the running
// alpha, so the batch is independent straight-line code (the loads of all its fetches are in
// flight together) and only the cheap front-to-back compositing is sequential.  Samples past
// ERT / t_exit are speculative: fetched from clamped (always valid) addresses, never composited.
//
// The kernel is bound by VALU issue, and the expensive part of a sample -- opacity correction
// (powr), and for samples above the shading threshold the gradient (32 voxel loads), the
// Blinn-Phong terms and a second powr -- only matters for samples whose opacity is not 0:
// often a few per cent of them, scattered over lanes and batch slots.  Under per-slot divergent
// branches that code would run up to kBatch times per round for a handful of lanes each.
// Instead the wave GATHERS its non-zero samples into LDS slots, evaluates the expensive scalars
// over the dense slot list (usually one pass over the active lanes; per-ray constants come from
// the owner lane by ds_bpermute) and hands four scalars per sample back.  Every sample sees the
// same fp32 operations as before, in another lane.
// FP: the kernel variant that reads the footprint volume (VolView::fp) instead of the plain
// layout -- default kernels only (no instrumentation, no XS extras); see launch_typed.
template <typename VT, int INSTR, bool XS, bool FP, typename V>
VR_DEV void eval_batch(const V &vol, const float4 *s_tff, int tffn, float *s_stage,
                       const RayCtx &c, const vrhip_rendering_params &rp,
                       const vrhip_raycast_params &rcp, float refInterval,
                       const float (&tk)[kBatch], const bool (&vk)[kBatch], float (&p0)[kBatch],
                       float (&p1)[kBatch], float (&p2)[kBatch], float (&op)[kBatch],
                       bool (&shaded)[kBatch])
{
    VR_MARK("E_pos");
    f3 pk[kBatch];
    float dens[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        f3 pos = add3(c.cam, scale3(c.dir, tk[k] - c.offset));
        pk[k] = mk3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f);
        dens[k] = 0.f;
    }
    VR_MARK("E_fetch");
    if (XS && rp.illumType == 4) {
        // handled below
    } else if (!XS || rp.useLinear) {   // (nearest filtering, contours and the depth cue: XS variants, see launch_typed)
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
            if (INSTR != 2 || vk[k]) dens[k] = vol.linear(pk[k].x, pk[k].y, pk[k].z);
    } else {
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
            if (INSTR != 2 || vk[k]) dens[k] = vol.nearest(pk[k].x, pk[k].y, pk[k].z);
    }
    if (XS && rp.illumType == 4) {
        // gradient magnitude through the transfer function (:796-799): no density fetch
#pragma unroll 1
        for (int k = 0; k < kBatch; ++k)
            dens[k] = (INSTR != 2 || vk[k]) ? vol.gradient_len(pk[k].x, pk[k].y, pk[k].z) : 0.f;
    }
    VR_MARK("E_tf");
    float4 tfc[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) tfc[k] = tff_linear(s_tff, tffn, dens[k]);

    // CL_RGBA / CL_RG volumes (:838-855): the voxel is the colour (RGBA) or (r, g) -> colour
    // (r, 0, 0) with opacity TF(|g|); neither is shaded.  dens[] holds channel 0 already.
    const bool multi = XS && vol.channels > 1 && rp.illumType != 4;
    if (multi) {
#pragma unroll 1
        for (int k = 0; k < kBatch; ++k) {
            if (INSTR == 2 && !vk[k]) continue;
            float ch[3] = {0.f, 0.f, 0.f};
            for (int j = 1; j < vol.channels; ++j) {
                const auto vc = vol.channel(j);
                ch[j - 1] = rp.useLinear ? vc.linear(pk[k].x, pk[k].y, pk[k].z)
                                         : vc.nearest(pk[k].x, pk[k].y, pk[k].z);
            }
            if (vol.channels == 4) tfc[k] = make_float4(dens[k], ch[0], ch[1], ch[2]);
            else tfc[k] = make_float4(dens[k], 0.f, 0.f, tff_linear(s_tff, tffn, fabsf(ch[0] / 1.f)).w);
        }
    }

    VR_MARK("E_slots");
    // ---- which samples need the expensive part, and their slots
    const bool shade_mode = XS ? (rp.illumType != 0 && rp.illumType != 4) : rp.illumType == 1;   // :809
    const bool want_grad = shade_mode || (XS && rcp.contours && !rp.illumType);
    const uint32_t lane = threadIdx.x & 63u;
    bool lit[kBatch], need[kBatch];
    uint32_t slot[kBatch];
    uint32_t n_slots = 0;
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        lit[k] = vk[k] && tfc[k].w > 0.1f && !(XS && rp.illumType == 4) && !multi;   // :809/:832, before the depth cue
        shaded[k] = lit[k] && shade_mode;
        if (XS && rcp.aerial) {                               // :858-862
            float depthCue = 1.f - (tk[k] - c.tnear) / c.sampleDist;
            tfc[k].w *= depthCue;
        }
        // opacity 0 gives op = 1 - powr(1, y) = 0 exactly: nothing of the sample survives
        need[k] = vk[k] && tfc[k].w != 0.f;
        const unsigned long long m = __ballot(need[k]);
        slot[k] = n_slots + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        n_slots += (uint32_t)__builtin_popcountll(m);
    }
    float ndl[kBatch], spc[kBatch], cnt[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) { ndl[k] = 0.f; spc[k] = 0.f; cnt[k] = 0.f; op[k] = 0.f; }

    VR_MARK("E_stage");
    if (n_slots) {   // wave-uniform
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (need[k]) {
                float *q = s_stage + kSlotFloats * slot[k];
                q[0] = pk[k].x; q[1] = pk[k].y; q[2] = pk[k].z;
                q[3] = tfc[k].w;
                q[4] = __uint_as_float(lane | ((lit[k] && want_grad) ? 64u : 0u));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // only the lanes inside this (divergent) call can work: slots go to them by rank
        const unsigned long long act = __ballot(true);
        const uint32_t n_act = (uint32_t)__builtin_popcountll(act);
        const uint32_t arank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
    VR_MARK("E_dense");
        for (uint32_t base = 0; base < n_slots; base += n_act) {
            const uint32_t sidx = base + arank;
            const bool mine = sidx < n_slots;
            float *q = s_stage + kSlotFloats * (mine ? sidx : 0u);
            const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
            const uint32_t tag = mine ? __float_as_uint(q[4]) : lane;
            const int owner = (int)(tag & 63u);
            const bool shade = mine && (tag & 64u);
            // the owner's per-ray constants (every lane takes part in the exchange)
            const f3 lgt = mk3(__shfl(c.lgt.x, owner, 64), __shfl(c.lgt.y, owner, 64),
                               __shfl(c.lgt.z, owner, 64));
            const f3 hv = mk3(__shfl(c.hv.x, owner, 64), __shfl(c.hv.y, owner, 64),
                              __shfl(c.hv.z, owner, 64));
            const int hvalid = __shfl(c.hvalid ? 1 : 0, owner, 64);
            f3 dirv = mk3(0.f, 0.f, 0.f);
            if (XS && rcp.contours)
                dirv = mk3(__shfl(c.dir.x, owner, 64), __shfl(c.dir.y, owner, 64),
                           __shfl(c.dir.z, owner, 64));
            float o_ndl = 0.f, o_sp = 0.f, o_cnt = 0.f, o_op = 0.f;
            if (__ballot(shade)) {
                if (shade) {
                    f3 g;
                    if (XS && rp.illumType == 2) {    // :816-818 central differences of TF opacities
                        const float4 gq = gradient_tff<VT, INSTR>(vol, s_tff, tffn, mk3(qx, qy, qz));
                        g = mk3(-gq.x, -gq.y, -gq.z);
                    } else if (XS && rp.illumType == 3) {   // :819-821 Sobel
                        g = vol.neg_sobel(qx, qy, qz);
                    } else {                          // 1, 5, and contours without illumination
                        g = vol.neg_gradient(qx, qy, qz);
                    }
                    // illumination (:294-303) with specularBlinnPhong (:280-291); cel shading
                    // (:306-319) only needs the diffuse term
                    o_ndl = vmax(0.f, dot3(g, lgt));
                    if (!(XS && rp.illumType == 5)) {
                        o_sp = hvalid ? vr_powr(vmax(dot3(g, hv), 0.f), 40.f) : 0.0f;
                        o_sp = o_sp * 0.15f;
                    }
                    o_cnt = fabsf(dot3(dirv, g));             // contours (:846-848)
                }
            }
            if (mine) {
                o_op = 1.f - vr_powr(1.f - qw, refInterval);  // opacity correction (:864)
                q[0] = o_ndl; q[1] = o_sp; q[2] = o_cnt; q[3] = o_op;
            }
        }
    VR_MARK("E_readback");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (need[k]) {
                const float *q = s_stage + kSlotFloats * slot[k];
                ndl[k] = q[0]; spc[k] = q[1]; cnt[k] = q[2]; op[k] = q[3];
            }
        }
    }

    VR_MARK("E_combine");
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        if (lit[k] && shade_mode && !(XS && rp.illumType == 5)) {
            tfc[k].x = ((tfc[k].x * 0.15f) + ((tfc[k].x * ndl[k]) * 0.7f)) + spc[k];
            tfc[k].y = ((tfc[k].y * 0.15f) + ((tfc[k].y * ndl[k]) * 0.7f)) + spc[k];
            tfc[k].z = ((tfc[k].z * 0.15f) + ((tfc[k].z * ndl[k]) * 0.7f)) + spc[k];
        }
        if (XS && lit[k] && rp.illumType == 5) {   // celShading (:306-319), intensity = ndl
            const float f = ndl[k] > 0.95f ? 1.0f : ndl[k] > 0.5f ? 0.6f : ndl[k] > 0.25f ? 0.4f : 0.2f;
            if (!(ndl[k] > 0.95f)) { tfc[k].x *= f; tfc[k].y *= f; tfc[k].z *= f; }
        }
        if (XS && lit[k] && rcp.contours) {
            tfc[k].x *= cnt[k]; tfc[k].y *= cnt[k]; tfc[k].z *= cnt[k];
        }
        tfc[k].x = c.env0 - tfc[k].x;
        tfc[k].y = c.env1 - tfc[k].y;
        tfc[k].z = c.env2 - tfc[k].z;
        p0[k] = tfc[k].x * op[k];
        p1[k] = tfc[k].y * op[k];
        p2[k] = tfc[k].z * op[k];
    }
}

// ---- eval_batch for the default kernels (no instrumentation, no XS extras), cut in three so that the dense
// pass over the gathered samples is run by ALL 64 lanes of the wave, not only by the lanes whose rays
// evaluate a batch this round.  A batch runs with ~30 of 64 lanes and gathers ~38 samples with a non-zero
// opacity on the headline: inside the divergent call that was two passes of the gradient / shading code
// more often than not; the wave's idle and empty-run-skipping lanes take the second half now.  The same
// operations per sample, in another lane (as before).
struct EvalFront {
    float4 tfc[kBatch];
    bool lit[kBatch], need[kBatch];
    uint32_t slot[kBatch];
};

// divergent part 1: density, transfer function, slots of the samples with an opacity, staged to LDS.
// Returns the number of slots (the same in every lane that calls).
template <typename VT, bool FP, typename V>
VR_DEV uint32_t eval_front(const V &vol, const float4 *s_tff, int tffn, float *s_stage, const RayCtx &c,
                           const vrhip_rendering_params &rp, const float (&tk)[kBatch], const bool (&vk)[kBatch],
                           bool ev, EvalFront &ef)
{
    // Called by the whole wave: a VALU instruction costs the same with 30 lanes as with 64, and straight
    // code spares the exec-mask bookkeeping of a divergent region around the batch.  Lanes whose ray does
    // not evaluate this round (ev false: every vk false) only skip the voxel loads.
    f3 pk[kBatch];
    float dens[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        f3 pos = add3(c.cam, scale3(c.dir, tk[k] - c.offset));
        pk[k] = mk3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f);
        dens[k] = 0.f;
    }
    if (ev) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) dens[k] = vol.linear(pk[k].x, pk[k].y, pk[k].z);
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) ef.tfc[k] = tff_linear(s_tff, tffn, dens[k]);
    const bool shade_mode = rp.illumType == 1;   // :809
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n_slots = 0;
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        ef.lit[k] = vk[k] && ef.tfc[k].w > 0.1f;   // :809/:832
        // opacity 0 gives op = 1 - powr(1, y) = 0 exactly: nothing of the sample survives
        ef.need[k] = vk[k] && ef.tfc[k].w != 0.f;
        const unsigned long long m = __ballot(ef.need[k]);
        ef.slot[k] = n_slots + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        n_slots += (uint32_t)__builtin_popcountll(m);
    }
    if (n_slots) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (ef.need[k]) {
                float *q = s_stage + kSlotFloats * ef.slot[k];
                q[0] = pk[k].x; q[1] = pk[k].y; q[2] = pk[k].z;
                q[3] = ef.tfc[k].w;
                q[4] = __uint_as_float(lane | ((ef.lit[k] && shade_mode) ? 64u : 0u));
            }
        }
    }
    return n_slots;
}

// wave-uniform part: every lane of the wave takes slots (n_slots > 0, the same in all 64 lanes)
template <typename VT, bool FP, typename V>
VR_DEV void eval_dense(const V &vol, float *s_stage, const RayCtx &c, float refInterval, uint32_t n_slots)
{
    const uint32_t lane = threadIdx.x & 63u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < n_slots; base += 64u) {
        const uint32_t sidx = base + lane;
        const bool mine = sidx < n_slots;
        float *q = s_stage + kSlotFloats * (mine ? sidx : 0u);
        const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
        const uint32_t tag = mine ? __float_as_uint(q[4]) : lane;
        const int owner = (int)(tag & 63u);
        const bool shade = mine && (tag & 64u);
        // the owner's per-ray constants (every lane takes part in the exchange)
        const f3 lgt = mk3(__shfl(c.lgt.x, owner, 64), __shfl(c.lgt.y, owner, 64), __shfl(c.lgt.z, owner, 64));
        const f3 hv = mk3(__shfl(c.hv.x, owner, 64), __shfl(c.hv.y, owner, 64), __shfl(c.hv.z, owner, 64));
        const int hvalid = __shfl(c.hvalid ? 1 : 0, owner, 64);
        float o_ndl = 0.f, o_sp = 0.f, o_op = 0.f;
        if (__ballot(shade)) {
            if (shade) {
                const f3 g = vol.neg_gradient(qx, qy, qz);
                // illumination (:294-303) with specularBlinnPhong (:280-291)
                o_ndl = vmax(0.f, dot3(g, lgt));
                o_sp = hvalid ? vr_powr(vmax(dot3(g, hv), 0.f), 40.f) : 0.0f;
                o_sp = o_sp * 0.15f;
            }
        }
        if (mine) {
            o_op = 1.f - vr_powr(1.f - qw, refInterval);  // opacity correction (:864)
            q[0] = o_ndl; q[1] = o_sp; q[2] = 0.f; q[3] = o_op;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// divergent part 2: the samples' scalars back from their slots, shading combined, colour x opacity
VR_DEV void eval_back(const float *s_stage, const RayCtx &c, const vrhip_rendering_params &rp, EvalFront &ef,
                      uint32_t n_slots, float (&p0)[kBatch], float (&p1)[kBatch], float (&p2)[kBatch],
                      float (&op)[kBatch])
{
    const bool shade_mode = rp.illumType == 1;
    float ndl[kBatch], spc[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) { ndl[k] = 0.f; spc[k] = 0.f; op[k] = 0.f; }
    if (n_slots) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (ef.need[k]) {
                const float *q = s_stage + kSlotFloats * ef.slot[k];
                ndl[k] = q[0]; spc[k] = q[1]; op[k] = q[3];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        float4 t = ef.tfc[k];
        if (ef.lit[k] && shade_mode) {
            t.x = ((t.x * 0.15f) + ((t.x * ndl[k]) * 0.7f)) + spc[k];
            t.y = ((t.y * 0.15f) + ((t.y * ndl[k]) * 0.7f)) + spc[k];
            t.z = ((t.z * 0.15f) + ((t.z * ndl[k]) * 0.7f)) + spc[k];
        }
        t.x = c.env0 - t.x;
        t.y = c.env1 - t.y;
        t.z = c.env2 - t.z;
        p0[k] = t.x * op[k];
        p1[k] = t.y * op[k];
        p2[k] = t.z * op[k];
    }
}

#ifndef VR_LOOK1
#define VR_LOOK1 24
#endif
#ifndef VR_LOOK_NUM
#define VR_LOOK_NUM 2   // the lookahead runs when at least 1 / VR_LOOK_NUM of the sampling lanes expect an empty sample
#endif
#ifndef VR_LOOK2
#define VR_LOOK2 8
#endif
// samples looked ahead for empty runs per lane and round (<= 32): phase 1 (one lane per ray) and
// phase 2 (four lanes per ray, each with its own window)
constexpr int kLook1 = VR_LOOK1, kLook2 = VR_LOOK2;

// Bit k set: sample k of the run t0, t0 + stepSize, ... lies in an EMPTY cell (CellView): its
// fetch can only map to opacity 0, so compositing it changes nothing (:864-879 with alpha == 0).
// The cell comes from a linearised voxel position u' = p * res (+ k * du'), in cells: the fetch's
// low-corner texel is x0 = floor(u' - 0.5), so x' = floor(u') is x0 or x0 + 1 -- also with the
// linearisation's error, which is far below half a texel -- and the voxels x0, x0 + 1 the fetch reads
// lie in [x' - 1, x' + 1], inside the extent [E c - 1, E c + E + 1] the cell of x' answers for (the
// halo is there for exactly this).  Three instructions per axis and sample, no voxel access.
// Positions outside the volume (samples before the entry face, speculative samples past the ray's
// end) clamp to the nearest border cell, like the fetch's clamp-to-edge addressing.
template <typename VT, int INSTR, int kLook, typename V>
VR_DEV uint32_t empty_mask(const CellView &cv, const V &vol, const RayCtx &c, float t0)
{
    const f3 p0 = add3(c.cam, scale3(c.dir, t0 - c.offset));
    const float inv_e = __uint_as_float((uint32_t)(127 - cv.eshift) << 23);   // 2^-eshift
    const float su = vol.fw * inv_e, sv = vol.fh * inv_e, ss = vol.fd * inv_e;
    const float u0 = (p0.x * 0.5f + 0.5f) * su;
    const float v0 = (p0.y * 0.5f + 0.5f) * sv;
    const float s0 = (p0.z * 0.5f + 0.5f) * ss;
    const float du = (c.dir.x * c.stepSize) * (0.5f * su);
    const float dv = (c.dir.y * c.stepSize) * (0.5f * sv);
    const float ds = (c.dir.z * c.stepSize) * (0.5f * ss);
    const float mx = (float)(cv.ecx - 1), my = (float)(cv.ecy - 1), mz = (float)(cv.ecz - 1);
    uint32_t w[kLook], sh[kLook];
#pragma unroll
    for (int k = 0; k < kLook; ++k) {
        const float fk = (float)k;
        // Signed clamp: real samples near tnear lie up to 2 |voxLen| BEFORE the entry face (t - offset, :733 /
        // :791) and are fetched clamp-to-edge, i.e. they read column 0 -- on an anisotropic grid that is several
        // cells below 0, and an unsigned clamp would send them to the far border's cell.  The clamp is taken in
        // the float domain, BEFORE the conversion (one v_med3_f32 instead of an integer max and min: the compiler
        // cannot prove 0 <= mx for a v_med3_i32): trunc(clamp(v, 0, mx)) == clamp(trunc(v), 0, mx) for every
        // finite v and integer mx >= 0 -- an index, not an fp32 result of the image.
        const uint32_t x = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(fk, du, u0), 0.f, mx);
        const uint32_t y = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(fk, dv, v0), 0.f, my);
        const uint32_t z = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(fk, ds, s0), 0.f, mz);
        const uint32_t idx = (z * (uint32_t)cv.ecy + y) * (uint32_t)cv.ecx + x;
        w[k] = cv.empty[idx >> 5];
        sh[k] = idx & 31u;
    }
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < kLook; ++k) m |= ((w[k] >> sh[k]) & 1u) << k;
    return m;
}

// Step over the leading empty samples of the run (the reference's own t sequence and loop
// exits, :790 and :868-879; nothing else of the loop body has an effect for them).  Returns
// true when all kLook1 samples were consumed and the run may continue.
//
// Branch-free (round 4): sample k of the run is stepped over when the samples before it were, its cell is empty
// (k < n1, the number of leading ones of the mask) and the inner loop's condition and the check after the sample
// let the ray go on -- t < t_exit (:790) and not t >= tfar (:868) -- i.e. t < lim = min(t_exit, tfar).  Once one of
// the two fails it fails for every later k (t stays), so the run needs no flag: one integer and one float compare,
// the add and a select per sample, where the nested ifs compiled to ~8 VALU and ~10 SALU instructions and a branch
// each.  The sample at which the run stops is looked at once, afterwards: still in an empty cell and t < t_exit, so
// t >= tfar -- the reference takes that sample (it composites nothing) and leaves the loop at :868.
template <int kLook>
VR_DEV uint32_t skip_empty_steps(uint32_t n1, const RayCtx &c, RayDyn &d, bool count, unsigned long long &c_taken)
{
    const float lim = vmin(d.t_exit, c.tfar);
    float tk = d.t;
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < kLook; ++k) {
        const bool adv = (uint32_t)k < n1 && tk < lim;
        n += adv ? 1u : 0u;
        tk = adv ? tk + c.stepSize : tk;
    }
    d.t = tk;
    const bool far_hit = n < (uint32_t)kLook && n < n1 && tk < d.t_exit;   // (t >= tfar: the sample is taken, then :868)
    if (count) c_taken += n + (far_hit ? 1u : 0u);
#ifdef VR_RAYLEN
    d.nsmp += n + (far_hit ? 1u : 0u);
#endif
    if (far_hit) d.state = S_DONE;
    return n;
}

// number of leading ones among the low kLook bits of a mask
template <int kLook>
VR_DEV uint32_t leading_ones(uint32_t mask)
{
    if (kLook < 32) return (uint32_t)__builtin_ctz(~mask | (1u << (kLook & 31)));
    return mask == 0xffffffffu ? 32u : (uint32_t)__builtin_ctz(~mask);
}

VR_DEV bool skip_empty_run(uint32_t mask, const RayCtx &c, RayDyn &d, bool count,
                           unsigned long long &c_taken)
{
    // leading samples in empty cells; a lane that is not sampling steps over nothing
    const uint32_t n1 = d.state == S_SAMPLE ? leading_ones<kLook1>(mask) : 0u;
    return skip_empty_steps<kLook1>(n1, c, d, count, c_taken) == (uint32_t)kLook1;
}

// Phase 2: the four lanes of a ray look at four consecutive windows of kLook2 samples; the run is
// then stepped over in chunks of kLook2 for as long as some ray of the wave is still skipping.
VR_DEV bool skip_empty_run_wide(const uint32_t (&masks)[4], const RayCtx &c, RayDyn &d, bool count,
                                unsigned long long &c_taken)
{
    bool run = d.state == S_SAMPLE;
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {
        if (!__ballot(run)) break;
        const uint32_t n1 = run ? leading_ones<kLook2>(masks[chunk]) : 0u;
        run = skip_empty_steps<kLook2>(n1, c, d, count, c_taken) == (uint32_t)kLook2;
    }
    return run;
}

// The lookahead costs a few hundred instructions for the whole wave: it runs when at least half
// of the sampling lanes expect their next sample to be empty (their last one was).
VR_DEV bool lookahead_pays(bool sampling, bool guess_empty)
{
    const int n_s = __builtin_popcountll(__ballot(sampling));
    const int n_g = __builtin_popcountll(__ballot(sampling && guess_empty));
    return n_g > 0 && VR_LOOK_NUM * n_g >= n_s;
}

#ifdef VR_EXPERIMENTS   // exact O(1) leaps over empty runs: opt-in experiments only (vr_experiments_leap.inc)
#include "vr_experiments_leap.inc"
#endif

// One front-to-back compositing step (:865-879) with the sample's colour*opacity (q0..q2),
// opacity qo and ray parameter ti.
VR_DEV void composite(const RayCtx &c, RayDyn &d, float q0, float q1, float q2, float qo, float ti)
{
    VR_RAYLEN_INC(d);
    float oma = 1.f - d.alpha;
    d.r0 = d.r0 - q0 * oma;
    d.r1 = d.r1 - q1 * oma;
    d.r2 = d.r2 - q2 * oma;
    d.alpha = d.alpha + qo * oma;
    // (double)alpha > 0.98 <=> alpha >= 0.98f (ERT_THRESHOLD, :28); `break`, then :882 breaks
    if (ti >= c.tfar || d.alpha >= 0.98f) {
        d.state = S_DONE;
        d.ert = !(ti >= c.tfar);   // :868 breaks before the ERT branch (:869-877) is looked at
        d.t_ert = ti;
    } else {
        d.t = ti + c.stepSize;
    }
}

// broadcast lane L of every quad (4 consecutive lanes): one DPP move, no LDS
template <int L> VR_DEV float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), L * 0x55, 0xf, 0xf, true));
}
template <int L> VR_DEV int quad_bcast(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, L * 0x55, 0xf, 0xf, true);
}

// ------------------------------------------------------------------ ambient occlusion

// hybrid Tausworthe / LCG generator on the per-pixel uint4 state (:50-80); the constant is a
// double literal: product in double, rounded to float on return
VR_DEV uint32_t taus_step(uint32_t &z, int s1, int s2, int s3, uint32_t m)
{
    const uint32_t b = (((z << (uint32_t)s1) ^ z) >> (uint32_t)s2);
    z = (((z & m) << (uint32_t)s3) ^ b);
    return z;
}
VR_DEV float hybrid_rand(uint32_t (&st)[4])
{
    const uint32_t a = taus_step(st[0], 13, 19, 12, 4294967294u);
    const uint32_t b = taus_step(st[1], 2, 25, 4, 4294967288u);
    const uint32_t c = taus_step(st[2], 3, 11, 17, 4294967280u);
    const uint32_t d = st[3];
    st[3] = 1664525u * st[3] + 1013904223u;
    return (float)(2.3283064365387e-10 * (double)(float)(a ^ b ^ c ^ d));
}

// calcAO (:368-392) at the sample that triggered early ray termination, with
// getUniformRandomSampleDirectionUpper (:353-364); scales the ray's colour by 1 - ao / 2 (:875).
// Rare mode, rolled loops.
template <typename VT, int INSTR, typename V>
VR_DEV void apply_ao(const V &vol, const float4 *s_tff, int tffn, const RayCtx &c,
                     RayDyn &d, const vrhip_rendering_params &rp, uint32_t gx, uint32_t gy)
{
    const f3 p0 = add3(c.cam, scale3(c.dir, d.t_ert - c.offset));
    const f3 pos = mk3(p0.x * 0.5f + 0.5f, p0.y * 0.5f + 0.5f, p0.z * 0.5f + 0.5f);
    const f3 n = vol.neg_gradient(pos.x, pos.y, pos.z);
    const float vl = len3(mk3(1.f / vol.fw, 1.f / vol.fh, 1.f / vol.fd));
    const float stepSize = vl * 0.9f, r = vl * 5.f;
    uint32_t st[4];
    st[0] = st[1] = st[2] = st[3] = parallel_rng3(gx, gy, rp.seed);   // :611
    float ao = 0.f;
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const float z = (hybrid_rand(st) * 2.f) - 1.f;
        const float phi = (hybrid_rand(st) * 2.f) * 3.14159274101257f;
        float sn, cs;
        vr_sincosf(phi, &sn, &cs);
        const float rad = sqrtf(1.f - z * z);
        f3 dir = mk3(rad * sn, rad * cs, z);
        if (dot3(n, dir) < 0) dir = mk3(dir.x * -1.f, dir.y * -1.f, dir.z * -1.f);
        float sample = 0.f;
        int cnt = 0;
#pragma unroll 1
        while ((float)cnt * stepSize < r) {
            ++cnt;
            const f3 p = add3(pos, scale3(scale3(dir, (float)cnt), stepSize));
            sample += tff_linear_alpha(s_tff, tffn, vol.linear(p.x, p.y, p.z));
            if (sample > 0.98f) break;
        }
        sample /= (float)cnt;
        ao += sample;
    }
    ao = ao / 16.f;
    const float f = 1.f - 0.5f * ao;
    d.r0 *= f; d.r1 *= f; d.r2 *= f;
}

// ---- image-order ESS (volumeraycast.cl:659-670, :912-925) and showEss (:888-896)

// what the work-items of an 8x8 work-group (= patch) did, for vr_hit_resolve_kernel
enum { HIT_SKIPPED = 0, HIT_FIRST_ENDS = 1, HIT_FIRST_MISSES = 2 };

// volumeraycast.cl:323-343 with bound = (0, 1)
VR_DEV bool check_bounding_box(f3 pos, f3 voxLen)
{
    const bool xl = pos.x < voxLen.x, xh = pos.x > 1.f - voxLen.x;
    const bool yl = pos.y < voxLen.y, yh = pos.y > 1.f - voxLen.y;
    const bool zl = pos.z < 0.f + voxLen.z, zh = pos.z > 1.f - voxLen.z;
    return (xl && zl) || (xl && yl) || (yl && zl) || (xh && zl) || (yh && zl) || (xh && zh) ||
           (yh && zh) || (xl && zh) || (yl && zh) || (xh && yl) || (xh && yh) || (xl && yh);
}

// getLastHit (:513-526): true when nothing was hit in or around this work-group last frame.
// Lanes 0..8 read one texel each; texels outside the hit image count as 0.
VR_DEV bool group_unhit(const FrameView &fr, uint32_t tx8, uint32_t ty8, uint32_t lane)
{
    uint32_t v = 0;
    if (lane < 9u) {
        const int x = (int)tx8 + (int)(lane % 3u) - 1, y = (int)ty8 + (int)(lane / 3u) - 1;
        if (x >= 0 && y >= 0 && x < (int)fr.hit_w && y < (int)fr.hit_h)
            v = fr.hit_in[(size_t)y * fr.hit_w + (size_t)x];
    }
    return __ballot(v != 0u) == 0ull;
}

// Runs once per patch, after its rays are set up.  Records what the group's first work-item
// will do (it has the last word on the hit texel, :918-924) and, for a group that is skipped,
// writes the background (:664-668).  Returns true for a skipped group.
VR_DEV bool image_ess_patch(const FrameView &fr, const vrhip_rendering_params &rp, const RayCtx &c,
                            const WaveTile &wt, uint32_t lane, bool inside, uint32_t gx, uint32_t gy,
                            size_t out_index)
{
    const bool unhit = group_unhit(fr, wt_col(wt), wt_row(wt), lane);
    const unsigned long long valid = __ballot(c.valid);
    if (lane == 0)
        fr.hit_status[(size_t)wt_row(wt) * fr.hit_w + wt_col(wt)] =
            (uint8_t)(unhit ? HIT_SKIPPED : ((valid & 1ull) ? HIT_FIRST_ENDS : HIT_FIRST_MISSES));
    if (unhit && inside) {
        float4 o = make_float4(c.env0, c.env1, c.env2, c.env3);
        if (rp.showEss) o = make_float4(1.f - o.x, 1.f - o.y, 1.f - o.z, 1.f - o.w);
        fr.fb[(size_t)gy * fr.W + gx] = o;
        if (fr.out) fr.out[out_index] = o;
    }
    return unhit;
}

// showEss (:888-896), running mean over iterations (:898-909, fp32 accumulate buffer), the two
// writes, and the image-order ESS hit flag (:912-917).  EXTRAS = false: the default kernels,
// which are never launched with showEss / imgEss set.
template <bool EXTRAS>
VR_DEV void write_pixel(const FrameView &fr, const vrhip_rendering_params &rp, const RayCtx &c,
                        const RayDyn &d, f3 voxLen, uint32_t gx, uint32_t gy, size_t out_index)
{
    const size_t fi = (size_t)gy * fr.W + gx;
    float r0 = d.r0, r1 = d.r1, r2 = d.r2, alpha = d.alpha;
    if (EXTRAS && rp.showEss && c.valid) {
        f3 pk = mk3(0.f, 0.f, 0.f);   // :722: the position of a ray that never sampled
        if (d.t_last >= 0.f) {
            const f3 pos = add3(c.cam, scale3(c.dir, d.t_last - c.offset));
            pk = mk3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f);
        }
        if (check_bounding_box(pk, voxLen)) {
            r0 = fabsf(1.f - rp.backgroundColor[0]);
            r1 = fabsf(1.f - rp.backgroundColor[1]);
            r2 = fabsf(1.f - rp.backgroundColor[2]);
            alpha = 1.f;
        }
    }
    if (rp.iteration != 0 && c.valid) {
        float4 prev = fr.fb[fi];
        float it1 = (float)(rp.iteration + 1u);
        r0 = prev.x + (r0 - prev.x) / it1;
        r1 = prev.y + (r1 - prev.y) / it1;
        r2 = prev.z + (r2 - prev.z) / it1;
    }
    float4 o = make_float4(r0, r1, r2, c.valid ? alpha : c.env3);
#ifdef VR_RAYLEN
    o.w = c.valid ? (float)d.nsmp : 0.f;
#endif
    fr.fb[fi] = o;
    if (fr.out) fr.out[out_index] = o;
    if (EXTRAS && rp.imgEss && c.valid && (r0 != c.env0 || r1 != c.env1 || r2 != c.env2))
        fr.hit_any[(size_t)(gy >> 3) * fr.hit_w + (gx >> 3)] = 1;
}

template <typename VT, int INSTR, bool FP, bool LS = false>
VR_DEV Vol<VT, INSTR, FP, LS> make_vol(const VolView &vv, uint32_t *touched)
{
    Vol<VT, INSTR, FP, LS> vol;
    vol.p = (const VT *)vv.data;
    vol.w1 = vv.w - 1; vol.h1 = vv.h - 1; vol.d1 = vv.d - 1;
    vol.fw = vv.fw; vol.fh = vv.fh; vol.fd = vv.fd;
    vol.inv_max = vv.inv_max;
    vol.nbx = vv.nbx; vol.nby = vv.nby;
    vol.ystride = vv.ystride; vol.zstride = (uint32_t)vv.zstride;
    vol.touched = touched;
    vol.pc[0] = (const VT *)vv.chan[0]; vol.pc[1] = (const VT *)vv.chan[1];
    vol.pc[2] = (const VT *)vv.chan[2];
    vol.channels = vv.channels;
    vol.fp = (const FpEntry<VT> *)vv.fp;
    vol.fp_ystride = vv.fp_nbx * 64u;
    vol.fp_zstride = vv.fp_nbx * vv.fp_nby * 64u;
    return vol;
}

VR_DEV Grid make_grid(const BrickView &bricks, const vrhip_raycast_params &rcp, uint32_t oob_word,
                      bool ess)
{
    Grid g;
    g.bw = bricks.bw; g.bh = bricks.bh; g.bd = bricks.bd;
    g.bl0 = g.bl1 = g.bl2 = 0.f;
    g.brickDia = 0.f;
    g.oob_word = oob_word;
    if (ess) {
        g.bl0 = 1.f / rcp.brickRes[0];
        g.bl1 = 1.f / rcp.brickRes[1];
        g.bl2 = 1.f / rcp.brickRes[2];
        g.brickDia = sqrtf(((g.bl0 * g.bl0) + (g.bl1 * g.bl1)) + (g.bl2 * g.bl2)) * 2.f;
    }
    return g;
}

VR_DEV void flush_counters(DevStats *stats, uint32_t lane, const unsigned long long (&c)[6])
{
    for (int i = 0; i < 6; ++i) {
        unsigned long long s = wave_sum(c[i]);
        if (lane == 0 && s) atomicAdd(&stats->v[i], s);
    }
}

// ------------------------------------------------------------------ patch culling

// maximum over the 64 lanes (all of them active), in every lane: six DPP steps -- row_shr 1, 2, 4, 8 leave
// each row's maximum in its lane 15, row_bcast15 / row_bcast31 carry it on to lane 63 -- and a readlane
template <int CTRL, int ROW_MASK>
VR_DEV float dpp_max_step(float v)
{
    // lanes without a source (and rows outside ROW_MASK) keep their own value: `old` = v
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
    return vmax(v, o);
}
VR_DEV float wave_max_f(float v)
{
    v = dpp_max_step<0x111, 0xf>(v);   // row_shr:1
    v = dpp_max_step<0x112, 0xf>(v);   // row_shr:2
    v = dpp_max_step<0x114, 0xf>(v);   // row_shr:4
    v = dpp_max_step<0x118, 0xf>(v);   // row_shr:8
    v = dpp_max_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// True when NO ray of the wave's 8x8 patch can visit a brick that is not skipped (SkipView::near_bits).
// Every point of every valid ray i, cam_i + t dir_i with t in [tn, tf] (the patch's smallest tnear and
// largest tfar), lies within rho = max|cam_i - cam_r| + tf max|dir_i - dir_r| of the point with the
// same t on a reference ray r of the patch; 256 test points on r (four per lane) leave no point of r
// farther than h / 2 from one of them.  The reference DDA of a ray only visits bricks that touch the
// ray within one brick (its crossing times are accumulated sums, off by far less than a brick), so
// all it can visit lies within floor((rho + h / 2) / brick) + 2 bricks of a test point's (clamped) brick
// (+1 because the test points' bricks are found with a reciprocal, good to a brick).
// If that is within the radius the bitmap was dilated by and every test point reads 0, every brick
// any of the rays visits is skipped: the rays end as they started.  Out-of-range cells read the
// (0, 0) decision, which must be "skip" for any of this to hold.
VR_DEV bool patch_is_clear(const SkipView &skip, const Grid &g, const RayCtx &c, uint32_t lane)
{
    const unsigned long long vm = __ballot(c.valid);
    if (!vm) return false;                       // (nothing to walk anyway: the normal path writes the pixels)
    if (skip.bits[skip.n_words] != 0xffffffffu) return false;
    const int ref = ((vm >> 27) & 1ull) ? 27 : (int)__builtin_ctzll(vm);   // a ray in the middle of the patch, if it has one
    const f3 rc = mk3(__shfl(c.cam.x, ref, 64), __shfl(c.cam.y, ref, 64), __shfl(c.cam.z, ref, 64));
    const f3 rd = mk3(__shfl(c.dir.x, ref, 64), __shfl(c.dir.y, ref, 64), __shfl(c.dir.z, ref, 64));
    const float dc = wave_max_f(c.valid ? len3(sub3(c.cam, rc)) : 0.f);
    const float dd = wave_max_f(c.valid ? len3(sub3(c.dir, rd)) : 0.f);
    const float tf = wave_max_f(c.valid ? c.tfar : -3.0e38f);
    const float tn = -wave_max_f(c.valid ? -c.tnear : -3.0e38f);
    if (!(tf > tn) || !(tf < 1.0e30f)) return false;
    const float h = (tf - tn) * (1.0f / 256.0f);
    const float need = (dc + tf * dd + 0.5f * h) * 1.0001f;
    const float bl[3] = {g.bl0, g.bl1, g.bl2};
    const int bres[3] = {g.bw, g.bh, g.bd};
    // a point within `need` of a test point lies at most floor(need / brick) + 1 bricks from the test
    // point's brick; one more for bricks the DDA visits next to the ray
    float ibs[3];   // bricks per world unit
    for (int i = 0; i < 3; ++i) {
        ibs[i] = 0.5f / bl[i];
        // (+3: one brick more than the derivation needs, for the reciprocal in the test points' bricks)
        if (!(floorf(need / (2.f * bl[i])) + 3.f <= (float)skip.near_r)) return false;
    }
    bool live = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = tn + ((float)(lane + 64u * (uint32_t)k) + 0.5f) * h;
        const float p[3] = {rc.x + t * rd.x, rc.y + t * rd.y, rc.z + t * rd.z};
        int cell[3];   // (good to a brick: the radius test above has a brick to spare for it)
        for (int i = 0; i < 3; ++i) cell[i] = iclamp((int)floorf((p[i] + 1.f) * ibs[i]), 0, bres[i] - 1);
        const uint32_t idx = ((uint32_t)cell[2] * (uint32_t)g.bh + (uint32_t)cell[1]) * (uint32_t)g.bw + (uint32_t)cell[0];
        live = live || ((skip.near_bits[idx >> 5] >> (idx & 31u)) & 1u);
    }
    return __ballot(live) == 0ull;
}

// ------------------------------------------------------------------ DDA pre-pass

// Most rays of a typical frame cross the volume without ever meeting a brick the ESS bitmap
// does not skip: all they do is the DDA walk.  In the marching kernels (two waves per SIMD) that
// walk is a latency chain; here it runs alone in a kernel small enough for high occupancy (the
// bitmap is read through the caches).  One wave per 8x8 patch, same set-up and dda_step as the
// march, hence the same decisions.  Rays that end without a sample write their pixel here;
// patches with rays that reach a brick to sample go to the `live` list for phase 1.
// Patch classes: one wave per 8x8 patch, ONCE per camera / parameters / skip bitmap / tile set -- the frames of
// a set, and the frames after it while nothing changes, differ only in the jitter seed, which moves every
// ray by less than a pixel on a square frame, by up to max(gsx, gsy) / gs pixels along an axis in general
// (:625-628).  The wave sets up 64 rays WITHOUT jitter on a regular grid over the patch's pixel positions grown
// by that jitter range on the + side and by two pixels on every side ([8 tx - 2, 8 tx + 10] x [8 ty - 2,
// 8 ty + 10] on a square frame): the jittered rays of the patch's pixels, in any frame, lie inside the hull
// of these.  Class 1 when
//  * all 64 hull rays hit the box SHRUNK by a thousandth of its size (rays through a convex box from one
//    eye point -- or parallel rays -- form a convex set, so every ray inside the hull hits the shrunk box,
//    and the real box by a margin far above the slab test's rounding): c.valid holds for every ray;
//  * patch_is_clear holds for the hull rays (its tube bounds |cam - cam_r| + t |dir - dir_r| by the maximum
//    over the 64 rays: the hull's corners, one pixel outside anything a jittered ray can reach) -- no ray of
//    the patch can visit a brick that is not skipped;
//  * the patch lies inside the frame, and the background is one colour (no gradient, no environment map:
//    checked by the host, with iteration 0, no showEss, no image-order ESS).
// Then every pixel of the patch ends as (backgroundColor.rgb, alpha 0) -- setup_ray_head's start values through
// write_pixel -- in every frame: the pre-pass writes that and returns, ~30 instructions instead of ~800.
__global__ __launch_bounds__(kBlockDim) void vr_patch_class_kernel(SkipView skip, FrameView fr, vrhip_camera_params cam,
                                                                   vrhip_rendering_params rp, Grid grid,
                                                                   uint32_t n_patches, uint32_t set_frames, uint8_t *cls)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pi = blockIdx.x * (kBlockDim / 64) + (threadIdx.x >> 6);
    if (pi >= n_patches) return;
    const WaveTile wt = fr.queue[(size_t)pi * set_frames];
    // make_ray's geometry (:614-650) at a fractional pixel position, no jitter
    const float *V = cam.viewMat;
    const f3 ms = mk3(rp.modelScale[0], rp.modelScale[1], rp.modelScale[2]);
    const int maxImg = (int)(fr.gsx > fr.gsy ? fr.gsx : fr.gsy);
    // The jitter moves a ray by rnd * 2 / gsx and rnd * 2 / gsy in NDC (:625-628) while a pixel is 2 / maxImg wide:
    // by up to maxImg / gsx pixels in +x and maxImg / gsy in +y -- one pixel on a square frame, `aspect` pixels
    // along the short axis of any other.  The hull is the patch's pixel positions [8 t, 8 t + 7] grown by that
    // (rounded up) on the + side and by two pixels of margin on both: [8 t - 2, 8 t + 9 + ceil(maxImg / gs)].
    const float spanx = (float)(11u + ((uint32_t)maxImg + fr.gsx - 1u) / fr.gsx);
    const float spany = (float)(11u + ((uint32_t)maxImg + fr.gsy - 1u) / fr.gsy);
    const float fx = (float)(wt_col(wt) * 8u) - 2.f + (float)(lane & 7u) * (spanx / 7.f);
    const float fy = (float)(wt_row(wt) * 8u) - 2.f + (float)(lane >> 3) * (spany / 7.f);
    float icx = (fx / (float)maxImg) * 2.f, icy = (fy / (float)maxImg) * 2.f;
    if (fr.gsx > fr.gsy) { icx -= 1.0f; icy -= fr.ray_aspect; }
    else { icx -= fr.ray_aspect; icy -= 1.0f; }
    icy *= -1.f;
    f3 npp = mk3(icx, icy, -1.0f);
    f3 rayDir = mk3(dot3(mk3(V[0], V[1], V[2]), npp), dot3(mk3(V[4], V[5], V[6]), npp), dot3(mk3(V[8], V[9], V[10]), npp));
    f3 camPos = mul3(mk3(V[3], V[7], V[11]), ms);
    if (cam.ortho) {
        camPos = mk3(V[3], V[7], V[11]);
        rayDir = neg3(mk3(V[2], V[6], V[10]));
        npp = add3(add3(camPos, scale3(mk3(V[0], V[4], V[8]), icx)), scale3(mk3(V[1], V[5], V[9]), icy));
        npp = scale3(npp, len3(camPos));
        camPos = mul3(npp, ms);
    }
    rayDir = normalize3(mul3(rayDir, ms));
    const float o[3] = {camPos.x, camPos.y, camPos.z}, dv[3] = {rayDir.x, rayDir.y, rayDir.z};
    float tn = -3.0e38f, tf = 3.0e38f, tns = -3.0e38f, tfs = 3.0e38f;
    for (int i = 0; i < 3; ++i) {
        const float inv = 1.0f / dv[i];
        const float m = 1.0e-3f * (cam.bbox_tr[i] - cam.bbox_bl[i]);
        const float tb = inv * (cam.bbox_bl[i] - o[i]), tt = inv * (cam.bbox_tr[i] - o[i]);
        const float tbs = inv * (cam.bbox_bl[i] + m - o[i]), tts = inv * (cam.bbox_tr[i] - m - o[i]);
        tn = vmax(tn, vmin(tt, tb)); tf = vmin(tf, vmax(tt, tb));
        tns = vmax(tns, vmin(tts, tbs)); tfs = vmin(tfs, vmax(tts, tbs));
    }
    const bool hit_shrunk = (tfs > tns) && !(tfs < 0.f) && (tfs - tns) > 0.f;
    RayCtx c;
    c.cam = camPos;
    c.dir = rayDir;
    c.valid = (tf > tn) && !(tf < 0.f);
    c.tnear = vmax(0.f, tn);
    c.tfar = tf;
    const bool inside = wt_col(wt) * 8u + 8u <= fr.W && wt_row(wt) * 8u + 8u <= fr.H;   // (uniform)
    const bool all_hit = __ballot(hit_shrunk) == ~0ull;
    bool clear = false;
    if (inside && all_hit && skip.near_bits) clear = patch_is_clear(skip, grid, c, lane);
    if (lane == 0) cls[pi] = clear ? 1 : 0;
}

template <typename VT>
__global__ __launch_bounds__(kBlockDim) void vr_dda_prepass_kernel(
    VolView vv, BrickView bricks, SkipView skip, FrameView fr, vrhip_camera_params cam,
    vrhip_rendering_params rp, vrhip_raycast_params rc, Grid grid, f3 voxLen)
{
    // (grid, voxLen: make_grid's and 1 / resolution's values, computed once on the host with the same
    // IEEE operations -- a wave lives for one patch here, and the divisions and the square root were
    // 90 of its ~1250 instructions)
    VR_ZERO_NEXT_CTRL(fr);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * (kBlockDim / 64) + (threadIdx.x >> 6);
    if (q >= fr.n_wave_tiles) return;
    const WaveTile wt = fr.queue[q];
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t gx = wt_col(wt) * 8u + lx, gy = wt_row(wt) * 8u + ly;
    const size_t out_index = (size_t)wt.out_base + (size_t)ly * fr.out_stride + lx;
    if (fr.patch_class && fr.patch_class[q / fr.set_frames]) {
        // class 1 (vr_patch_class_kernel): every ray of this patch -- in any frame of the set -- hits the box,
        // meets no brick to sample, and the background is one colour: what the walk would leave, without a ray
        const float4 o = make_float4(rp.backgroundColor[0], rp.backgroundColor[1], rp.backgroundColor[2], 0.f);
        fr.fb[(size_t)gy * fr.W + gx] = o;
        if (fr.out) fr.out[out_index] = o;
        return;
    }
    const uint32_t seed = fr.seeds ? fr.seeds[wt_frame(wt)] : rp.seed;
    const bool inside = gx < fr.W && gy < fr.H;
    const f3 resf = mk3(vv.fw, vv.fh, vv.fd);
    RayCtx c;
    RayDyn d;
    float rnd;
    setup_ray_head<false>(gx, gy, inside, fr, cam, rp, c, d, seed, rnd);   // (nothing is shaded here)
    if (rp.imgEss && image_ess_patch(fr, rp, c, wt, lane, inside, gx, gy, out_index)) return;
    if (skip.near_bits && patch_is_clear(skip, grid, c, lane)) {
        // no ray of this patch can meet a brick that is not skipped: what the walk would leave
        // (step size and DDA set-up are not needed for that)
        if (inside) write_pixel<true>(fr, rp, c, d, voxLen, gx, gy, out_index);
        return;
    }
    setup_ray_tail<true>(rc, resf, voxLen, grid, c, d, rnd);
    fetch_skip_word(skip.bits, grid, d);
    unsigned long long n0 = 0, n1 = 0;
#ifdef VR_MARCH_STATS
    unsigned long long w_steps = 0, l_steps = 0;
    while (__ballot(d.state == S_BRICK)) {
        w_steps++;
        l_steps += __builtin_popcountll(__ballot(d.state == S_BRICK));
        dda_step<0>(skip.bits, grid, c, d, n0, n1);
    }
    if (lane == 0) {
        atomicAdd(&g_march_stats[28], 1ull);                                     // patches that walk
        atomicAdd(&g_march_stats[29], w_steps);                                  // DDA step executions
        atomicAdd(&g_march_stats[30], l_steps);                                  // lanes in them
        atomicAdd(&g_march_stats[31], (unsigned long long)__builtin_popcountll(__ballot(c.valid)));   // valid rays
    }
#else
    while (__ballot(d.state == S_BRICK)) dda_step<0>(skip.bits, grid, c, d, n0, n1);
#endif
    const bool live = d.state == S_SAMPLE;
    const unsigned long long m = __ballot(live);
    if (inside && !live) {
        // what the march would leave for a ray without samples: background colour, alpha 0
        write_pixel<true>(fr, rp, c, d, voxLen, gx, gy, out_index);
    }
    if (fr.live_rays) {
        // ray list for phase 1 (vr_raycast_rays_kernel): the live rays with the DDA state they have
        // reached, so that phase 1 neither repeats the walk nor carries the patch's dead lanes
        if (m) {
            // (list q % kLiveLists: vr_internal.h)
            const uint32_t list = q % kLiveLists;
            uint32_t base = 0;
#ifdef VR_DIAG_NO_LIVE_ATOMIC   // diagnostic build (WRONG frames: the lists are not compact and their counts stay 0, so the
            base = (q / kLiveLists) * 64u;   // marching kernels leave at once): what does the pre-pass cost without its counters?
#else
            if (lane == (uint32_t)__builtin_ctzll(m))
                base = atomicAdd(fr.live_list_count + list * kLiveStride, (uint32_t)__builtin_popcountll(m));
            base = __shfl(base, __builtin_ctzll(m), 64);
#endif
            base += list * live_list_cap(fr.n_wave_tiles);
            if (live) {
                ContRec r;
                r.pix = gx | (gy << 16);
                r.out_index = (uint32_t)out_index;
                r.state = d.state | (int32_t)(wt_frame(wt) << 8);
                r.t = d.t; r.t_exit = d.t_exit; r.alpha = d.alpha;
                r.r0 = d.r0; r.r1 = d.r1; r.r2 = d.r2;
                r.cx = d.c0; r.cy = d.c1; r.cz = d.c2;
                r.tv0 = d.tv0; r.tv1 = d.tv1; r.tv2 = d.tv2;
                r.pad = 0;
                fr.live_rays[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = r;
            }
        }
        return;
    }
    if (m && lane == 0) {
        const uint32_t slot = atomicAdd(fr.live_count, 1u);
        LiveTile lt;
        lt.wt = wt;
        lt.mask_lo = (uint32_t)m;
        lt.mask_hi = (uint32_t)(m >> 32);
        fr.live[slot] = lt;
    }
}

// ------------------------------------------------------------------ phase 1 on the ray list

// Phase 1 for the default modes with ESS: one lane per ray, the rays taken from the pre-pass's ray
// list (FrameView::live_rays) with the DDA state reached there; a lane whose ray ends -- or is
// suspended for phase 2 after `round_budget` rounds of its own -- takes the next ray once
// 4 * refill_min lanes of the wave are idle (default 64: the whole wave, measured best).  Same per-ray operation sequence as the patch kernel above; no dead
// lanes carried through a patch, no second DDA walk.  Exit condition reached by every wave: the
// list head only grows, and every ray ends or is suspended.
// WAVES: waves per workgroup.  4 (256 threads): the compiler's register choice (184 VGPRs), two workgroups per CU by
// their 72 KiB of LDS = two waves per SIMD.  12 (768 threads): launch bounds that leave 170 VGPRs (168 used, three to
// six spilled), ONE workgroup per CU = three waves per SIMD that share one transfer function and one skip bitmap in
// LDS (60 + 16 + 32 KiB): pays where waves wait more than they issue -- launch sets of several frames, volumes whose
// ESS bricks are too small for the empty-run lookahead -- and not one frame at a time with the lookahead, where a
// third wave only stretches the chain of the longest rays.  See launch_variant.
constexpr int kWavesWide = 12;

template <typename VT, bool SKIP_LDS, bool FP, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64) void vr_raycast_rays_kernel(
    VolView vv, BrickView bricks, TfView tf, SkipView skip, CellView cells, FrameView fr,
    vrhip_camera_params cam, vrhip_rendering_params rp, vrhip_raycast_params rc)
{
    // The pre-pass (previous kernel on the stream) has written kLiveLists lists; they are read interleaved, 64 rays from
    // each in turn: virtual ray v is ray (v / 64 / kLiveLists) * 64 + v % 64 of list (v / 64) % kLiveLists, and there are
    // kLiveLists * 64 * ceil(longest list / 64) virtual rays (those beyond a list's end do not exist: idle lanes).
    __shared__ uint32_t s_live_n[kLiveLists];
    uint32_t longest = 0;
#pragma unroll
    for (uint32_t k = 0; k < kLiveLists; ++k) {
        const uint32_t n_k = fr.live_list_count[k * kLiveStride];
        longest = n_k > longest ? n_k : longest;
        if (threadIdx.x == k) s_live_n[k] = n_k;
    }
    const uint32_t n_rays = ((longest + 63u) >> 6) * kLiveLists * 64u;
    if (n_rays == 0) return;
    const uint32_t list_cap = live_list_cap(fr.n_wave_tiles);
    extern __shared__ float4 s_mem[];
    float *s_stage = reinterpret_cast<float *>(s_mem) + (threadIdx.x >> 6) * kStageFloatsPerWave;
    float4 *s_tff = s_mem + WAVES * kStageFloatsPerWave / 4;
    uint32_t *s_skip = reinterpret_cast<uint32_t *>(s_tff + tf.tff_n);
    for (uint32_t i = threadIdx.x; i < tf.tff_n; i += WAVES * 64) s_tff[i] = tf.tff[i];
    if (SKIP_LDS)
        for (uint32_t i = threadIdx.x; i <= skip.n_words; i += WAVES * 64) s_skip[i] = skip.bits[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const int tffn = (int)tf.tff_n;
    const Vol<VT, 0, FP> vol = make_vol<VT, 0, FP>(vv, nullptr);
    const f3 resf = mk3(vol.fw, vol.fh, vol.fd);
    const f3 voxLen = mk3(1.f / vol.fw, 1.f / vol.fh, 1.f / vol.fd);
    const float refInterval = 1.f / rc.samplingRate;
    const Grid grid = make_grid(bricks, rc, skip.n_words, true);
    const uint32_t *sb = SKIP_LDS ? s_skip : skip.bits;
    const bool skip_empty = cells.empty != nullptr && rp.useLinear != 0;
#ifdef VR_LEAP   // A/B build (VR_EXPERIMENTS + VR_LEAP_STEPPING); VRHIP_MARCH_MICRO = leap steps per round
    const bool use_mask = skip_empty && cells.bmask != nullptr && fr.march_micro != 0;
    const uint32_t leap_iters = fr.march_micro;
#endif
    const uint32_t budget = fr.round_budget ? fr.round_budget : 0xffffffffu;
    const uint32_t kRefillLanes = (fr.refill_min ? fr.refill_min : 16u) * 4u;   // idle lanes before a refill

    unsigned long long n0 = 0, n1 = 0;
    bool have = false, drained = false, first_draw = true;
    uint32_t gx = 0, gy = 0, out_index = 0, my_rounds = 0, frame_idx = 0;
    bool guess_empty = true;
    uint32_t cool = 0;   // evaluation batches before the ray guesses "empty" again (see the lookahead below)
#ifdef VR_LEAP
    LeapCache lc;
    lc.key0 = lc.key1 = 0xffffffffu; lc.m0 = lc.m1 = 0ull; lc.du = lc.dv = lc.ds = lc.inv_step = 0.f;
#endif
#ifdef VR_MARCH_STATS
    unsigned long long ms_acc[16] = {0};
#endif
    RayCtx c;
    RayDyn d;
    setup_ray<true>(0u, 0u, false, fr, cam, rp, rc, resf, voxLen, grid, c, d, rp.seed);   // S_DONE

    for (;;) {
        VR_MARK("R_top");
        {
            const bool idle = d.state == S_DONE;
            const unsigned long long idle_m = __ballot(idle);
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle_m);
            const bool all_idle = idle_m == ~0ull;
            if ((!drained && n_idle >= kRefillLanes) || all_idle) {
                VR_MS(8, 1);                                                   // refills
                if (idle && have) {   // retire a finished ray (suspended ones have given up `have`)
                    write_pixel<false>(fr, rp, c, d, voxLen, gx, gy, (size_t)out_index);
                    have = false;
                }
                if (!drained) {
                    // a wave's FIRST 64 rays are its own by position -- no ticket: 2 048 - 3 072 waves drawing from one
                    // counter at the same instant is a queue at one L2 address before anything marches --, the
                    // later ones are drawn behind those
                    uint32_t base = 0;
                    if (first_draw) {
                        base = (blockIdx.x * (uint32_t)WAVES + (threadIdx.x >> 6)) * 64u;
                    } else {
                        if (lane == 0) base = atomicAdd(fr.queue_head, n_idle);
                        base = __builtin_amdgcn_readfirstlane(base) + gridDim.x * (uint32_t)WAVES * 64u;
                    }
                    first_draw = false;
                    if (base + n_idle >= n_rays) drained = true;
                    if (idle) {
                        const uint32_t v = base + (uint32_t)__builtin_popcountll(idle_m & ((1ull << lane) - 1ull));
                        const uint32_t chunk = v >> 6, list = chunk % kLiveLists;
                        const uint32_t pos = ((chunk / kLiveLists) << 6) | (v & 63u);
                        have = v < n_rays && pos < s_live_n[list];
                        if (have) {
                            const ContRec rec = fr.live_rays[list * list_cap + pos];
                            gx = rec.pix & 0xffffu;
                            gy = rec.pix >> 16;
                            out_index = rec.out_index;
                            frame_idx = (uint32_t)rec.state >> 8;
                            setup_ray<true>(gx, gy, true, fr, cam, rp, rc, resf, voxLen, grid, c, d,
                                            fr.seeds ? fr.seeds[frame_idx] : rp.seed);
                            d.state = rec.state & 0xff;
                            d.t = rec.t; d.t_exit = rec.t_exit; d.alpha = rec.alpha;
                            d.r0 = rec.r0; d.r1 = rec.r1; d.r2 = rec.r2;
                            d.c0 = rec.cx; d.c1 = rec.cy; d.c2 = rec.cz;
                            d.tv0 = rec.tv0; d.tv1 = rec.tv1; d.tv2 = rec.tv2;
                            fetch_skip_word(sb, grid, d);
                            my_rounds = 0;
                            guess_empty = true;
                            cool = 0;
#ifdef VR_LEAP
                            leap_reset(lc, c, vol.fw, vol.fh, vol.fd);
#endif
                        }
                    }
                }
                if (!__ballot(d.state != S_DONE)) {
                    if (drained) break;
                    continue;
                }
            }
        }
        VR_MARK("R_dda");
        // ---- one round (the patch kernel's)
        VR_MS(0, 1);                                                           // rounds
        VR_MS(1, __builtin_popcountll(__ballot(d.state != S_DONE)));           // live lanes, summed over rounds
        for (int it = 0;; ++it) {
            if (!__ballot(d.state == S_BRICK)) break;
            if (it >= kMaxBrickSteps && __ballot(d.state == S_SAMPLE)) break;
            VR_MS(2, 1);                                                       // DDA step executions
            VR_MS(3, __builtin_popcountll(__ballot(d.state == S_BRICK)));      // lanes in them
            dda_step<0>(sb, grid, c, d, n0, n1);
        }
        if (!__ballot(d.state != S_DONE)) continue;
        VR_MARK("R_susp");
        // a ray that has used its rounds goes to the continuation buffer (phase 2)
        {
            const bool susp = d.state != S_DONE && my_rounds >= budget;
            const unsigned long long cm = __ballot(susp);
            if (cm) {
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(cm))
                    base = atomicAdd(fr.cont_count, (uint32_t)__builtin_popcountll(cm));
                base = __shfl(base, __builtin_ctzll(cm), 64);
                if (susp) {
                    ContRec r;
                    r.pix = gx | (gy << 16);
                    r.out_index = out_index;
                    r.state = d.state | (int32_t)(frame_idx << 8);
                    r.t = d.t; r.t_exit = d.t_exit; r.alpha = d.alpha;
                    r.r0 = d.r0; r.r1 = d.r1; r.r2 = d.r2;
                    r.cx = d.c0; r.cy = d.c1; r.cz = d.c2;
                    r.tv0 = d.tv0; r.tv1 = d.tv1; r.tv2 = d.tv2;
                    r.pad = fr.cost ? (uint32_t)fr.cost[(size_t)gy * fr.W + gx] : 0u;   // sort key
                    fr.cont[base + (uint32_t)__builtin_popcountll(cm & ((1ull << lane) - 1ull))] = r;
                    d.state = S_DONE;
                    have = false;
                }
                if (!__ballot(d.state != S_DONE)) continue;
            }
        }
        VR_MARK("R_look");
        if (__ballot(d.state == S_SAMPLE)) my_rounds += d.state == S_SAMPLE ? 1u : 0u;
        bool more_empty = false;
#ifdef VR_LEAP
        if (use_mask) {
            // steps over runs of empty samples (leap_step) until the ray stands at a sample to evaluate;
            // rays that leave their segment go on with the DDA in the same pass
            bool ready = false;
            for (uint32_t it = 0; it < leap_iters; ++it) {
                const bool act = d.state == S_SAMPLE && !ready;
                if (!__ballot(act)) break;
                if (act) ready = leap_step<true>(cells, vol, grid, c, d, lc, false, n0);
                if (__ballot(d.state == S_BRICK)) dda_step<0>(sb, grid, c, d, n0, n1);
            }
            more_empty = !ready;
        } else
#endif
        if (skip_empty && lookahead_pays(d.state == S_SAMPLE, guess_empty)) {
            VR_MS(4, 1);                                                       // lookahead executions
            VR_MS(5, __builtin_popcountll(__ballot(d.state == S_SAMPLE)));     // lanes in them
            if (d.state == S_SAMPLE) {
                const uint32_t em = empty_mask<VT, 0, kLook1>(cells, vol, c, d.t);
                more_empty = skip_empty_run(em, c, d, false, n0);
                guess_empty = (em & 1u) != 0u;
                // a transparent sample in a cell that is NOT empty (the rim of a structure) says little
                // about the samples behind it: no new guess for the next `cool` batches.  How long: as
                // many batches as the mask shows non-empty samples ahead
                if (!(em & 1u)) cool = (uint32_t)(__builtin_ctz(em | 0x10000u) / kBatch);
                after_segment<true>(c, d);
            }
        }
        VR_MARK("R_batch");
        if (__ballot(d.state == S_SAMPLE && !more_empty)) {
            VR_MS(6, 1);                                                       // evaluation batches
            VR_MS(7, __builtin_popcountll(__ballot(d.state == S_SAMPLE && !more_empty)));   // lanes in them
        }
        // the evaluation batch as wave-uniform code in three parts (eval_front / eval_dense / eval_back): every
        // lane of the wave works in the dense pass, whether its own ray evaluates this round or not
        const bool ev = d.state == S_SAMPLE && !more_empty;
        if (__ballot(ev)) {
            float tk[kBatch];
            bool vk[kBatch];
            tk[0] = d.t;
            vk[0] = ev && d.t < d.t_exit;   // inner loop condition (:790)
#pragma unroll
            for (int k = 1; k < kBatch; ++k) {
                tk[k] = tk[k - 1] + c.stepSize;                                     // :879
                vk[k] = vk[k - 1] && !(tk[k - 1] >= c.tfar) && (tk[k] < d.t_exit);  // :868, :790
            }
#ifdef VR_MARCH_STATS
            for (int k = 0; k < kBatch; ++k) ms_acc[9] += vk[k] ? 1 : 0;      // valid samples evaluated (per lane: summed below)
#endif
            EvalFront ef;
            const uint32_t ns = eval_front<VT, FP>(vol, s_tff, tffn, s_stage, c, rp, tk, vk, ev, ef);
            if (ns) eval_dense<VT, FP>(vol, s_stage, c, refInterval, ns);
            VR_MARK("R_comp");
            float p0[kBatch], p1[kBatch], p2[kBatch], opk[kBatch];
            eval_back(s_stage, c, rp, ef, ns, p0, p1, p2, opk);
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
                if (vk[k] && d.state == S_SAMPLE) composite(c, d, p0[k], p1[k], p2[k], opk[k], tk[k]);
            if (ev) {
                if (vk[kBatch - 1]) guess_empty = opk[kBatch - 1] == 0.f && cool == 0u;
                if (cool) --cool;
                after_segment<true>(c, d);
            }
        }
    }
#ifdef VR_MARCH_STATS
    if (lane == 0)
        for (int i = 0; i < 9; ++i) atomicAdd(&g_march_stats[16 + i], ms_acc[i]);
    {
        unsigned long long w = wave_sum(ms_acc[9]);
        if (lane == 0) atomicAdd(&g_march_stats[16 + 9], w);
    }
#endif
}

// ------------------------------------------------------------------ phase 1

template <typename VT, bool ESS, int INSTR, bool SKIP_LDS, bool XS, bool FP>
__global__ __launch_bounds__(kBlockDim) VR_OCC void vr_raycast_kernel(
    VolView vv, BrickView bricks, TfView tf, SkipView skip, CellView cells, FrameView fr,
    vrhip_camera_params cam, vrhip_rendering_params rp, vrhip_raycast_params rc, DevStats *stats,
    uint32_t *touched)
{
    VR_ZERO_NEXT_CTRL(fr);   // (also when the pre-pass has done it: the block stays unused until the next set)
    // LDS: [gradient staging, 4 waves][tff_n float4][skip words + 1]
    extern __shared__ float4 s_mem[];
    VR_STAMP_DECL;
    float *s_stage = reinterpret_cast<float *>(s_mem) + (threadIdx.x >> 6) * kStageFloatsPerWave;
    float4 *s_tff = s_mem + kStageF4;
    uint32_t *s_skip = reinterpret_cast<uint32_t *>(s_tff + tf.tff_n);
    for (uint32_t i = threadIdx.x; i < tf.tff_n; i += kBlockDim) s_tff[i] = tf.tff[i];
    if (ESS && SKIP_LDS)
        for (uint32_t i = threadIdx.x; i <= skip.n_words; i += kBlockDim) s_skip[i] = skip.bits[i];
    __syncthreads();
    VR_STAMP(8);

    const uint32_t lane = threadIdx.x & 63u;
    const int tffn = (int)tf.tff_n;
    unsigned long long c_taken = 0, c_nominal = 0, c_shaded = 0, c_bricks = 0, c_skipped = 0,
                       c_hit = 0;
    const Vol<VT, INSTR, FP> vol = make_vol<VT, INSTR, FP>(vv, touched);
    const f3 resf = mk3(vol.fw, vol.fh, vol.fd);
    const f3 voxLen = mk3(1.f / vol.fw, 1.f / vol.fh, 1.f / vol.fd);
    const float refInterval = 1.f / rc.samplingRate;
    const Grid grid = make_grid(bricks, rc, skip.n_words, ESS);
    const uint32_t *sb = SKIP_LDS ? s_skip : skip.bits;
    // empty-run skipping needs the linear sampler's footprint; the traffic-instrumented variant
    // reproduces the reference's fetch set instead
    const bool skip_empty = INSTR != 2 && cells.empty != nullptr && rp.useLinear != 0 &&
                            !(XS && (rp.illumType == 4 || rp.showEss));   // showEss tracks every sample

    // every wave pulls 8x8 patches until the queue is drained (exit condition reached by every
    // wave: the head only grows).  The next ticket is drawn while the current patch is marched,
    // which hides the contended atomic; every wave draws exactly one ticket past the end.
    const bool use_live = ESS && INSTR == 0 && fr.live != nullptr;
    const uint32_t n_tiles = use_live ? *fr.live_count : fr.n_wave_tiles;
    uint32_t q_next = 0;
    if (lane == 0) q_next = atomicAdd(fr.queue_head, 1u);
    for (;;) {
        const uint32_t q = __builtin_amdgcn_readfirstlane(q_next);
        if (q >= n_tiles) break;
        WaveTile wt;
        unsigned long long live_mask = ~0ull;
        if (use_live) {
            const LiveTile lt = fr.live[q];
            wt = lt.wt;
            live_mask = (unsigned long long)lt.mask_lo | ((unsigned long long)lt.mask_hi << 32);
        } else {
            wt = fr.queue[q];
        }
        if (lane == 0) q_next = atomicAdd(fr.queue_head, 1u);
        VR_STAMP(0);
        VR_COUNT(11);
        const uint32_t lx = lane & 7u, ly = lane >> 3;
        const uint32_t gx = wt_col(wt) * 8u + lx, gy = wt_row(wt) * 8u + ly;
        const uint32_t frame_idx = wt_frame(wt);
        const uint32_t seed = fr.seeds ? fr.seeds[frame_idx] : rp.seed;
        const bool inside = gx < fr.W && gy < fr.H;

        RayCtx c;
        RayDyn d;
        setup_ray<ESS>(gx, gy, inside, fr, cam, rp, rc, resf, voxLen, grid, c, d, seed);
        if (XS && rp.imgEss && !use_live &&   // (with a live list the pre-pass has done this)
            image_ess_patch(fr, rp, c, wt, lane, inside, gx, gy,
                            (size_t)wt.out_base + (size_t)ly * fr.out_stride + lx))
            continue;
        // rays the pre-pass has finished (their pixel is written) stay out of the march
        const bool prepass_done = !((live_mask >> lane) & 1ull);
        if (prepass_done) d.state = S_DONE;
        if (ESS) fetch_skip_word(sb, grid, d);
        if (INSTR && c.valid) { c_hit++; c_nominal += (unsigned long long)c.nominal; }
        VR_STAMP(1);

        // ---- flattened DDA / sample state machine, at most round_budget sample rounds
        uint32_t rounds_left = fr.round_budget ? fr.round_budget : 0xffffffffu;
        bool suspended = false;
        bool guess_empty = true;   // this ray's next sample is expected to lie in an empty cell
        for (;;) {
            VR_COUNT(10);
            if (ESS) {
                for (int it = 0;; ++it) {
                    if (!__ballot(d.state == S_BRICK)) break;
                    if (it >= kMaxBrickSteps && __ballot(d.state == S_SAMPLE)) break;
                    VR_COUNT(9);
                    dda_step<INSTR>(sb, grid, c, d, c_bricks, c_skipped);
                }
            }
            VR_STAMP(2);
            if (!__ballot(d.state != S_DONE)) break;
            if (rounds_left == 0) { suspended = true; break; }
            if (__ballot(d.state == S_SAMPLE)) --rounds_left;
            bool more_empty = false;
            if (skip_empty && lookahead_pays(d.state == S_SAMPLE, guess_empty)) {
                if (d.state == S_SAMPLE) {
                    // ---- step over a run of up to kLook1 samples in empty cells
                    const uint32_t em = empty_mask<VT, INSTR, kLook1>(cells, vol, c, d.t);
                    more_empty = skip_empty_run(em, c, d, INSTR != 0, c_taken);
                    guess_empty = (em & 1u) != 0u;
                    after_segment<ESS>(c, d);
                }
            }
            VR_STAMP(4);
            if (d.state == S_SAMPLE && !more_empty) {
                // ---- up to kBatch consecutive samples of this ray per round
                float tk[kBatch];
                bool vk[kBatch], litk[kBatch];
                tk[0] = d.t;
                vk[0] = d.t < d.t_exit;   // inner loop condition (:790)
#pragma unroll
                for (int k = 1; k < kBatch; ++k) {
                    tk[k] = tk[k - 1] + c.stepSize;                                     // :879
                    vk[k] = vk[k - 1] && !(tk[k - 1] >= c.tfar) && (tk[k] < d.t_exit);  // :868, :790
                    // the traffic-instrumented variant must not touch speculative voxels
                    if (INSTR == 2) vk[k] = false;
                }
                float p0[kBatch], p1[kBatch], p2[kBatch], opk[kBatch];
                eval_batch<VT, INSTR, XS, FP>(vol, s_tff, tffn, s_stage, c, rp, rc, refInterval, tk, vk, p0, p1, p2,
                                      opk, litk);
                VR_STAMP(3);
                // sequential front-to-back compositing (:865-879)
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    if (vk[k] && d.state == S_SAMPLE) {
                        if (INSTR) { c_taken++; if (litk[k]) c_shaded++; }
                        if (XS) d.t_last = tk[k];
                        composite(c, d, p0[k], p1[k], p2[k], opk[k], tk[k]);
                    }
                }
                if (vk[kBatch - 1]) guess_empty = opk[kBatch - 1] == 0.f;
                after_segment<ESS>(c, d);
                VR_STAMP(6);
            }
        }

        // rays that outlived the budget go to the continuation buffer (phase 2)
        const bool cont = suspended && d.state != S_DONE;
        const unsigned long long cm = __ballot(cont);
        if (cm) {
            uint32_t base = 0;
            if (lane == (uint32_t)__builtin_ctzll(cm))
                base = atomicAdd(fr.cont_count, (uint32_t)__builtin_popcountll(cm));
            base = __shfl(base, __builtin_ctzll(cm), 64);
            if (cont) {
                ContRec r;
                r.pix = gx | (gy << 16);
                r.out_index = wt.out_base + ly * fr.out_stride + lx;
                r.state = d.state | (int32_t)(frame_idx << 8);   // (states fit 8 bits)
                r.t = d.t; r.t_exit = d.t_exit; r.alpha = d.alpha;
                r.r0 = d.r0; r.r1 = d.r1; r.r2 = d.r2;
                r.cx = d.c0; r.cy = d.c1; r.cz = d.c2;
                r.tv0 = d.tv0; r.tv1 = d.tv1; r.tv2 = d.tv2;
                r.pad = fr.cost ? (uint32_t)fr.cost[(size_t)gy * fr.W + gx] : 0u;   // sort key
#ifdef VR_RAYLEN
                r.pad = d.nsmp;
#endif
                const uint32_t rank = (uint32_t)__builtin_popcountll(cm & ((1ull << lane) - 1ull));
                fr.cont[base + rank] = r;
            }
        }
        if (XS && rc.useAO && inside && !cont && !prepass_done && d.ert)
            apply_ao<VT, INSTR>(vol, s_tff, tffn, c, d, rp, gx, gy);
        if (inside && !cont && !prepass_done)
            write_pixel<XS>(fr, rp, c, d, voxLen, gx, gy,
                            (size_t)wt.out_base + (size_t)ly * fr.out_stride + lx);
        VR_STAMP(7);
    }
    VR_STAMP_FLUSH_AT(0);

    if (INSTR) {
        const unsigned long long cc[6] = {c_taken, c_nominal, c_shaded, c_bricks, c_skipped, c_hit};
        flush_counters(stats, lane, cc);
    }
}

// ------------------------------------------------------------------ phase 2

// composite the kBatch samples evaluated by lane O of every quad, in order
template <int O>
VR_DEV void composite_from(const RayCtx &c, RayDyn &d, const float (&p0)[kBatch],
                           const float (&p1)[kBatch], const float (&p2)[kBatch],
                           const float (&op)[kBatch], const float (&tk)[kBatch],
                           const int (&fl)[kBatch], bool count, unsigned long long &c_taken,
                           unsigned long long &c_shaded)
{
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
        const float q0 = quad_bcast<O>(p0[k]), q1 = quad_bcast<O>(p1[k]), q2 = quad_bcast<O>(p2[k]);
        const float qo = quad_bcast<O>(op[k]), ti = quad_bcast<O>(tk[k]);
        const int f = quad_bcast<O>(fl[k]);
        if ((f & 1) && d.state == S_SAMPLE) {
            if (count) { c_taken++; if (f & 2) c_shaded++; }
            composite(c, d, q0, q1, q2, qo, ti);
        }
    }
}

// Resumes suspended rays with kSplit = 4 lanes per ray (16 rays per wave).  The 4 lanes of a ray
// hold the same state and take the same decisions; lane `slot` evaluates samples
// 4*slot .. 4*slot+3 of the next 16 consecutive samples (same batch code as phase 1), then every
// lane replays the compositing of all 16 in ray order, fetching the other lanes' results with
// in-quad DPP broadcasts -- the fp32 operation sequence per ray is exactly phase 1's (and the
// reference's), the serial chain of a long ray is 4x shorter.
template <typename VT, bool ESS, int INSTR, bool SKIP_LDS, bool XS, bool FP, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64) void vr_raycast_split_kernel(
    VolView vv, BrickView bricks, TfView tf, SkipView skip, CellView cells, FrameView fr,
    vrhip_camera_params cam, vrhip_rendering_params rp, vrhip_raycast_params rc, DevStats *stats,
    uint32_t *touched)
{
    static_assert(kSplit == 4 && (kBatch == 4 || kBatch == 8), "phase 2 is written for 4 lanes x 4 (8: A/B build) samples");
    const uint32_t n_rays = *fr.cont_count;   // written by phase 1 (previous kernel on the stream)
    if (n_rays == 0) return;
    extern __shared__ float4 s_mem[];
    VR_STAMP_DECL;
    float *s_stage = reinterpret_cast<float *>(s_mem) + (threadIdx.x >> 6) * kStageFloatsPerWave;
    float4 *s_tff = s_mem + WAVES * kStageFloatsPerWave / 4;
    uint32_t *s_skip = reinterpret_cast<uint32_t *>(s_tff + tf.tff_n);
    for (uint32_t i = threadIdx.x; i < tf.tff_n; i += WAVES * 64) s_tff[i] = tf.tff[i];
    if (ESS && SKIP_LDS)
        for (uint32_t i = threadIdx.x; i <= skip.n_words; i += WAVES * 64) s_skip[i] = skip.bits[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = lane & (kSplit - 1);   // 16 rays per wave, 4 lanes each
    constexpr uint32_t kRaysPerWave = 64 / kSplit;
    const int tffn = (int)tf.tff_n;
    unsigned long long c_taken = 0, c_shaded = 0, c_bricks = 0, c_skipped = 0;
    const Vol<VT, INSTR, FP> vol = make_vol<VT, INSTR, FP>(vv, touched);
    const f3 resf = mk3(vol.fw, vol.fh, vol.fd);
    const f3 voxLen = mk3(1.f / vol.fw, 1.f / vol.fh, 1.f / vol.fd);
    const float refInterval = 1.f / rc.samplingRate;
    const Grid grid = make_grid(bricks, rc, skip.n_words, ESS);
    const uint32_t *sb = SKIP_LDS ? s_skip : skip.bits;
    const bool skip_empty = INSTR != 2 && cells.empty != nullptr && rp.useLinear != 0 &&
                            !(XS && rp.illumType == 4);
#ifdef VR_LEAP   // A/B build (VR_EXPERIMENTS + VR_LEAP_STEPPING); VRHIP_MARCH_MICRO = leap steps per round
    const bool use_mask = skip_empty && cells.bmask != nullptr && fr.march_micro != 0;
    const uint32_t leap_iters = fr.march_micro;
#endif
#ifdef VR_LEAP
    LeapCache lc;
    lc.key0 = lc.key1 = 0xffffffffu; lc.m0 = lc.m1 = 0ull; lc.du = lc.dv = lc.ds = lc.inv_step = 0.f;
#endif
    // Rays are handed out one by one from the sorted list: when `refill_min` ray slots (quads) of
    // the wave are idle they retire their rays and take the next ones (their set-up runs
    // together).  With the default, 16, a wave refills when all its rays are done, but draws as
    // many rays as it has slots from wherever the list head is -- finer than fixed groups of 16,
    // and with the longest rays first the tail is made of the shortest ones.  Smaller values keep
    // the waves fuller at the price of more (divergent) set-ups: measured better by 2-3 % on the
    // "shells" volumes, worse by as much on dense ones.  Exit condition reached by every wave: the
    // list head only grows, and every ray ends.
    const uint32_t refill_min = fr.refill_min ? fr.refill_min : kRaysPerWave;
    unsigned long long dummy0 = 0, dummy1 = 0;
    const bool count = INSTR && slot == 0;
    bool have = false, drained = false, first_draw = true;
    uint32_t gx = 0, gy = 0, out_index = 0, my_rounds = 0;
    bool guess_empty = true;   // identical in the four lanes of a ray, like all of its state
    uint32_t cool = 0;         // evaluation batches before the ray guesses "empty" again
    RayCtx c;
    RayDyn d;
    setup_ray<ESS>(0u, 0u, false, fr, cam, rp, rc, resf, voxLen, grid, c, d, rp.seed);   // S_DONE

    for (;;) {
        {
            const bool idle = d.state == S_DONE;
            const unsigned long long idle_m = __ballot(idle);
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle_m) / kSplit;
            const bool all_idle = idle_m == ~0ull;
            if ((!drained && n_idle >= refill_min) || all_idle) {
                VR_COUNT(11);
                // retire the finished rays
                if (idle && have) {
                    if (XS && rc.useAO && slot == 0 && d.ert)
                        apply_ao<VT, INSTR>(vol, s_tff, tffn, c, d, rp, gx, gy);
                    if (slot == 0) {
                        write_pixel<XS>(fr, rp, c, d, voxLen, gx, gy, (size_t)out_index);
                        if (fr.cost)
                            fr.cost[(size_t)gy * fr.W + gx] = (uint16_t)(my_rounds < 65535u ? my_rounds : 65535u);
                    }
                    have = false;
                }
                if (!drained) {
                    // (a wave's first 16 rays are its own by position, the later ones are drawn behind those: see
                    // vr_raycast_rays_kernel)
                    uint32_t base = 0;
                    if (first_draw) {
                        base = (blockIdx.x * (uint32_t)WAVES + (threadIdx.x >> 6)) * kRaysPerWave;
                    } else {
                        if (lane == 0) base = atomicAdd(fr.cont_head, n_idle);
                        base = __builtin_amdgcn_readfirstlane(base) + gridDim.x * (uint32_t)WAVES * kRaysPerWave;
                    }
                    first_draw = false;
                    if (base + n_idle >= n_rays) drained = true;
                    if (idle) {
                        // my quad's rank among the idle quads (every lane of a quad is idle or none is)
                        const uint32_t below = (uint32_t)__builtin_popcountll(
                            idle_m & ((1ull << (lane & ~(uint64_t)(kSplit - 1))) - 1ull)) / kSplit;
                        const uint32_t ri = base + below;
                        have = ri < n_rays;
                        if (have) {
                            const uint32_t rix = fr.order ? fr.order[ri] : ri;
                            const ContRec rec = fr.cont[rix];
                            gx = rec.pix & 0xffffu;
                            gy = rec.pix >> 16;
                            out_index = rec.out_index;
                            const uint32_t f = (uint32_t)rec.state >> 8;
                            setup_ray<ESS>(gx, gy, true, fr, cam, rp, rc, resf, voxLen, grid, c, d,
                                           fr.seeds ? fr.seeds[f] : rp.seed);
                            d.state = rec.state & 0xff;
                            d.t = rec.t; d.t_exit = rec.t_exit; d.alpha = rec.alpha;
                            d.r0 = rec.r0; d.r1 = rec.r1; d.r2 = rec.r2;
                            d.c0 = rec.cx; d.c1 = rec.cy; d.c2 = rec.cz;
                            d.tv0 = rec.tv0; d.tv1 = rec.tv1; d.tv2 = rec.tv2;
#ifdef VR_RAYLEN
                            d.nsmp = rec.pad;
#endif
                            if (ESS) fetch_skip_word(sb, grid, d);
                            my_rounds = 0;
                            guess_empty = true;
                            cool = 0;
#ifdef VR_LEAP
                            leap_reset(lc, c, vol.fw, vol.fh, vol.fd);
#endif
                        }
                    }
                }
                VR_STAMP(1);
                if (!__ballot(d.state != S_DONE)) {
                    if (drained) break;
                    continue;
                }
            }
        }
        {   // ---- one round
            if (ESS) {
                for (int it = 0;; ++it) {
                    if (!__ballot(d.state == S_BRICK)) break;
                    if (it >= kMaxBrickSteps && __ballot(d.state == S_SAMPLE)) break;
                    VR_COUNT(9);
#ifdef VR_DDA_TWICE   // diagnostic build: what does a phase-2 DDA step cost?
                    {
                        RayDyn d2 = d;
                        dda_step<0>(sb, grid, c, d2, dummy0, dummy1);
                        asm volatile("" ::"v"(d2.t), "v"(d2.state), "v"(d2.skw), "v"(d2.tv0), "v"(d2.tv1),
                                     "v"(d2.tv2), "v"(d2.c0), "v"(d2.c1), "v"(d2.c2), "v"(d2.t_exit));
                    }
#endif
                    if (count) dda_step<INSTR>(sb, grid, c, d, c_bricks, c_skipped);
                    else dda_step<0>(sb, grid, c, d, dummy0, dummy1);
                }
            }
            VR_STAMP(2);
            if (!__ballot(d.state != S_DONE)) continue;
            VR_COUNT(10);
            my_rounds += d.state != S_DONE ? 1u : 0u;
#ifdef VR_CAP_ROUNDS   // diagnostic build (wrong image): is phase 2 bound by its longest rays?
            if (my_rounds >= VR_CAP_ROUNDS) d.state = S_DONE;
#endif
            bool more_empty = false;
#ifdef VR_LEAP
            if (use_mask) {
                // (the four lanes of a ray hold the same state and cache and take the same steps)
                bool ready = false;
                for (uint32_t it = 0; it < leap_iters; ++it) {
                    const bool act = d.state == S_SAMPLE && !ready;
                    if (!__ballot(act)) break;
                    if (act) ready = leap_step<ESS>(cells, vol, grid, c, d, lc, count, c_taken);
                    if (ESS && __ballot(d.state == S_BRICK)) {
                        if (count) dda_step<INSTR>(sb, grid, c, d, c_bricks, c_skipped);
                        else dda_step<0>(sb, grid, c, d, dummy0, dummy1);
                    }
                }
                more_empty = !ready;
            } else
#endif
            if (skip_empty && lookahead_pays(d.state == S_SAMPLE, guess_empty)) {
                if (d.state == S_SAMPLE) {
                    // the four lanes of a ray hold the same state and take the same decisions;
                    // lane `slot` looks at samples [kLook2 * slot, kLook2 * (slot + 1)) of the run
                    // (its window start is approximate, which is all the cell lookup needs)
                    const float t_win = d.t + (float)(kLook2 * (int)slot) * c.stepSize;
#ifdef VR_TWICE_LOOK   // diagnostic build: sensitivity to the lookahead's cost
                    {
                        const int em2 = (int)empty_mask<VT, INSTR, kLook2>(cells, vol, c, t_win + 1e-7f);
                        asm volatile("" ::"v"(em2));
                    }
#endif
                    const int em = (int)empty_mask<VT, INSTR, kLook2>(cells, vol, c, t_win);
                    const uint32_t m4[4] = {(uint32_t)quad_bcast<0>(em), (uint32_t)quad_bcast<1>(em),
                                            (uint32_t)quad_bcast<2>(em), (uint32_t)quad_bcast<3>(em)};
                    more_empty = skip_empty_run_wide(m4, c, d, count, c_taken);
                    guess_empty = (m4[0] & 1u) != 0u;
                    // (see phase 1: no new guess while the mask shows samples in cells that are not empty)
                    if (!(m4[0] & 1u)) {
                        const uint32_t all = m4[0] | (m4[1] << kLook2) | (m4[2] << (2 * kLook2)) | (m4[3] << (3 * kLook2));
                        cool = (uint32_t)((all ? __builtin_ctz(all) : 4 * kLook2) / (kSplit * kBatch));
                    }
                    after_segment<ESS>(c, d);
                }
            }
            VR_STAMP(4);
            const bool ev2 = d.state == S_SAMPLE && !more_empty;   // (the same in the four lanes of a ray)
            if (INSTR == 0 && !XS) {
                // default kernels: the batch as wave-uniform code (phase 1's eval_front / eval_dense /
                // eval_back): the dense pass over the gathered samples is run by all 64 lanes, also those
                // of rays that step over empty runs or have ended
                if (__ballot(ev2)) {
                    float tk[kBatch] = {0.f, 0.f, 0.f, 0.f};
                    bool vk[kBatch] = {false, false, false, false};
                    float tc = d.t;
                    bool v = ev2 && d.t < d.t_exit;
#pragma unroll
                    for (int i = 0; i < kSplit * kBatch; ++i) {
                        if ((int)slot == i / kBatch) { tk[i % kBatch] = tc; vk[i % kBatch] = v; }
                        const float tn = tc + c.stepSize;
                        v = v && !(tc >= c.tfar) && (tn < d.t_exit);
                        tc = tn;
                    }
                    EvalFront ef;
                    const uint32_t ns = eval_front<VT, FP>(vol, s_tff, tffn, s_stage, c, rp, tk, vk, ev2, ef);
                    if (ns) eval_dense<VT, FP>(vol, s_stage, c, refInterval, ns);
                    float p0[kBatch], p1[kBatch], p2[kBatch], opk[kBatch];
                    eval_back(s_stage, c, rp, ef, ns, p0, p1, p2, opk);
                    VR_STAMP(3);
                    int fl[kBatch];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) fl[k] = (vk[k] ? 1 : 0) | ((ef.lit[k] && rp.illumType == 1) ? 2 : 0);
                    composite_from<0>(c, d, p0, p1, p2, opk, tk, fl, false, c_taken, c_shaded);
                    composite_from<1>(c, d, p0, p1, p2, opk, tk, fl, false, c_taken, c_shaded);
                    composite_from<2>(c, d, p0, p1, p2, opk, tk, fl, false, c_taken, c_shaded);
                    composite_from<3>(c, d, p0, p1, p2, opk, tk, fl, false, c_taken, c_shaded);
                    {   // the ray's 16th sample of this round: lane 3 of the quad, slot kBatch - 1
                        const int f3v = quad_bcast<3>(fl[kBatch - 1]);
                        const float o3 = quad_bcast<3>(opk[kBatch - 1]);
                        if (ev2) {
                            if (f3v & 1) guess_empty = o3 == 0.f && cool == 0u;
                            if (cool) --cool;
                            after_segment<ESS>(c, d);
                        }
                    }
                    VR_STAMP(6);
                }
            } else
            if (d.state == S_SAMPLE && !more_empty) {
                // parameters (t += stepSize, :879) and validity (:790, :868) of the ray's next 16
                // samples; this lane keeps numbers 4*slot .. 4*slot+3
                float tk[kBatch] = {0.f, 0.f, 0.f, 0.f};
                bool vk[kBatch] = {false, false, false, false};
                float tc = d.t;
                bool v = d.t < d.t_exit;
#pragma unroll
                for (int i = 0; i < kSplit * kBatch; ++i) {
                    if ((int)slot == i / kBatch) { tk[i % kBatch] = tc; vk[i % kBatch] = v; }
                    const float tn = tc + c.stepSize;
                    v = v && !(tc >= c.tfar) && (tn < d.t_exit);
                    tc = tn;
                }
                if (INSTR == 2) {   // no speculative voxel touches: one sample per round
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) vk[k] = vk[k] && slot == 0 && k == 0;
                }
                float p0[kBatch], p1[kBatch], p2[kBatch], opk[kBatch];
                bool litk[kBatch];
#ifdef VR_TWICE_EVAL   // diagnostic build: sensitivity of the frame time to the evaluation's cost
                {
                    float tk2[kBatch];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) tk2[k] = tk[k] + 1e-7f;
                    eval_batch<VT, INSTR, XS, FP>(vol, s_tff, tffn, s_stage, c, rp, rc, refInterval, tk2, vk, p0, p1,
                                          p2, opk, litk);
                    asm volatile("" ::"v"(p0[0]), "v"(p0[1]), "v"(p0[2]), "v"(p0[3]), "v"(p1[0]), "v"(p1[1]),
                                 "v"(p1[2]), "v"(p1[3]), "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]),
                                 "v"(opk[0]), "v"(opk[1]), "v"(opk[2]), "v"(opk[3]));
                }
#endif
                eval_batch<VT, INSTR, XS, FP>(vol, s_tff, tffn, s_stage, c, rp, rc, refInterval, tk, vk, p0, p1, p2,
                                      opk, litk);
                VR_STAMP(3);
                int fl[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k) fl[k] = (vk[k] ? 1 : 0) | (litk[k] ? 2 : 0);
#ifdef VR_TWICE_COMP   // diagnostic build: sensitivity to the compositing replay's cost
                {
                    RayDyn d2 = d;
                    unsigned long long z0 = 0, z1 = 0;
                    composite_from<0>(c, d2, p0, p1, p2, opk, tk, fl, false, z0, z1);
                    composite_from<1>(c, d2, p0, p1, p2, opk, tk, fl, false, z0, z1);
                    composite_from<2>(c, d2, p0, p1, p2, opk, tk, fl, false, z0, z1);
                    composite_from<3>(c, d2, p0, p1, p2, opk, tk, fl, false, z0, z1);
                    asm volatile("" ::"v"(d2.t), "v"(d2.state), "v"(d2.alpha), "v"(d2.r0), "v"(d2.r1), "v"(d2.r2));
                }
#endif
                composite_from<0>(c, d, p0, p1, p2, opk, tk, fl, count, c_taken, c_shaded);
                composite_from<1>(c, d, p0, p1, p2, opk, tk, fl, count, c_taken, c_shaded);
                composite_from<2>(c, d, p0, p1, p2, opk, tk, fl, count, c_taken, c_shaded);
                composite_from<3>(c, d, p0, p1, p2, opk, tk, fl, count, c_taken, c_shaded);
                {   // the ray's 16th sample of this round: lane 3 of the quad, slot kBatch - 1
                    const int f3v = quad_bcast<3>(fl[kBatch - 1]);
                    const float o3 = quad_bcast<3>(opk[kBatch - 1]);
                    if (f3v & 1) guess_empty = o3 == 0.f && cool == 0u;
                    if (cool) --cool;
                }
                after_segment<ESS>(c, d);
                VR_STAMP(6);
            }
        }
    }
    VR_STAMP_FLUSH_AT(16);

    if (INSTR) {
        const unsigned long long cc[6] = {c_taken, 0, c_shaded, c_bricks, c_skipped, 0};
        flush_counters(stats, lane, cc);
    }
}

// ------------------------------------------------------------------ phase-2 ordering

// Counting sort of the suspended rays by their key (ContRec::pad = rounds needed last frame,
// clamped to kSortBins - 1), longest first.  Two small launches between the phases.
__global__ __launch_bounds__(kBlockDim) void vr_cont_hist_kernel(const ContRec *cont,
                                                                 const uint32_t *count,
                                                                 uint32_t *bins)
{
    __shared__ uint32_t s_bins[kSortBins];
    for (uint32_t i = threadIdx.x; i < kSortBins; i += kBlockDim) s_bins[i] = 0;
    __syncthreads();
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * kBlockDim + threadIdx.x; i < n; i += gridDim.x * kBlockDim) {
        const uint32_t k = cont[i].pad < kSortBins ? cont[i].pad : kSortBins - 1u;
        atomicAdd(&s_bins[k], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kSortBins; i += kBlockDim)
        if (s_bins[i]) atomicAdd(&bins[i], s_bins[i]);
}

__global__ __launch_bounds__(kBlockDim) void vr_cont_scatter_kernel(const ContRec *cont,
                                                                    const uint32_t *count,
                                                                    const uint32_t *bins,
                                                                    uint32_t *cursors,
                                                                    uint32_t *order)
{
    __shared__ uint32_t s_off[kSortBins], s_cnt[kSortBins], s_base[kSortBins];
    // descending keys: bin k starts after all bins above it
    for (uint32_t k = threadIdx.x; k < kSortBins; k += kBlockDim) s_cnt[k] = bins[k];
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kSortBins; k += kBlockDim) {
        uint32_t o = 0;
        for (uint32_t j = k + 1; j < kSortBins; ++j) o += s_cnt[j];
        s_off[k] = o;
    }
    const uint32_t n = *count;
    const uint32_t chunk = kBlockDim * 4u;
    for (uint32_t c0 = blockIdx.x * chunk; c0 < n; c0 += gridDim.x * chunk) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < kSortBins; k += kBlockDim) s_cnt[k] = 0;
        __syncthreads();
        uint32_t key[4], rank[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = c0 + (uint32_t)j * kBlockDim + threadIdx.x;
            key[j] = kSortBins;
            rank[j] = 0;
            if (i < n) {
                key[j] = cont[i].pad < kSortBins ? cont[i].pad : kSortBins - 1u;
                rank[j] = atomicAdd(&s_cnt[key[j]], 1u);
            }
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < kSortBins; k += kBlockDim)
            s_base[k] = s_cnt[k] ? atomicAdd(&cursors[k], s_cnt[k]) : 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = c0 + (uint32_t)j * kBlockDim + threadIdx.x;
            if (i < n) order[s_off[key[j]] + s_base[key[j]] + rank[j]] = i;
        }
    }
}

// Image-order ESS, end of the frame (:918-924): the hit texel of every group of this launch.
// A skipped group and a group whose first work-item missed the box leave 0; otherwise the first
// work-item reports whether any work-item that reached the end changed its pixel.
__global__ __launch_bounds__(kBlockDim) void vr_hit_resolve_kernel(FrameView fr, uint8_t *hit_out)
{
    const uint32_t i = blockIdx.x * kBlockDim + threadIdx.x;
    if (i >= fr.n_wave_tiles) return;
    const WaveTile wt = fr.queue[i];
    const size_t g = (size_t)wt_row(wt) * fr.hit_w + wt_col(wt);
    hit_out[g] = fr.hit_status[g] == HIT_FIRST_ENDS ? fr.hit_any[g] : (uint8_t)0;
}

template <typename K>
hipError_t prepare_variant(K kernel, size_t lds, int *nb_out, const char *what, int num_cus, int block_dim = kBlockDim)
{
    return vr_prepare_kernel(kernel, block_dim, lds, nb_out, what, num_cus);
}

// dynamic LDS of the marching kernels: a stage per wave, the transfer function, the skip bitmap if it is kept there
inline size_t march_lds(int waves, const RaycastLaunch &a, bool skip_lds)
{
    return (size_t)waves * kStageFloatsPerWave * sizeof(float) + (size_t)a.tf.tff_n * sizeof(float4) +
           (skip_lds ? ((size_t)a.skip.n_words + 1) * sizeof(uint32_t) : 0);
}
// does a workgroup of kWavesWide waves with the skip bitmap fit a CU's LDS?
inline bool wide_fits_lds(const RaycastLaunch &a) { return march_lds(kWavesWide, a, true) <= (size_t)160 * 1024; }

#ifdef VR_EXPERIMENTS   // vr_march_kernel, vr_raycast_staged_kernel and their launchers: A/B builds only
#include "vr_experiments_kernels.inc"
#endif


template <typename VT, bool ESS, int INSTR, bool SKIP_LDS, bool XS, bool FP = false>
hipError_t launch_variant(const RaycastLaunch &a, hipStream_t stream)
{
    auto k1 = vr_raycast_kernel<VT, ESS, INSTR, SKIP_LDS, XS, FP>;
    auto k2 = vr_raycast_split_kernel<VT, ESS, INSTR, SKIP_LDS, XS, FP>;
    // three waves per SIMD (RaycastLaunch::occ3 / occ3_split): the default kernels on the footprint volume as ONE
    // workgroup of kWavesWide waves per CU (launch_typed keeps SKIP_LDS for them when the bitmap fits beside 12 stages)
    constexpr bool kWide = ESS && INSTR == 0 && !XS && FP;
    const bool wide2 = kWide && a.occ3_split;
    if (wide2) k2 = vr_raycast_split_kernel<VT, ESS, INSTR, SKIP_LDS, XS, FP, kWide ? kWavesWide : 4>;
    const int waves2 = wide2 ? kWavesWide : 4;
    const size_t lds = march_lds(4, a, ESS && SKIP_LDS);
    const size_t lds2 = march_lds(waves2, a, ESS && SKIP_LDS);
    int nb1 = 0, nb2 = 0;
    {
        hipError_t e = prepare_variant(k1, lds, &nb1, "raycast phase 1", a.num_cus);
        if (e == hipSuccess) e = prepare_variant(k2, lds2, &nb2, "raycast phase 2", a.num_cus, waves2 * 64);
        if (e != hipSuccess) return e;
    }
    const uint32_t cus = (uint32_t)(a.num_cus > 0 ? a.num_cus : 256);
    uint32_t want = (a.frame.n_wave_tiles + 3u) / 4u;
    uint32_t cap = cus * (uint32_t)nb1;
    dim3 grid(want < cap ? want : cap), block(kBlockDim);
    if (grid.x == 0) return hipSuccess;
    FrameView frame = a.frame;
    if (XS || INSTR != 0 || !ESS) frame.live_rays = nullptr;   // the ray list serves the default kernels
    if (a.info) {   // what this call launches, for vrhip_last_launch_info (completed below)
        vrhip_launch_info &li = *a.info;
        li.technique = 0;
        li.work_items = a.frame.n_wave_tiles;
        li.round_budget = a.frame.round_budget;
        li.footprint = FP ? 1u : 0u;
        li.instrumented = (uint32_t)INSTR;
        li.extras = XS ? 1u : 0u;
        li.skip_in_lds = (ESS && SKIP_LDS) ? 1u : 0u;
        li.phase1_waves = 4;
        li.phase2_waves = a.frame.round_budget ? (uint32_t)waves2 : 0u;
        li.sorted_phase2 = (a.frame.round_budget && a.frame.order) ? 1u : 0u;
        // the lookahead's condition in the kernels: cells.empty && useLinear (and INSTR != 2, not the XS modes that track every sample)
        li.empty_skip = (INSTR != 2 && a.cells.empty != nullptr && a.render.useLinear != 0 &&
                         !(XS && (a.render.illumType == 4 || a.render.showEss))) ? 1u : 0u;
    }
    // the events of the frame's timing ride on the launches themselves (RaycastLaunch::stop_event, start_event)
    hipEvent_t start_ev = (a.bind_events && a.start_bound) ? a.start_event : nullptr;
    if (ESS && INSTR == 0 && frame.live) {
        // what make_grid(bricks, rc, n_words, true) and 1 / resolution give on the device
        Grid hg;
        hg.bw = a.bricks.bw; hg.bh = a.bricks.bh; hg.bd = a.bricks.bd;
        hg.oob_word = a.skip.n_words;
        hg.bl0 = 1.f / a.raycast.brickRes[0];
        hg.bl1 = 1.f / a.raycast.brickRes[1];
        hg.bl2 = 1.f / a.raycast.brickRes[2];
        hg.brickDia = sqrtf(((hg.bl0 * hg.bl0) + (hg.bl1 * hg.bl1)) + (hg.bl2 * hg.bl2)) * 2.f;
        f3 hv;
        hv.x = 1.f / a.vol.fw; hv.y = 1.f / a.vol.fh; hv.z = 1.f / a.vol.fd;
        vr_launch_kernel(vr_dda_prepass_kernel<VT>, dim3(want), block, 0, stream, start_ev, nullptr, a.vol, a.bricks,
                         a.skip, frame, a.cam, a.render, a.raycast, hg, hv);
        hipError_t pe = hipGetLastError();
        if (pe != hipSuccess) return pe;
        if (start_ev) { *a.start_bound = true; start_ev = nullptr; }
        if (a.info) { a.info->prepass = 1; a.info->patch_classes = frame.patch_class ? 1u : 0u; }
    } else {
        frame.live = nullptr;
    }
    hipError_t e;
    const bool resolve_follows = a.render.imgEss && a.hit_out && a.frame.n_wave_tiles;
    const bool bind_stop = a.bind_events && a.stop_event && a.stop_bound && !resolve_follows;
    const bool p1_last = a.frame.round_budget == 0;
    const hipEvent_t p1_ev = !a.bind_events ? nullptr : (p1_last && bind_stop) ? a.stop_event : a.mid_event;
#ifdef VR_EXPERIMENTS
    if (ESS && INSTR == 0 && !XS && frame.live && frame.live_rays && frame.march && !a.raycast.contours &&
        !a.raycast.aerial) {
        if (start_ev && hipEventRecord(start_ev, stream) == hipSuccess) *a.start_bound = true;
        return launch_march<VT, SKIP_LDS, FP>(a, frame, block, cus, stream);   // the whole frame in one launch
    }
#endif
    if (ESS && INSTR == 0 && !XS && frame.live && frame.live_rays) {   // phase 1 on the ray list
        // phase 1 picks its own schedule: two or three waves per SIMD (three: footprint volume only), the skip
        // bitmap in LDS whenever it fits
        constexpr bool kWideR = FP;
        const bool r3 = kWideR && a.occ3;
        const bool rlds = a.skip.in_lds != 0 && (!r3 || wide_fits_lds(a));
        const int waves_r = r3 ? kWavesWide : 4;
        auto kr = vr_raycast_rays_kernel<VT, false, FP>;
        if (r3) {
            kr = vr_raycast_rays_kernel<VT, false, FP, kWideR ? kWavesWide : 4>;
            if (rlds) kr = vr_raycast_rays_kernel<VT, true, FP, kWideR ? kWavesWide : 4>;
        } else if (rlds) {
            kr = vr_raycast_rays_kernel<VT, true, FP>;
        }
        const size_t lds_r = march_lds(waves_r, a, rlds);
        int nbr = 0;
        e = prepare_variant(kr, lds_r, &nbr, "raycast phase 1 (ray list)", a.num_cus, waves_r * 64);
        if (e != hipSuccess) return e;
        vr_launch_kernel(kr, dim3(cus * (uint32_t)nbr), dim3(waves_r * 64), lds_r, stream, start_ev, p1_ev, a.vol, a.bricks,
                         a.tf, a.skip, a.cells, frame, a.cam, a.render, a.raycast);
        if (a.info) { a.info->ray_list = 1; a.info->phase1_waves = (uint32_t)waves_r; a.info->skip_in_lds = rlds ? 1u : 0u; }
    } else {
        vr_launch_kernel(k1, grid, block, lds, stream, start_ev, p1_ev, a.vol, a.bricks, a.tf, a.skip, a.cells, frame, a.cam,
                         a.render, a.raycast, a.stats, a.touched);
    }
    e = hipGetLastError();
    if (e == hipSuccess && start_ev) *a.start_bound = true;
    if (e == hipSuccess && p1_ev && p1_ev == a.stop_event) *a.stop_bound = true;
    if (e == hipSuccess && a.mid_event && p1_ev != a.mid_event) e = hipEventRecord(a.mid_event, stream);
    if (e != hipSuccess || p1_last) return e;
    if (a.frame.order) {   // longest rays first (keys: last frame's phase-2 rounds per pixel)
        hipLaunchKernelGGL(vr_cont_hist_kernel, dim3(128), block, 0, stream, a.frame.cont,
                           a.frame.cont_count, a.frame.sort_ws);
        hipLaunchKernelGGL(vr_cont_scatter_kernel, dim3(128), block, 0, stream, a.frame.cont,
                           a.frame.cont_count, a.frame.sort_ws, a.frame.sort_ws + kSortBins,
                           a.frame.order);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // phase 2: persistent grid; exits at once when nothing was suspended
    dim3 grid2(cus * (uint32_t)nb2);
    vr_launch_kernel(k2, grid2, dim3(waves2 * 64), lds2, stream, nullptr, bind_stop ? a.stop_event : nullptr, a.vol, a.bricks,
                     a.tf, a.skip, a.cells, frame, a.cam, a.render, a.raycast, a.stats, a.touched);
    e = hipGetLastError();
    if (e == hipSuccess && bind_stop) *a.stop_bound = true;
    return e;
}

template <typename VT>
hipError_t launch_typed(const RaycastLaunch &a, hipStream_t stream)
{
    // (phase 2 and the patch kernels; phase 1 on the ray list: launch_variant)
    const bool lds = a.skip.in_lds != 0 && (!a.occ3_split || wide_fits_lds(a));
    // the rarely used shading modes 2-5, contours, the depth cue and nearest filtering live in kernel
    // variants of their own (XS), so that their code and registers do not tax the default ones
    const bool xs = a.render.illumType >= 2 || a.raycast.useAO != 0 || a.render.showEss != 0 ||
                    a.render.imgEss != 0 || a.vol.channels > 1 || a.raycast.contours != 0 ||
                    a.raycast.aerial != 0 || a.render.useLinear == 0;
#ifdef VR_EXPERIMENTS
    if (a.frame.lds_stage && !xs && a.instr == 0 && a.use_ess && sizeof(VT) == 1 && !a.raycast.contours &&
        !a.raycast.aerial && a.frame.n_wave_tiles) {
        if (a.bind_events && a.start_bound && hipEventRecord(a.start_event, stream) == hipSuccess) *a.start_bound = true;
        return launch_staged(a, stream);
    }
#endif
    // the default kernels read the footprint volume when the host has provided one for this frame
    if (!xs && a.instr == 0 && a.vol.fp) {
        if (!a.use_ess) return launch_variant<VT, false, 0, false, false, true>(a, stream);
        return lds ? launch_variant<VT, true, 0, true, false, true>(a, stream)
                   : launch_variant<VT, true, 0, false, false, true>(a, stream);
    }
    if (a.use_ess) {
        if (lds) {
            if (a.instr == 0) return xs ? launch_variant<VT, true, 0, true, true>(a, stream) : launch_variant<VT, true, 0, true, false>(a, stream);
            if (a.instr == 1) return xs ? launch_variant<VT, true, 1, true, true>(a, stream) : launch_variant<VT, true, 1, true, false>(a, stream);
            return xs ? launch_variant<VT, true, 2, true, true>(a, stream) : launch_variant<VT, true, 2, true, false>(a, stream);
        }
        if (a.instr == 0) return xs ? launch_variant<VT, true, 0, false, true>(a, stream) : launch_variant<VT, true, 0, false, false>(a, stream);
        if (a.instr == 1) return xs ? launch_variant<VT, true, 1, false, true>(a, stream) : launch_variant<VT, true, 1, false, false>(a, stream);
        return xs ? launch_variant<VT, true, 2, false, true>(a, stream) : launch_variant<VT, true, 2, false, false>(a, stream);
    }
    if (a.instr == 0) return xs ? launch_variant<VT, false, 0, false, true>(a, stream) : launch_variant<VT, false, 0, false, false>(a, stream);
    if (a.instr == 1) return xs ? launch_variant<VT, false, 1, false, true>(a, stream) : launch_variant<VT, false, 1, false, false>(a, stream);
    return xs ? launch_variant<VT, false, 2, false, true>(a, stream) : launch_variant<VT, false, 2, false, false>(a, stream);
}

} // namespace

#ifdef VR_MARCH_STATS
// diagnostic builds only: read (and optionally clear) the march kernel's round statistics
extern "C" int vrhip_debug_march_stats(unsigned long long out[32], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_stats), 32 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_march_stats), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef VR_STAMPS
// diagnostic builds only: start/end clock of every wave of the last launches
extern "C" int vrhip_debug_wave_spans(unsigned long long *out /* [2][2][8192] */)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_span), sizeof(unsigned long long) * 2 * 2 * 8192) != hipSuccess)
        return -1;
    unsigned long long *z = (unsigned long long *)calloc(2 * 2 * 8192, sizeof(unsigned long long));
    if (!z) return -1;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_wave_span), z, sizeof(unsigned long long) * 2 * 2 * 8192);
    free(z);
    return e == hipSuccess ? 0 : -1;
}
// diagnostic builds only: read (and optionally clear) the per-phase cycle totals
extern "C" int vrhip_debug_stamps(unsigned long long out[32], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 32 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

hipError_t vr_launch_raycast(const RaycastLaunch &a, hipStream_t stream)
{
    hipError_t e;
    switch (a.format) {
    case VRHIP_UCHAR: e = launch_typed<uint8_t>(a, stream); break;
    case VRHIP_USHORT: e = launch_typed<uint16_t>(a, stream); break;
    case VRHIP_FLOAT: e = launch_typed<float>(a, stream); break;
    default: return hipErrorInvalidValue;
    }
    if (e == hipSuccess && a.render.imgEss && a.hit_out && a.frame.n_wave_tiles) {
        const bool bind_stop = a.bind_events && a.stop_event && a.stop_bound;
        vr_launch_kernel(vr_hit_resolve_kernel, dim3((a.frame.n_wave_tiles + kBlockDim - 1) / kBlockDim),
                         dim3(kBlockDim), 0, stream, nullptr, bind_stop ? a.stop_event : nullptr, a.frame, a.hit_out);
        e = hipGetLastError();
        if (e == hipSuccess && bind_stop) *a.stop_bound = true;
    }
    return e;
}

namespace {

// SkipView::near_bits: bytes "brick is not skipped", dilated by `radius` bricks along one axis per
// launch, then packed to bits (same bit order as the skip bitmap)
__global__ __launch_bounds__(kBlockDim) void vr_skip_live_kernel(const uint32_t *bits, size_t n, uint8_t *live)
{
    const size_t i = (size_t)blockIdx.x * kBlockDim + threadIdx.x;
    if (i < n) live[i] = ((bits[i >> 5] >> (i & 31u)) & 1u) ? 0 : 1;
}
__global__ __launch_bounds__(kBlockDim) void vr_skip_dilate_kernel(const uint8_t *in, uint8_t *out, int bw, int bh,
                                                                   int bd, int axis, int radius)
{
    const size_t n = (size_t)bw * bh * bd;
    const size_t i = (size_t)blockIdx.x * kBlockDim + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % (size_t)bw), y = (int)((i / (size_t)bw) % (size_t)bh), z = (int)(i / ((size_t)bw * bh));
    const int pos = axis == 0 ? x : axis == 1 ? y : z, len = axis == 0 ? bw : axis == 1 ? bh : bd;
    const size_t stride = axis == 0 ? 1 : axis == 1 ? (size_t)bw : (size_t)bw * bh;
    uint8_t v = 0;
    for (int k = max(pos - radius, 0); k <= min(pos + radius, len - 1); ++k) v |= in[i + (size_t)(k - pos) * stride];
    out[i] = v;
}
__global__ __launch_bounds__(kBlockDim) void vr_skip_pack_kernel(const uint8_t *live, size_t n, uint32_t *bits,
                                                                 uint32_t n_words)
{
    const size_t i = (size_t)blockIdx.x * kBlockDim + threadIdx.x;
    const unsigned long long m = __ballot(i < n && live[i] != 0);
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (i >> 6) * 2;
        if (w < n_words) bits[w] = (uint32_t)m;
        if (w + 1 < n_words) bits[w + 1] = (uint32_t)(m >> 32);
    }
}

} // namespace

hipError_t vr_launch_skip_near(const BrickView &bricks, const uint32_t *bits, uint32_t n_words, uint32_t radius,
                               uint8_t *scratch, uint32_t *near_bits, hipStream_t stream)
{
    const size_t n = (size_t)bricks.bw * bricks.bh * bricks.bd;
    dim3 grid((unsigned)((n + kBlockDim - 1) / kBlockDim)), block(kBlockDim);
    uint8_t *a = scratch, *b = scratch + n;
    hipLaunchKernelGGL(vr_skip_live_kernel, grid, block, 0, stream, bits, n, a);
    for (int axis = 0; axis < 3; ++axis) {
        hipLaunchKernelGGL(vr_skip_dilate_kernel, grid, block, 0, stream, (const uint8_t *)a, b, bricks.bw, bricks.bh,
                           bricks.bd, axis, (int)radius);
        uint8_t *t = a; a = b; b = t;
    }
    hipLaunchKernelGGL(vr_skip_pack_kernel, grid, block, 0, stream, (const uint8_t *)a, n, near_bits, n_words);
    return hipGetLastError();
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

hipError_t vr_launch_patch_classes(const RaycastLaunch &a, uint32_t n_patches, uint32_t set_frames, uint8_t *cls,
                                   hipStream_t stream)
{
    if (!n_patches) return hipSuccess;
    Grid hg;   // (as in launch_variant: what make_grid gives on the device)
    hg.bw = a.bricks.bw; hg.bh = a.bricks.bh; hg.bd = a.bricks.bd;
    hg.oob_word = a.skip.n_words;
    hg.bl0 = 1.f / a.raycast.brickRes[0];
    hg.bl1 = 1.f / a.raycast.brickRes[1];
    hg.bl2 = 1.f / a.raycast.brickRes[2];
    hg.brickDia = sqrtf(((hg.bl0 * hg.bl0) + (hg.bl1 * hg.bl1)) + (hg.bl2 * hg.bl2)) * 2.f;
    hipLaunchKernelGGL(vr_patch_class_kernel, dim3((n_patches + 3u) / 4u), dim3(kBlockDim), 0, stream, a.skip, a.frame,
                       a.cam, a.render, hg, n_patches, set_frames, cls);
    return hipGetLastError();
}

// 1 when this library was built with the opt-in experiment kernels (VRHIP_MARCH, VRHIP_LDS_STAGE, VRHIP_MARCH_MICRO)
int vr_experiments_built()
{
#ifdef VR_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
