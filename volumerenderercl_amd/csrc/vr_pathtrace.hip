// vr_pathtrace.hip -- the Woodcock-tracking path-tracing mode of the reference kernel
// (/root/reference/src/kernel/volumeraycast.cl:686-706 with trace_volume :463-503,
// sample_interaction :407-437, get_dir_phase_function :453-460, gradientCentralDiffTff
// :181-206) for gfx950.
//
// One sample per pixel and launch; the caller accumulates over `iteration` (running mean in
// the fp32 frame buffer, as the ray caster does).  A pixel is up to three tracking walks --
// primary, optional scatter, shadow -- each a chain of up to 513 steps with a PER-CALL CONSTANT
// stride and acceptance threshold (`rand2 = ParallelRNG(rand)` is loop-invariant, :423-424;
// SURVEY A.8), so a walk is a strided march whose length differs wildly between neighbouring
// pixels (exponentially distributed stride, uniformly distributed threshold).
//
// Execution design: persistent workgroups, transfer function in LDS, and LANE-LEVEL work
// refill: every lane runs a small state machine (fetch pixel -> primary walk -> shade ->
// scatter walk -> shadow walk -> write) and draws its next pixel as soon as its current one is
// written, instead of waiting for the longest walk of an 8x8 patch.  Pixels are handed out in
// blocks of 64 from the same centre-first queue of 8x8 patches the ray caster uses, so a wave
// still works on neighbouring pixels most of the time.  Each round evaluates kPtBatch
// consecutive tracking steps of the lane's walk as independent straight-line code (addresses
// are clamped, so steps past the end of the walk are harmless speculation) and then resolves
// the walk's sequential exit conditions in order: the per-pixel operation sequence is the
// reference's.
#include "vr_sampling.h"

namespace {

// Diagnostic build only (-DVR_STAMPS, tools/pt_stamps.py): where a wave of the path tracer spends its time -- summed shader
// clock per stage (a stamp waits for the memory queue, so a stage owns the latency of what it issued), calls, lifetime.
#ifdef VR_STAMPS
__device__ unsigned long long g_pt_stamps[16];
#define PT_STAMP(i) do { const unsigned long long n_ = vr_stamp(); pt_acc[i] += n_ - pt_last; pt_last = n_; } while (0)
#define PT_COUNT(i) pt_acc[i] += 1
#else
#define PT_STAMP(i)
#define PT_COUNT(i)
#endif

#ifndef VR_PT_BATCH
#define VR_PT_BATCH 6
#endif
constexpr int kPtBatch = VR_PT_BATCH;

enum : int { P_FETCH = 0, P_PRIMARY = 1, P_SCATTER = 2, P_SHADOW = 3, P_WRITE = 4, P_ENDED = 8 };
#ifndef VR_PT_STAGE_MIN
#define VR_PT_STAGE_MIN 16
#endif
// lanes that must wait for the (divergent) refill / shading code before it is worth running
constexpr int kStageMin = VR_PT_STAGE_MIN;
#ifndef VR_LEAP_MARGIN
#define VR_LEAP_MARGIN 1.f
#endif
#ifndef VR_LEAP_PIECES
#define VR_LEAP_PIECES 1
#endif
constexpr int kLeapPieces = VR_LEAP_PIECES;   // closed-form stretches of a leap (one binade each)
#ifndef VR_LEAP_CHAIN
#define VR_LEAP_CHAIN 1
#endif
constexpr int kLeapChain = VR_LEAP_CHAIN;     // leaps of a walk per round, each from the landing of the one before
#ifndef VR_PT_SHADE_MIN
#define VR_PT_SHADE_MIN VR_PT_STAGE_MIN
#endif
constexpr int kShadeMin = VR_PT_SHADE_MIN;   // the same for stage 3 (walk ends: shading, next walk, pixel write)

VR_DEV bool in_volume(f3 p)   // volumeraycast.cl:93-96
{
    return vmax(fabsf(p.x), vmax(fabsf(p.y), fabsf(p.z))) < 1.f;
}

// get_dir_phase_function, volumeraycast.cl:453-460
VR_DEV f3 dir_phase_function(uint32_t rnd)
{
    const uint32_t rand2 = parallel_rng(rnd);
    const float phi = (float)(2.0 * (double)3.14159274101257f) * map_uint_float(rand2);
    const float cos_theta = 1.0f - 2.0f * map_uint_float(parallel_rng(rand2));
    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float s, c;
    vr_sincosf(phi, &s, &c);
    return mk3(c * sin_theta, s * sin_theta, cos_theta);
}

// illumination (:294-303) with specularBlinnPhong (:280-291)
VR_DEV f3 illumination(f3 color, f3 toLightDir, f3 n)
{
    const f3 l = normalize3(toLightDir);
    const float ndl = vmax(0.f, dot3(n, l));
    f3 h = add3(toLightDir, l);
    float sp = 0.0f;
    if (!(dot3(h, h) < 1.e-6f)) {
        h = normalize3(h);
        sp = vr_powr(vmax(dot3(n, h), 0.f), 40.f);
    }
    sp = sp * 0.15f;
    return mk3(((color.x * 0.15f) + ((color.x * ndl) * 0.7f)) + sp,
               ((color.y * 0.15f) + ((color.y * ndl) * 0.7f)) + sp,
               ((color.z * 0.15f) + ((color.z * ndl) * 0.7f)) + sp);
}

struct PtPixel {   // the pixel a lane is working on
    uint32_t gx, gy, out_index;
    uint32_t rnd;           // ParallelRNG3(x, y, seed), :688
    float dt, thr;          // stride (log(1-u)/max_extinction) and acceptance threshold of every walk
    float env0, env1, env2, env3;
    f3 dir;                 // primary ray direction
    f3 hit_pos;             // position of the primary interaction
    float c0, c1, c2;       // colour carried through trace_volume
    // current walk
    f3 org, wdir;
    float t;
    uint32_t cnt;
    // how it ended
    bool accepted;
    f3 apos;
    float adens;
};

#ifdef VR_PT_WAVES_PER_EU   // A/B builds: more waves per SIMD at fewer registers
#define VR_PT_OCC __attribute__((amdgpu_waves_per_eu(VR_PT_WAVES_PER_EU, VR_PT_WAVES_PER_EU)))
#else
#define VR_PT_OCC
#endif
template <typename VT, int INSTR>
__global__ __launch_bounds__(kBlockDim) VR_PT_OCC void vr_pathtrace_kernel(
    VolView vv, TfView tf, CellView grid, FrameView fr, vrhip_camera_params cam,
    vrhip_rendering_params rp, vrhip_pathtrace_params pt, DevStats *stats, uint32_t *touched)
{
    VR_ZERO_NEXT_CTRL(fr);
    extern __shared__ float4 s_mem[];
    float4 *s_tff = s_mem;
    for (uint32_t i = threadIdx.x; i < tf.tff_n; i += kBlockDim) s_tff[i] = tf.tff[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const int tffn = (int)tf.tff_n;
    unsigned long long c_taken = 0, c_hit = 0, c_culled = 0, c_leaped = 0;
    // The instrumented traffic variant (INSTR 2) reproduces the reference's fetch set: no culling.
    // INSTR 3 records the micro-bricks of the fetches the culling lets through -- what the production
    // kernel needs from the volume (vrhip_count_fetched).
    const bool cull = INSTR != 2 && grid.bound != nullptr;
    constexpr int VI = INSTR == 3 ? 2 : INSTR;   // the sampler's instrumentation level
    Vol<VT, VI> vol;
    vol.p = (const VT *)vv.data;
    vol.w1 = vv.w - 1; vol.h1 = vv.h - 1; vol.d1 = vv.d - 1;
    vol.fw = vv.fw; vol.fh = vv.fh; vol.fd = vv.fd;
    vol.inv_max = vv.inv_max;
    vol.nbx = vv.nbx; vol.nby = vv.nby;
    vol.ystride = vv.ystride; vol.zstride = (uint32_t)vv.zstride;
    vol.touched = touched;
    // the traffic-instrumented variant must not touch speculative voxels: one step per round
    constexpr int B = INSTR >= 2 ? 1 : kPtBatch;
    const bool leap = cull && grid.cbound != nullptr;   // leaps over macro cells (stage 2)

    PtPixel px = {};
    int state = P_FETCH;
    uint32_t patch_taken = 64;         // pixels of the current patch already handed out (wave-uniform)
    bool drained = false;              // queue exhausted (wave-uniform)
    bool first_patch = true;
    uint32_t sub = (blockIdx.x * (kBlockDim / 64u) + (threadIdx.x >> 6)) % kDrawCounters, sub_tried = 0;   // (stage 1's counters)

    WaveTile wt = {0, 0, 0};

    // Exit condition reached by every wave: the queue head only grows, every walk ends after at
    // most 513 steps, and the refill / transition stages run unconditionally once no lane walks.
    // The lanes by state, as wave masks kept across the rounds (a round in which nobody is handed a pixel or shaded --
    // most rounds -- recomputes only the two that stage 2 changes): idle (P_FETCH), walking, walk ended (P_ENDED).
    unsigned long long idle_m = ~0ull, walk_m = 0ull;
#ifdef VR_STAMPS
    unsigned long long pt_acc[16] = {0}, pt_last = vr_stamp(), pt_first = pt_last, pt_drained = 0;
#endif
    for (;;) {
        PT_STAMP(6);   // loop control
        // ---- stage 1: hand out pixels to idle lanes -- when enough lanes are idle to pay for the
        //      ray set-up code, or when nothing else is left to do
        if (!drained && idle_m && ((int)__builtin_popcountll(idle_m) >= kStageMin || !walk_m)) {
            unsigned long long idle = idle_m;
            while (idle && !drained) {
                if (patch_taken >= 64) {
                    // (a wave's first patch is its own by position, the later ones are drawn behind those: no queue
                    // of 3 072 waves at one counter when the kernel starts -- see vr_raycast_rays_kernel)
                    uint32_t q = 0;
                    if (first_patch) {
                        q = blockIdx.x * (kBlockDim / 64u) + (threadIdx.x >> 6);
                    } else {
                        // -- through one of kDrawCounters counters (a cache line each): counter k hands out the patches
                        // G + 8 j + k, a wave starts at k = its number mod 8 and moves on to the next counter when one has
                        // run out.  (One counter for 3 072 waves: +7 % on the sphere, +12 % on the shells.)
                        const uint32_t G = gridDim.x * (kBlockDim / 64u);
                        for (;;) {
                            uint32_t j = 0;
                            if (lane == 0) j = atomicAdd(fr.draw_count + sub * kLiveStride, 1u);
                            j = __builtin_amdgcn_readfirstlane(j);
                            q = G + j * kDrawCounters + sub;
                            if (q < fr.n_wave_tiles || ++sub_tried >= kDrawCounters) break;
                            sub = (sub + 1u) % kDrawCounters;
                        }
                    }
                    first_patch = false;
                    if (q >= fr.n_wave_tiles) { drained = true; break; }
                    patch_taken = 0;
                    wt = fr.queue[q];
                }
                // the i-th idle lane takes pixel patch_taken + i of the patch
                const uint32_t rank = (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
                const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
                const uint32_t avail = 64u - patch_taken;
                const uint32_t n_take = n_idle < avail ? n_idle : avail;
                if (state == P_FETCH && rank < n_take) {
                    const uint32_t pi = patch_taken + rank;
                    const uint32_t lx = pi & 7u, ly = pi >> 3;
                    px.gx = wt_col(wt) * 8u + lx;
                    px.gy = wt_row(wt) * 8u + ly;
                    px.out_index = wt.out_base + ly * fr.out_stride + lx;
                    // pixels outside the frame (ragged right / bottom patches) are dropped
                    if (px.gx < fr.W && px.gy < fr.H) {
                        const Ray ray = make_ray(px.gx, px.gy, fr, cam, rp, rp.seed);
                        px.env0 = ray.env[0]; px.env1 = ray.env[1];
                        px.env2 = ray.env[2]; px.env3 = ray.env[3];
                        if (!ray.hit) {   // :677-683
                            const float4 o = make_float4(px.env0, px.env1, px.env2, px.env3);
                            fr.fb[(size_t)px.gy * fr.W + px.gx] = o;
                            if (fr.out) fr.out[px.out_index] = o;
                        } else {
                            if (INSTR) c_hit++;
                            px.rnd = parallel_rng3(px.gx, px.gy, rp.seed);   // :688
                            const uint32_t rand2 = parallel_rng(px.rnd);     // :423
                            px.dt = vr_logf(1.f - map_uint_float(rand2)) / pt.max_extinction;
                            px.thr = map_uint_float(px.rnd);
                            px.dir = ray.dir;
                            px.c0 = px.env0; px.c1 = px.env1; px.c2 = px.env2;
                            px.org = add3(ray.cam, scale3(ray.dir, ray.tnear));   // :473
                            px.wdir = ray.dir;
                            px.t = 0.f;
                            px.cnt = 0;
                            state = P_PRIMARY;
                        }
                    }
                }
                patch_taken += n_take;
                idle = __ballot(state == P_FETCH);
            }
            idle_m = idle;
            walk_m = __ballot(state >= P_PRIMARY && state <= P_SHADOW);
            PT_STAMP(0);
            PT_COUNT(8);
#ifdef VR_STAMPS
            if (drained && !pt_drained) pt_drained = pt_last;
#endif
        }

        // ---- stage 2: kPtBatch consecutive tracking steps of every walking lane (:419-431)
        const bool walking = state >= P_PRIMARY && state <= P_SHADOW;
        if (walk_m) {
            float tk[B], dens[B], al[B];
            f3 pk[B];
            bool ink[B], need[B];
            float tc = px.t;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                tc = tc - px.dt;
                tk[k] = tc;
                pk[k] = add3(px.org, scale3(px.wdir, tc));
                ink[k] = in_volume(pk[k]);
                need[k] = walking && ink[k];
            }
            // majorant cull: no value a fetch in this cell can return maps to an opacity that
            // reaches the walk's threshold -> the step is a rejection whatever the voxels hold
            float lax = 0.f, lbx = 0.f, lay = 0.f, lby = 0.f, laz = 0.f, lbz = 0.f, lcb = 0.f;   // (for the leap below)
            uint32_t lcx = 0, lcy = 0, lcz = 0, lrad = 0, llev = 0xffffffffu;
            if (cull) {
                // cell of step k from the walk's voxel-space line u'(t) = a + b * t, in cells (one fma,
                // one conversion and one clamp per axis).  u' = p * res: the fetch's low-corner texel is
                // x0 = floor(u' - 0.5), so x' = floor(u') is x0 or x0 + 1 (the line's rounding is far
                // below half a texel), and the voxels x0, x0 + 1 lie in [x' - 1, x' + 1] -- inside the
                // extent [E c - 1, E c + E + 1] the cell of x' answers for.  Steps outside the volume
                // (never fetched: `ink`) clamp to a cell inside.
                const float inv_e = __uint_as_float((uint32_t)(127 - grid.shift) << 23);   // 2^-shift
                const float hx = 0.5f * vol.fw * inv_e, hy = 0.5f * vol.fh * inv_e, hz = 0.5f * vol.fd * inv_e;
                const float ax = __builtin_fmaf(px.org.x, hx, hx), bx = px.wdir.x * hx;
                const float ay = __builtin_fmaf(px.org.y, hy, hy), by = px.wdir.y * hy;
                const float az = __builtin_fmaf(px.org.z, hz, hz), bz = px.wdir.z * hz;
                // (the clamp in the float domain -- one v_med3_f32, where the integer clamp is a max and a min; the
                // conversion truncates towards zero, so the cell is the same: a coordinate in (-1, 0) becomes 0 either way)
                const float mx = (float)(grid.cx - 1), my = (float)(grid.cy - 1), mz = (float)(grid.cz - 1);
                float bnd[B];
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const uint32_t x = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(bx, tk[k], ax), 0.f, mx);
                    const uint32_t y = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(by, tk[k], ay), 0.f, my);
                    const uint32_t z = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(bz, tk[k], az), 0.f, mz);
                    bnd[k] = grid.bound[(z * (uint32_t)grid.cy + y) * (uint32_t)grid.cx + x];
                    if (k == B - 1 && leap) {   // the macro cell of the batch's last step, and its bound
                        lcx = x >> kLeapShift; lcy = y >> kLeapShift; lcz = z >> kLeapShift;
                        const uint32_t ci = (lcz * (uint32_t)grid.ccy + lcy) * (uint32_t)grid.ccx + lcx;
                        lcb = grid.cbound[ci];
                        if (grid.cdist) {   // how far the macro cells around are free at the level below the walk's threshold
                            const uint32_t j = (uint32_t)(px.thr * 8.f);   // tau_j = j / 8 <= thr (exact: a power of two)
                            const uint32_t jj = j < (uint32_t)kLeapLevels ? j : (uint32_t)kLeapLevels;
                            if (jj) {
                                llev = (jj - 1u) * (uint32_t)(grid.ccx * grid.ccy * grid.ccz);
                                lrad = grid.cdist[llev + ci];
                            }
                        }
                    }
                }
                lax = ax; lbx = bx; lay = ay; lby = by; laz = az; lbz = bz;
#pragma unroll
                for (int k = 0; k < B; ++k) need[k] = need[k] && !(bnd[k] < px.thr);
            }
            PT_STAMP(1);   // positions, cells, bound loads
            PT_COUNT(9);
            // fetch and transfer function -- behind ONE wave-uniform test: with the cull, seven rounds in eight need
            // neither for any lane, and a guard per step is a handful of scalar instructions each
            bool any_need = !cull;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                dens[k] = 0.f;
                al[k] = 0.f;
                any_need = any_need || need[k];
            }
            if (__ballot(any_need)) {
#pragma unroll
                for (int k = 0; k < B; ++k)
                    if (need[k] || (INSTR < 2 && !cull))
                        dens[k] = vol.linear(pk[k].x * 0.5f + 0.5f, pk[k].y * 0.5f + 0.5f, pk[k].z * 0.5f + 0.5f);
                // (the opacity only matters where the step was fetched: :432 below tests `need` first)
#pragma unroll
                for (int k = 0; k < B; ++k)
                    if (need[k] || !cull) al[k] = tff_linear_alpha(s_tff, tffn, dens[k]);
            }
            PT_STAMP(2);   // fetch + TF
            // the walk's exit conditions, in step order -- as selects, not branches (the bodies are assignments; as
            // nested ifs they were ~25 scalar mask instructions per step): step k happens while `run`; it leaves the
            // volume (:426-427), exceeds the step limit (:430-431) or is accepted (:432), in that order
            bool run = walking;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const bool st = run;
                px.cnt += st ? 1u : 0u;
                px.t = st ? tk[k] : px.t;
                const bool out = !ink[k];
                const bool over = px.cnt > 512u;
                const bool acc = need[k] && !(al[k] < px.thr);
                const bool accept = st && !out && !over && acc;
                const bool stop = st && (out || over || acc);
                if (INSTR) {
                    c_taken += (st && !out) ? 1u : 0u;
                    c_culled += (st && !out && !need[k]) ? 1u : 0u;
                }
                px.accepted = stop ? accept : px.accepted;
                px.apos.x = accept ? pk[k].x : px.apos.x;
                px.apos.y = accept ? pk[k].y : px.apos.y;
                px.apos.z = accept ? pk[k].z : px.apos.z;
                px.adens = accept ? dens[k] : px.adens;
                run = st && !stop;
            }
            if (walking && !run) state |= P_ENDED;
            PT_STAMP(3);   // exit conditions

            // ---- the leap: a walk whose batch ended with a rejected step in a macro cell (4^3 cells) whose bound is
            // below its threshold takes ALL its further steps inside that macro cell at once -- or inside the cube of
            // macro cells around it that CellView::cdist says are free as well.  Exact, because
            //  * every one of those steps is a rejection: its cell lies in the macro cell (the cube), so its bound is below
            //    the threshold -- and it lies in there because the last step taken does, the landing step does
            //    (checked below with the stepping code's own arithmetic) and a step's cell coordinate
            //    trunc(med3(fma(b, t, a))) is monotone in t, as is its position org + wdir * t, axis by axis: what holds
            //    at both ends of a stretch of the walk (the same macro cell, inside the volume) holds in between;
            //  * t after n steps is known in closed form while it stays in its binade: t + s is rounded to a multiple of
            //    ulp(t), s / ulp(t) = q + f with the same q and f at every step, so every step adds inc = q (f < 1/2) or
            //    q + 1 (f > 1/2) ulps to the bit pattern of t (f = 1/2 -- a tie, resolved by the parity of the sum --
            //    has no closed form and ends the stretch); at the binade's top one real step t + s crosses over, and the next
            //    stretch has its own inc;
            //  * the step counter stays within the limit of 512 (:430).
            // The number of steps comes from the macro cell's exit along the walk's line in cell space and the room in
            // the binade, both estimated (reciprocals) and then VERIFIED: landing cell, landing position, bit pattern.
            // A leap that lands (one step short of its cube's face) in a macro cell with free macro cells around it starts
            // the next one from there -- the landing step is a rejected step in a known macro cell like the batch's last
            // one -- for the price of that macro cell's two table entries (a dependent load, but from tables of a few
            // hundred KB) instead of a whole round: kLeapChain hops at most.
            bool go = run;
#pragma unroll 1
            for (int hop = 0; leap && hop < kLeapChain; ++hop) {
                const float sst = -px.dt;
                const uint32_t ti = __float_as_uint(px.t), si = __float_as_uint(sst);
                const int et = (int)(ti >> 23), es = (int)(si >> 23);   // (a sign bit makes the exponent >= 256)
                // (lrad != 0: the macro cell and those within lrad - 1 around it have bounds < tau_j <= thr)
                const bool can = go && (lrad != 0u || lcb < px.thr) && et > 0 && et < 255 && es > 0 && es < 255;
                if (!__ballot(can)) break;
                {
                    // -- how many steps: to where the line leaves the cube of macro cells [C - R, C + R] per axis, in cells
                    // (R = 0: the macro cell [4 C, 4 C + 4) itself), cut at the volume's own faces (cell coordinate 0 and
                    // res / E: a walk that leaves the volume inside the cube lands just before it does, instead of failing
                    // the check below and taking no leap at all) -- and within the step limit
                    const int R = lrad ? (int)lrad - 1 : 0;
                    const float inv_e = __uint_as_float((uint32_t)(127 - grid.shift) << 23);
                    const float ux = vol.fw * inv_e, uy = vol.fh * inv_e, uz = vol.fd * inv_e;
                    const float ex = __builtin_amdgcn_fmed3f((float)(((int)lcx + (lbx > 0.f ? R + 1 : -R)) * (1 << kLeapShift)), 0.f, ux);
                    const float ey = __builtin_amdgcn_fmed3f((float)(((int)lcy + (lby > 0.f ? R + 1 : -R)) * (1 << kLeapShift)), 0.f, uy);
                    const float ez = __builtin_amdgcn_fmed3f((float)(((int)lcz + (lbz > 0.f ? R + 1 : -R)) * (1 << kLeapShift)), 0.f, uz);
                    const float tx = lbx != 0.f ? (ex - lax) * __builtin_amdgcn_rcpf(lbx) : 3.0e38f;
                    const float ty = lby != 0.f ? (ey - lay) * __builtin_amdgcn_rcpf(lby) : 3.0e38f;
                    const float tz = lbz != 0.f ? (ez - laz) * __builtin_amdgcn_rcpf(lbz) : 3.0e38f;
                    const float t_out = vmin(tx, vmin(ty, tz));
                    const float n_cell = (t_out - px.t) * __builtin_amdgcn_rcpf(sst);
                    const float n_f = vmin(n_cell - VR_LEAP_MARGIN, (float)(512u - px.cnt));   // (cnt <= 512 while `run`)
                    uint32_t left = (can && n_f >= 1.f) ? (uint32_t)n_f : 0u;
                    // -- the steps themselves, on the bit pattern of t: kLeapPieces stretches in closed form, each within the
                    // binade t is in (inc ulps per step; none on a tie), with one real step t + s between two of them --
                    // the step that crosses into the next binade when the stretch before it reached the top
                    const uint32_t ms = (si & 0x7fffffu) | 0x800000u;
                    uint32_t tb = ti, n = 0;
#pragma unroll
                    for (int piece = 0; piece < kLeapPieces; ++piece) {
                        const int e_t = (int)(tb >> 23), dsh = e_t - es;
                        const bool closed = dsh >= 1 && dsh <= 24 && e_t < 255;
                        const uint32_t sh = (uint32_t)dsh & 31u;
                        const uint32_t q = ms >> sh, rem = ms & ((1u << sh) - 1u), half = (1u << sh) >> 1;
                        const uint32_t inc = q + (rem > half ? 1u : 0u);
                        const uint32_t room = (tb | 0x7fffffu) - tb;   // ulps to the top of the binade
                        // k = min(left, floor(room / inc)): the quotient from a reciprocal, exact after one correction
                        // each way when it matters (quotients up to left + 4 <= 516; a larger estimate is >= left for sure)
                        uint32_t k = (uint32_t)((float)room * __builtin_amdgcn_rcpf((float)(inc ? inc : 1u)));
                        if (k > left + 4u) {
                            k = left;
                        } else {
                            k -= (k * inc > room) ? 1u : 0u;
                            k += ((k + 1u) * inc <= room) ? 1u : 0u;
                            k = k < left ? k : left;
                        }
                        k = (closed && rem != half && __umulhi(k, inc) == 0u && k * inc <= room) ? k : 0u;
                        tb += k * inc;
                        left -= k;
                        n += k;
                        if (piece + 1 < kLeapPieces) {
                            const bool one = left != 0u;
                            tb = one ? __float_as_uint(__uint_as_float(tb) + sst) : tb;
                            left -= one ? 1u : 0u;
                            n += one ? 1u : 0u;
                        }
                    }
                    // -- the landing, with the stepping code's own arithmetic
                    const float tn = __uint_as_float(tb);
                    const f3 pn = add3(px.org, scale3(px.wdir, tn));
                    const float gmx = (float)(grid.cx - 1), gmy = (float)(grid.cy - 1), gmz = (float)(grid.cz - 1);
                    const uint32_t xn = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(lbx, tn, lax), 0.f, gmx);
                    const uint32_t yn = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(lby, tn, lay), 0.f, gmy);
                    const uint32_t zn = (uint32_t)(int)__builtin_amdgcn_fmed3f(__builtin_fmaf(lbz, tn, laz), 0.f, gmz);
                    const int ddx = (int)(xn >> kLeapShift) - (int)lcx, ddy = (int)(yn >> kLeapShift) - (int)lcy,
                              ddz = (int)(zn >> kLeapShift) - (int)lcz;
                    const bool ok = can && n != 0u && px.cnt + n <= 512u && in_volume(pn) && ddx >= -R && ddx <= R &&
                                    ddy >= -R && ddy <= R && ddz >= -R && ddz <= R;
                    px.t = ok ? tn : px.t;
                    px.cnt += ok ? n : 0u;
                    if (INSTR) {
                        c_taken += ok ? n : 0u;
                        c_culled += ok ? n : 0u;
                        c_leaped += ok ? n : 0u;
                    }
                    // the next hop: from the landing's macro cell, if the tables say there is room around it
                    go = false;
                    if (hop + 1 < kLeapChain && grid.cdist) {
                        const bool more = ok && px.cnt < 512u && llev != 0xffffffffu;
                        if (!__ballot(more)) break;
                        if (more) {
                            lcx = xn >> kLeapShift; lcy = yn >> kLeapShift; lcz = zn >> kLeapShift;
                            const uint32_t ci = (lcz * (uint32_t)grid.ccy + lcy) * (uint32_t)grid.ccx + lcx;
                            lrad = grid.cdist[llev + ci];
                            lcb = 0.f;                 // (lrad != 0 says the cell is free at the walk's level; 0: no hop)
                            go = lrad >= 2u;           // a cube of radius >= 1: further than the one step left in this cell
                        }
                    }
                }
            }
        }

        PT_STAMP(4);   // leap
        // ---- stage 3: trace_volume's control flow (:463-503) for lanes whose walk ended -- again
        //      only when enough lanes wait, or no lane walks any more
        const unsigned long long pend = __ballot((state & P_ENDED) != 0);
        walk_m = __ballot(state >= P_PRIMARY && state <= P_SHADOW);
        // (once the queue is empty nothing new will join the lanes that wait: a walk's end is handled at once, or its
        // pixel's next walk would start only when every other walk of the wave has ended -- walks in series, not side by side)
#ifndef VR_PT_DRAIN_SHADE_MIN
#define VR_PT_DRAIN_SHADE_MIN 1
#endif
        // (A wave learns that the queue is empty when it next draws from it, not before.  Looking at the queue's head
        // every few rounds instead -- thousands of waves reading the one address the draws update -- was measured at 2.4x
        // the kernel's time.)
        const int shade_min = drained ? VR_PT_DRAIN_SHADE_MIN : kShadeMin;
        const bool shade = pend && ((int)__builtin_popcountll(pend) >= shade_min || !walk_m);
        if (shade && (state & P_ENDED)) {
            const int ended = state & ~P_ENDED;
            bool start_shadow = false;
            if (ended == P_PRIMARY) {
                if (!px.accepted) {
                    state = P_WRITE;   // no interaction: the background colour, w = 1
                } else {
                    const float4 col = tff_linear(s_tff, tffn, px.adens);
                    px.c0 = col.x; px.c1 = col.y; px.c2 = col.z;
                    px.hit_pos = px.apos;
                    const f3 sp = mk3(px.apos.x * 0.5f + 0.5f, px.apos.y * 0.5f + 0.5f,
                                      px.apos.z * 0.5f + 0.5f);
#ifdef VR_DIAG_NO_GRAD   // diagnostic build (wrong image): what do the gradient neighbourhoods of the interactions cost?
                    const float4 gq = make_float4(sp.x, sp.y, sp.z, 0.f);
#else
                    const float4 gq = gradient_tff<VT, VI>(vol, s_tff, tffn, sp);
#endif
                    const float g0 = -gq.x, g1 = -gq.y, g2 = -gq.z, g3 = -gq.w;
                    const float glen = sqrtf((((g0 * g0) + (g1 * g1)) + (g2 * g2)) + (g3 * g3));
                    if (glen > 0.5f) {   // :483-486 high gradient: Phong
                        const f3 light = add3(neg3(px.dir), mk3(0.5f, 0.5f, 0.f));
                        const f3 c = illumination(mk3(px.c0, px.c1, px.c2), light, mk3(g0, g1, g2));
                        px.c0 = c.x; px.c1 = c.y; px.c2 = c.z;
                        start_shadow = true;
                    } else {             // :487-494 low gradient: second scatter ray
                        px.org = px.apos;
                        px.wdir = dir_phase_function(px.rnd);
                        px.t = 0.f;
                        px.cnt = 0;
                        state = P_SCATTER;
                    }
                }
            } else if (ended == P_SCATTER) {
                float s0 = px.env0, s1 = px.env1, s2 = px.env2;
                if (px.accepted) {
                    const float4 col = tff_linear(s_tff, tffn, px.adens);
                    s0 = col.x; s1 = col.y; s2 = col.z;
                }
                px.c0 = px.c0 + (s0 - px.c0) * 0.5f;   // mix(color, scatterColor, 0.5f), :493
                px.c1 = px.c1 + (s1 - px.c1) * 0.5f;
                px.c2 = px.c2 + (s2 - px.c2) * 0.5f;
                start_shadow = true;
            } else {   // P_SHADOW
                const float w = px.accepted ? 0.6f : 1.f;   // :497-499
                px.c0 = px.c0 * w; px.c1 = px.c1 * w; px.c2 = px.c2 * w;
                state = P_WRITE;
            }
            if (start_shadow) {   // :496-497 shadow ray towards the light, from the interaction
                px.org = px.hit_pos;
                px.wdir = add3(neg3(px.dir), mk3(0.5f, 0.5f, 0.f));
                px.t = 0.f;
                px.cnt = 0;
                state = P_SHADOW;
            }
            if (state == P_WRITE) {    // :689-704 accumulate + write
                const size_t fi = (size_t)px.gy * fr.W + px.gx;
                float r0 = px.c0, r1 = px.c1, r2 = px.c2;
                if (rp.iteration != 0) {
                    const float4 prev = fr.fb[fi];
                    const float it1 = (float)(rp.iteration + 1u);
                    r0 = prev.x + (r0 - prev.x) / it1;
                    r1 = prev.y + (r1 - prev.y) / it1;
                    r2 = prev.z + (r2 - prev.z) / it1;
                }
                const float4 o = make_float4(r0, r1, r2, 1.f);
                fr.fb[fi] = o;
                if (fr.out) fr.out[px.out_index] = o;
                state = P_FETCH;
            }
        }
        if (shade) {
            idle_m = __ballot(state == P_FETCH);
            walk_m = __ballot(state >= P_PRIMARY && state <= P_SHADOW);
            PT_STAMP(5);   // walk ends: shading, next walk, pixel write
            PT_COUNT(10);
        }
        if (drained && idle_m == ~0ull) break;
    }

#ifdef VR_STAMPS
    if (lane == 0) {
        const unsigned long long end_ = vr_stamp();
        for (int i_ = 0; i_ < 11; ++i_) atomicAdd(&g_pt_stamps[i_], pt_acc[i_]);
        atomicAdd(&g_pt_stamps[11], end_ - pt_first);                              // wave lifetime
        atomicAdd(&g_pt_stamps[12], pt_drained ? end_ - pt_drained : 0ull);        // ... of it after the queue was empty
        atomicMax(&g_pt_stamps[13], end_ - pt_first);                              // the longest wave
        atomicAdd(&g_pt_stamps[14], 1ull);
    }
#endif
    if (INSTR) {
        unsigned long long s = wave_sum(c_taken);
        if (lane == 0 && s) atomicAdd(&stats->v[0], s);
        s = wave_sum(c_hit);
        if (lane == 0 && s) atomicAdd(&stats->v[5], s);
        // technique 1 reuses the two brick counters: steps whose bound was consulted / culled
        s = wave_sum(c_culled);
        if (lane == 0 && s) atomicAdd(&stats->v[4], s);
        s = wave_sum(c_leaped);   // samples_nominal: the steps taken in leaps (among the culled ones)
        if (lane == 0 && s) atomicAdd(&stats->v[1], s);
        s = wave_sum(cull ? c_taken : 0ull);
        if (lane == 0 && s) atomicAdd(&stats->v[3], s);
    }
}

template <typename VT, int INSTR>
hipError_t launch_pt(const RaycastLaunch &a, hipStream_t stream)
{
    auto k = vr_pathtrace_kernel<VT, INSTR>;
    const size_t lds = (size_t)a.tf.tff_n * sizeof(float4);
    int nb = 0;
    {
        hipError_t e = vr_prepare_kernel(k, kBlockDim, lds, &nb, "pathtrace", a.num_cus);
        if (e != hipSuccess) return e;
    }
    const uint32_t cus = (uint32_t)(a.num_cus > 0 ? a.num_cus : 256);
    const uint32_t want = (a.frame.n_wave_tiles + 3u) / 4u;
    const uint32_t cap = cus * (uint32_t)nb;
    dim3 grid(want < cap ? want : cap), block(kBlockDim);
    if (grid.x == 0) return hipSuccess;
    const bool bind_stop = a.bind_events && a.stop_event && a.stop_bound;   // (one launch: it carries the frame's end)
    const bool bind_start = a.bind_events && a.start_event && a.start_bound;
    vr_launch_kernel(k, grid, block, lds, stream, bind_start ? a.start_event : nullptr, bind_stop ? a.stop_event : nullptr, a.vol, a.tf, a.cells, a.frame, a.cam,
                     a.render, a.pathtrace, a.stats, a.touched);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && bind_stop) *a.stop_bound = true;
    if (e == hipSuccess && bind_start) *a.start_bound = true;
    if (e == hipSuccess && a.mid_event) e = hipEventRecord(a.mid_event, stream);
    return e;
}

#ifdef VR_STAMPS
} // namespace
extern "C" int vrhip_debug_pt_stamps(unsigned long long out[16], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pt_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_pt_stamps), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
namespace {
#endif

template <typename VT>
hipError_t launch_pt_typed(const RaycastLaunch &a, hipStream_t stream)
{
    if (a.instr == 0) return launch_pt<VT, 0>(a, stream);
    if (a.instr == 1) return launch_pt<VT, 1>(a, stream);
    if (a.instr == 3) return launch_pt<VT, 3>(a, stream);
    return launch_pt<VT, 2>(a, stream);
}

} // namespace

hipError_t vr_launch_pathtrace(const RaycastLaunch &a, hipStream_t stream)
{
    if (a.info) {
        a.info->technique = 1;
        a.info->work_items = a.frame.n_wave_tiles;
        a.info->instrumented = (uint32_t)a.instr;
    }
    switch (a.format) {
    case VRHIP_UCHAR: return launch_pt_typed<uint8_t>(a, stream);
    case VRHIP_USHORT: return launch_pt_typed<uint16_t>(a, stream);
    case VRHIP_FLOAT: return launch_pt_typed<float>(a, stream);
    default: return hipErrorInvalidValue;
    }
}
