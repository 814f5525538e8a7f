// vrhip_api.hip -- implementation of the C ABI in include/vrhip.h on the HIP runtime.
// Owns every device allocation of one renderer (like VolumeRenderCL owns its cl::Image
// objects, /root/reference/src/core/volumerendercl.h:446-463).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vr_internal.h"

namespace {

struct VolumeSlot {
    void *dev = nullptr;      // micro-bricked voxels (channel 0 of a multi-channel volume)
    void *chan[3] = {nullptr, nullptr, nullptr};   // channels 1..3 of CL_RG / CL_RGBA volumes
    void *bricks = nullptr;   // (min,max) grid
    bool bricks_built = false;   // `bricks` holds the (min,max) of the voxels now in `dev`
    float2 *pt_minmax = nullptr;   // path tracer: per-cell (min,max) incl. halo, built on demand
    bool pt_minmax_valid = false;
    float2 *fine_minmax = nullptr; // ray caster: the same on the finer grid of the empty bits
    bool fine_minmax_valid = false;
    bool borrowed = false;    // dev / chan / bricks belong to another renderer (vrhip_share_volumes)
};

std::string g_create_error;

constexpr uint32_t kSkipLdsMaxBytes = 64 * 1024;   // bitmap staged in LDS up to this size

} // namespace

struct vrhip_renderer {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    mutable std::string err;
    std::string devname;

    uint32_t res[3] = {0, 0, 0};
    int format = -1;
    int channels = 1;                 // 1 = CL_R, 2 = CL_RG, 4 = CL_RGBA
    // HBM layout of a time step: 4x4x4-voxel micro-bricks (vr_internal.h, DESIGN.md "Data
    // layout"); nb = ceil(res / 4)
    uint32_t nb[3] = {0, 0, 0};
    std::vector<VolumeSlot> vols;
    uint32_t timestep = 0;

    float4 *tff = nullptr;
    uint32_t tff_n = 0;
    uint32_t *prefix = nullptr;
    uint32_t prefix_n = 0;

    uint32_t brick_tex[3] = {0, 0, 0}, brick_edge[3] = {0, 0, 0};
    float brick_res[3] = {1, 1, 1};
    bool bricks_valid = false;

    // ESS skip bitmap of the current timestep (derived from bricks + TF + prefix)
    uint32_t *skip_bits = nullptr;
    uint32_t skip_words = 0, skip_cap = 0;
    bool skip_dirty = true;
    // patch culling of the DDA pre-pass: the skip bitmap dilated by cull_radius bricks (SkipView::near_bits)
    uint32_t *near_bits = nullptr;
    uint8_t *near_scratch = nullptr;
    uint32_t cull_radius = 4;         // VRHIP_CULL_RADIUS (bricks; 0 = no patch culling)

    // cell grid of the current timestep + TF (CellView, vr_internal.h): opacity bound for the
    // path tracer, empty bits for the ray caster
    CellView cells = {nullptr, nullptr, 0, 0, 0, 3, 0, 0, 0, 3, nullptr, 0, 0, 0};
    unsigned long long *cell_bmask = nullptr;   // CellView::bmask (per ESS brick)
    size_t bmask_cap = 0;
    float *cell_bound = nullptr;
    uint32_t *cell_empty = nullptr;
    float *cell_sparse = nullptr;  // 13 x 4096 floats of scratch for the TF range-max table
    size_t cell_cap = 0, empty_cap = 0;
    bool cells_have_bound = false, cells_have_empty = false;   // r->cells' tables match volume, time step and TF
    bool pt_dirty = true;          // cells out of date (volume, timestep or TF changed)
    bool pt_cull = true;           // VRHIP_PT_NO_CULL=1 disables the path tracer's culling
    bool pt_leap = true;           // VRHIP_PT_NO_LEAP=1: no leaps over macro cells (A/B)
    bool pt_leap_far = true;       // VRHIP_PT_NO_FAR_LEAP=1: leaps stay inside one macro cell (A/B)
    uint8_t *cell_dist = nullptr;  // CellView::cdist (two buffers)
    const uint8_t *cell_dist_table = nullptr;
    size_t cell_dist_cap = 0;
    bool skip_empty = true;        // VRHIP_NO_EMPTY_SKIP=1 disables the ray caster's empty runs
    bool skip_empty_force = false; // VRHIP_EMPTY_SKIP=1: also where it is not expected to pay (see ray_skip_empty)

    vrhip_camera_params cam;
    vrhip_rendering_params render;
    vrhip_raycast_params raycast;
    vrhip_pathtrace_params pathtrace;
    bool use_ess = true;

    float4 *fb = nullptr;
    uint32_t fb_w = 0, fb_h = 0;

    DevStats *stats_dev = nullptr;
    bool stats_enabled = false;

    // work queue of 8x8 wave tiles (centre first) for the current frame/tile set
    WaveTile *queue_dev = nullptr;
    uint32_t queue_n = 0, queue_cap = 0;
    uint32_t *queue_head = nullptr;   // 2 x kControlWords (queue head, cont count, cont head, pad, sort bins + cursors):
                                      // the sets of launches alternate between the two blocks (FrameView::next_ctrl)
    uint8_t *patch_class = nullptr;   // FrameView::patch_class of the current queue, camera, parameters and skip bitmaps
    size_t patch_class_cap = 0;
    std::vector<uint8_t> patch_class_key;   // what the classes were computed for (empty: nothing valid)
    uint32_t skip_version = 0;        // bumped whenever the skip bitmaps are rebuilt
    uint32_t queue_version = 0;       // bumped whenever the work queue is rebuilt
    uint32_t queue_frames = 1;        // frames per set of the current queue
    bool use_patch_classes = true;    // VRHIP_NO_PATCH_CLASS=1 disables
    int occ_force = 0;                // VRHIP_OCC=2|3: waves per SIMD of the default marching kernels (0 = by volume)
    int occ_force_split = 0;          // ... of phase 2 (VRHIP_OCC sets both, VRHIP_OCC_P1 / VRHIP_OCC_P2 one)
    uint32_t ctrl_sel = 0;            // the block the next set of launches uses
    bool ctrl_clean[2] = {false, false};   // that block is known to hold zeroes
    bool phase_timing = false;        // vrhip_set_phase_timing: an event between the phases of a frame
    int event_bind = 2;               // VRHIP_EVENT_BIND (launch_timed)
    bool frame_timing = true;         // vrhip_set_frame_timing: events around a frame's launches (vrhip_last_kernel_seconds)
    uint16_t *cost = nullptr;         // per pixel: phase-2 rounds of the previous frame (sort key)
    uint32_t *order = nullptr;        // sorted permutation of the suspended rays
    ContRec *live_rays = nullptr;     // pre-pass output: live rays with their DDA state (phase 1's list)
    bool ray_list = true;             // VRHIP_NO_RAYLIST=1: phase 1 walks the live patches instead
    bool march = false;               // VRHIP_MARCH=1: vr_march_kernel instead of the two-phase march (measured, not faster)
    uint32_t lds_stage = 0;           // VRHIP_LDS_STAGE=1|2: the LDS brick staging experiment (vr_raycast_staged_kernel)
    uint32_t march_micro = 0, march_fill = 0;   // VRHIP_MARCH_MICRO / VRHIP_MARCH_FILL (0 = built-in)
    uint32_t *seeds_dev = nullptr;    // kMaxBatchFrames jitter seeds of a batch of frames
    bool sort_cont = true;            // VRHIP_NO_SORT=1 disables
    LiveTile *live = nullptr;         // DDA pre-pass output: patches with rays that sample
    bool prepass = true;              // VRHIP_NO_PREPASS=1 disables
    ContRec *cont = nullptr;          // suspended rays of the two-phase march
    size_t cont_cap = 0;
    uint32_t round_budget = 10;       // phase-1 sample rounds per patch (0 = single phase)
    uint32_t refill_min = 16;         // phase 2: idle ray slots per wave before a refill (VRHIP_REFILL_MIN)
    std::vector<uint32_t> queue_key;   // W, H, tile_w, tile_h, tile ids...
    // image-order ESS: ping-pong hit images (volumerendercl.cpp:482-488, :524-530) + per-frame scratch
    uint8_t *hit_in = nullptr, *hit_out = nullptr, *hit_status = nullptr, *hit_any = nullptr;
    uint32_t hit_w = 0, hit_h = 0;
    void *fp = nullptr;               // footprint volume of the current time step (VolView::fp)
    size_t fp_cap = 0;                // bytes allocated
    bool fp_valid = false;
    uint32_t fp_timestep = 0;         // the time step `fp` was built for
    uint32_t fp_candidate = 0xffffffffu, fp_candidate_frames = 0;   // time series: see ensure_footprint
    bool fp_active = false;           // this frame reads it
    const void *fp_use = nullptr;     // what this frame reads: the renderer's own `fp` or its owner's
    vrhip_renderer *vol_owner = nullptr;   // vrhip_share_volumes: whose voxels (and footprint volume) this renderer renders from
    std::vector<vrhip_renderer *> sharers; // the renderers that render from THIS one's voxels (their vol_owner is this)
    size_t fp_failed_bytes = 0;       // a footprint allocation of this size failed: not retried until the volume or the cap changes
    bool use_fp = true;               // VRHIP_NO_FOOTPRINT=1 disables
    size_t fp_max_bytes = (size_t)96 << 30;   // VRHIP_FOOTPRINT_MAX_GB
    float4 *env = nullptr;            // environment map (float RGBA), or nullptr
    uint32_t env_w = 0, env_h = 0;

    hipEvent_t ev0 = nullptr, ev1 = nullptr, evm = nullptr, evb0 = nullptr, evb1 = nullptr;
    bool timed = false, bricks_timed = false, phase_timed = false;
    vrhip_launch_info last_info;      // what the last render call launched (vrhip_last_launch_info)
    bool have_info = false;

    vrhip_renderer()
    {
        // defaults of volumerendercl.h:43-81
        std::memset(&cam, 0, sizeof cam);
        std::memset(&render, 0, sizeof render);
        std::memset(&raycast, 0, sizeof raycast);
        const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        std::memcpy(cam.viewMat, ident, sizeof ident);
        for (int i = 0; i < 3; ++i) { cam.bbox_bl[i] = -1.f; cam.bbox_tr[i] = 1.f; }
        for (int i = 0; i < 4; ++i) render.backgroundColor[i] = 1.f;
        for (int i = 0; i < 3; ++i) render.modelScale[i] = 1.f;
        render.illumType = 1;
        render.useLinear = 1;
        render.seed = 42;
        raycast.samplingRate = 1.5f;
        for (int i = 0; i < 3; ++i) raycast.brickRes[i] = 1.f;
        pathtrace.max_extinction = 100.f;
    }
};

namespace {

int fail(const vrhip_renderer *r, int code, const std::string &msg)
{
    if (r) r->err = msg;
    else g_create_error = msg;
    return code;
}

#define VR_HIP(r, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(r, VRHIP_ERR_HIP,                                                     \
                        std::string("ERROR: " #call " (") + hipGetErrorString(e_) + ")");     \
    } while (0)

#define VR_REQUIRE(r, cond, code, msg)                                                        \
    do {                                                                                      \
        if (!(cond)) return fail(r, code, msg);                                               \
    } while (0)

size_t fmt_bytes(int format) { return format == VRHIP_UCHAR ? 1 : format == VRHIP_USHORT ? 2 : 4; }

size_t volume_bytes(const vrhip_renderer *r)   // dense (host-side) size
{
    return (size_t)r->res[0] * r->res[1] * r->res[2] * fmt_bytes(r->format);
}

size_t volume_alloc_bytes(const vrhip_renderer *r)
{
    return (size_t)r->nb[0] * r->nb[1] * r->nb[2] * 64 * fmt_bytes(r->format);
}

void set_layout(vrhip_renderer *r)
{
    for (int i = 0; i < 3; ++i) r->nb[i] = (r->res[i] + 3) / 4;
}

VolView make_vol_view(const vrhip_renderer *r, const void *data);

// Dense x-fastest array (host memory, or device memory when dense_is_device) <-> micro-bricks,
// slab by slab (multiples of 4 slices) through a bounded device staging buffer.
hipError_t copy_volume(const vrhip_renderer *r, void *bricks_dev, void *dense, bool to_bricks,
                       bool dense_is_device, hipStream_t stream)
{
    const size_t bpv = fmt_bytes(r->format);
    const size_t slice_b = (size_t)r->res[0] * r->res[1] * bpv;
    const VolView v = make_vol_view(r, bricks_dev);
    if (dense_is_device) {
        hipError_t e = vr_launch_retile(v, r->format, dense, 0, (int)r->res[2], to_bricks, stream);
        return e != hipSuccess ? e : hipStreamSynchronize(stream);
    }
    int slab = (int)std::max<size_t>(4, ((size_t)256 << 20) / std::max<size_t>(slice_b, 1) / 4 * 4);
    slab = std::min<int>(slab, (int)((r->res[2] + 3) / 4 * 4));
    void *stage = nullptr;
    hipError_t e = hipMalloc(&stage, (size_t)slab * slice_b);
    if (e != hipSuccess) return e;
    for (int z0 = 0; z0 < (int)r->res[2] && e == hipSuccess; z0 += slab) {
        const int nz = std::min<int>(slab, (int)r->res[2] - z0);
        char *h = (char *)dense + (size_t)z0 * slice_b;
        if (to_bricks) {
            e = hipMemcpyAsync(stage, h, (size_t)nz * slice_b, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = vr_launch_retile(v, r->format, stage, z0, nz, true, stream);
        } else {
            e = vr_launch_retile(v, r->format, stage, z0, nz, false, stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(h, stage, (size_t)nz * slice_b, hipMemcpyDeviceToHost, stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);   // staging buffer is reused
    }
    (void)hipFree(stage);
    return e;
}

size_t bricks_bytes(const vrhip_renderer *r)
{
    return 2 * (size_t)r->brick_tex[0] * r->brick_tex[1] * r->brick_tex[2] * fmt_bytes(r->format);
}

// volumerendercl.cpp:39-54
uint32_t round_pow2(uint32_t n)
{
    uint32_t val = n - 1u;
    val |= val >> 1;
    val |= val >> 2;
    val |= val >> 4;
    val |= val >> 8;
    val |= val >> 16;
    val++;
    uint32_t x = val >> 1;
    return (val - n) > (n - x) ? x : val;
}

float inv_max_of(int format)
{
    return format == VRHIP_UCHAR ? 1.0f / 255.0f : format == VRHIP_USHORT ? 1.0f / 65535.0f : 1.0f;
}

VolView make_vol_view(const vrhip_renderer *r, const void *data)
{
    VolView v;
    v.data = data;
    v.w = (int)r->res[0]; v.h = (int)r->res[1]; v.d = (int)r->res[2];
    v.fw = (float)v.w; v.fh = (float)v.h; v.fd = (float)v.d;
    v.inv_max = inv_max_of(r->format);
    v.nbx = r->nb[0];
    v.nby = r->nb[1];
    v.nbz = r->nb[2];
    v.ystride = r->nb[0] * 64u;
    v.zstride = (unsigned long long)r->nb[0] * r->nb[1] * 64ull;
    v.chan[0] = v.chan[1] = v.chan[2] = nullptr;
    v.channels = 1;
    v.fp = nullptr;
    v.fp_nbx = (r->res[0] + 4u) >> 2;
    v.fp_nby = (r->res[1] + 4u) >> 2;
    return v;
}

// the render view of a time step: all channels
VolView make_render_view(const vrhip_renderer *r, const VolumeSlot &s)
{
    VolView v = make_vol_view(r, s.dev);
    for (int i = 0; i < 3; ++i) v.chan[i] = s.chan[i];
    v.channels = r->channels;
    v.fp = r->fp_active ? r->fp_use : nullptr;
    return v;
}

BrickView make_brick_view(const vrhip_renderer *r, const void *data)
{
    BrickView b;
    b.data = data;
    b.bw = (int)r->brick_tex[0];
    b.bh = (int)r->brick_tex[1];
    b.bd = (int)r->brick_tex[2];
    return b;
}

TfView make_tf_view(const vrhip_renderer *r)
{
    TfView t;
    t.tff = r->tff;
    t.tff_n = r->tff_n;
    t.prefix = r->prefix;
    t.prefix_n = r->prefix_n;
    return t;
}

// log2 of the cell edges of the two cell grids (CellView): opacity bounds on at most 256 cells per axis,
// empty bits on at most 512 (edges 8 and 4 up to 2048^3)
void cell_shifts(const vrhip_renderer *r, int *shift, int *eshift)
{
    const uint32_t mres = std::max(r->res[0], std::max(r->res[1], r->res[2]));
    int sh = 3, es = 2;
    while (((mres + (1u << sh) - 1) >> sh) > 256u) ++sh;
    if (const char *e = getenv("VRHIP_CELL_SHIFT")) es = std::max(2, std::min(5, atoi(e)));   // A/B: 3 = one grid
    while (((mres + (1u << es) - 1) >> es) > 512u) ++es;
    *shift = sh;
    *eshift = std::min(es, sh);
}

// Does the ray caster step over empty cells (CellView::empty)?  Where the ESS bricks are not much
// larger than the cells the brick skipping has done the work already and the lookahead only costs.
// Measured on MI355X with cells of 4 voxels: -13 % frame time at 1024^3 (bricks of 16), -30 % at 2048^3
// (bricks of 32), +3 % at 256^3 (bricks of 4); with cells of 8 (round 1): +6 % at 1024^3 and 256^3,
// -25 % at 2048^3.  So: bricks of at least four cells.  Without ESS bricks the cells are all there is.
bool ray_skip_empty(const vrhip_renderer *r)
{
    if (!r->skip_empty || r->channels > 1) return false;
    if (r->skip_empty_force || !r->use_ess || !r->bricks_valid) return true;
    int shift, eshift;
    cell_shifts(r, &shift, &eshift);
    return std::max(r->brick_edge[0], std::max(r->brick_edge[1], r->brick_edge[2])) >= (4u << eshift);
}

int set_device(const vrhip_renderer *r)
{
    VR_HIP(r, hipSetDevice(r->device));
    return VRHIP_OK;
}

// Sharers of `owner` lose their (borrowed) volumes: nothing of theirs points into the owner any more.
void detach_sharers(vrhip_renderer *owner)
{
    std::vector<vrhip_renderer *> list;
    list.swap(owner->sharers);
    for (vrhip_renderer *sh : list) {
        sh->vol_owner = nullptr;           // (so that clearing does not look for itself in `owner->sharers`)
        (void)vrhip_clear_volumes(sh);     // waits for the sharer's stream; borrowed slots are not freed
    }
}

// (re)allocate a slot for `timestep`, checking that res/format agree with other timesteps
int prepare_slot(vrhip_renderer *r, const uint32_t res[3], int format, uint32_t timestep,
                 VolumeSlot **slot, int channels = 1)
{
    VR_REQUIRE(r, res && res[0] && res[1] && res[2], VRHIP_ERR_INVALID,
               "Volume resolution must be non-zero.");
    VR_REQUIRE(r, format >= VRHIP_UCHAR && format <= VRHIP_FLOAT, VRHIP_ERR_INVALID,
               "Unknown or invalid volume data format.");   // volumerendercl.cpp:730
    VR_REQUIRE(r, res[0] <= 8192 && res[1] <= 8192 && res[2] <= 8192, VRHIP_ERR_INVALID,
               "Volume resolution above 8192 per axis is not supported.");
    bool same = r->format == format && r->res[0] == res[0] && r->res[1] == res[1] &&
                r->res[2] == res[2] && r->channels == channels;
    for (const VolumeSlot &vs : r->vols) same = same && !vs.borrowed;   // never write into shared voxels
    if (!same) {
        VR_REQUIRE(r, timestep == 0 || r->vols.empty(), VRHIP_ERR_INVALID,
                   "Volume size does not match size of the other time steps.");
        int rc = vrhip_clear_volumes(r);
        if (rc) return rc;
        std::memcpy(r->res, res, sizeof r->res);
        r->format = format;
        r->channels = channels;
        set_layout(r);
    }
    // Renderers that share these voxels hold copies of the slots' device pointers.  A slot that is
    // overwritten in place keeps its pointers: the sharers finish what they have in flight and drop
    // everything they derived from the old voxels.  A new time step would outgrow their slot lists, so
    // they are detached (they report "No volume data is loaded." until they share again).
    if (!r->sharers.empty()) {
        if (r->vols.size() <= timestep || !r->vols[timestep].dev) detach_sharers(r);
        for (vrhip_renderer *sh : r->sharers) {
            VR_HIP(r, hipStreamSynchronize(sh->stream));
            sh->skip_dirty = true;
            sh->pt_dirty = true;
            sh->cells_have_bound = sh->cells_have_empty = false;
            sh->bricks_valid = false;
            if (timestep < sh->vols.size()) {
                sh->vols[timestep].pt_minmax_valid = false;
                sh->vols[timestep].fine_minmax_valid = false;
            }
        }
    }
    if (r->vols.size() <= timestep) r->vols.resize(timestep + 1);
    VolumeSlot &s = r->vols[timestep];
    if (!s.dev) VR_HIP(r, hipMalloc(&s.dev, volume_alloc_bytes(r)));
    for (int c = 1; c < channels; ++c)
        if (!s.chan[c - 1]) VR_HIP(r, hipMalloc(&s.chan[c - 1], volume_alloc_bytes(r)));
    r->bricks_valid = false;
    s.bricks_built = false;
    r->skip_dirty = true;
    r->pt_dirty = true;
    r->fp_valid = false;
    r->fp_failed_bytes = 0;
    s.pt_minmax_valid = false;
    s.fine_minmax_valid = false;
    *slot = &s;
    return VRHIP_OK;
}

int ensure_fb(vrhip_renderer *r, uint32_t w, uint32_t h)
{
    if (r->fb && r->fb_w == w && r->fb_h == h) return VRHIP_OK;
    VR_HIP(r, hipStreamSynchronize(r->stream));
    if (r->fb) VR_HIP(r, hipFree(r->fb));
    r->fb = nullptr;
    VR_HIP(r, hipMalloc((void **)&r->fb, (size_t)w * h * sizeof(float4)));
    VR_HIP(r, hipMemsetAsync(r->fb, 0, (size_t)w * h * sizeof(float4), r->stream));
    if (r->cost) VR_HIP(r, hipFree(r->cost));
    r->cost = nullptr;
    VR_HIP(r, hipMalloc((void **)&r->cost, (size_t)w * h * sizeof(uint16_t)));
    VR_HIP(r, hipMemsetAsync(r->cost, 0, (size_t)w * h * sizeof(uint16_t), r->stream));
    r->fb_w = w;
    r->fb_h = h;
    return VRHIP_OK;
}

int check_renderable(vrhip_renderer *r, uint32_t width, uint32_t height)
{
    VR_REQUIRE(r, width > 0 && height > 0 && width <= 16384 && height <= 16384,
               VRHIP_ERR_INVALID, "Invalid output image size.");
    VR_REQUIRE(r, !r->vols.empty() && r->timestep < r->vols.size() && r->vols[r->timestep].dev,
               VRHIP_ERR_NODATA, "No volume data is loaded.");
    VR_REQUIRE(r, r->tff && r->tff_n, VRHIP_ERR_NODATA, "No transfer function set.");
    if (r->use_ess && r->render.technique == 0) {
        VR_REQUIRE(r, r->bricks_valid && r->vols[r->timestep].bricks, VRHIP_ERR_NODATA,
                   "ESS bricks not built: call vrhip_build_bricks after the volume upload.");
        VR_REQUIRE(r, r->prefix && r->prefix_n, VRHIP_ERR_NODATA,
                   "No transfer function prefix sum set.");
    }
    VR_REQUIRE(r, r->render.technique <= 1, VRHIP_ERR_INVALID, "Unknown rendering technique.");
    // the path-tracing branch of the kernel returns before illumType is looked at (:686-706)
    VR_REQUIRE(r, r->render.illumType <= 5 || r->render.technique == 1, VRHIP_ERR_INVALID,
               "Unknown illumination type.");
    VR_REQUIRE(r, r->render.technique == 0 || r->pathtrace.max_extinction > 0.f, VRHIP_ERR_INVALID,
               "max_extinction must be positive.");
    // technique 1 and the traffic / downsampling helpers read channel 0 only, like the kernel's
    // .x readers; nothing else to check for CL_RG / CL_RGBA volumes
    // the path-tracing branch returns before the hit image is written (:686-706): its state
    // would never change
    VR_REQUIRE(r, !(r->render.imgEss && r->render.technique == 1), VRHIP_ERR_UNSUPPORTED,
               "image-order ESS is not supported with the path tracer.");
    return VRHIP_OK;
}

// The two hit images of image-order ESS, (W/8 + 1) x (H/8 + 1) texels, as updateOutputImg
// creates them (volumerendercl.cpp:482-488): the input image is initialised from a vector of
// 32-bit ones read as one byte per texel (bytes 1,0,0,0,1,...); the output image is left
// uninitialised there, zero here.
int ensure_hit_images(vrhip_renderer *r, uint32_t w, uint32_t h)
{
    const uint32_t hw = w / 8u + 1u, hh = h / 8u + 1u;
    if (r->hit_in && r->hit_w == hw && r->hit_h == hh) return VRHIP_OK;
    VR_HIP(r, hipStreamSynchronize(r->stream));
    for (uint8_t **p : {&r->hit_in, &r->hit_out, &r->hit_status, &r->hit_any}) {
        if (*p) VR_HIP(r, hipFree(*p));
        *p = nullptr;
    }
    r->hit_w = r->hit_h = 0;
    const size_t n = (size_t)hw * hh;
    for (uint8_t **p : {&r->hit_in, &r->hit_out, &r->hit_status, &r->hit_any}) {
        VR_HIP(r, hipMalloc((void **)p, n));
        VR_HIP(r, hipMemset(*p, 0, n));
    }
    std::vector<uint8_t> init(n, 0);
    for (size_t i = 0; i < n; i += 4) init[i] = 1;
    VR_HIP(r, hipMemcpy(r->hit_in, init.data(), n, hipMemcpyHostToDevice));
    r->hit_w = hw;
    r->hit_h = hh;
    return VRHIP_OK;
}

// Footprint volume of the current time step (VolView::fp, DESIGN.md "Footprint volume"): 8x the
// volume's bytes, so that a trilinear fetch is one load and none of the micro-brick address
// arithmetic.  The march is bound by VALU issue, so the instructions saved are frame time: -12 % at
// 2048^3 UCHAR in throughput mode (69 GB; 0.297 -> 0.261 ms), -4 % one frame at a time, -7 % on the
// dense "haze" volume (round 2; round 1 had measured +-0 one frame at a time and capped it at 33 GB).
// 288 GB of HBM are what this is for: the cap is 96 GB, one copy per GPU (renderers that share an
// owner's voxels read the owner's).  Built on first use, kept until the volume or the time step
// changes.
int ensure_footprint(vrhip_renderer *r)
{
    r->fp_active = false;
    const vrhip_rendering_params &rp = r->render;
    if (!r->use_fp || r->lds_stage || r->channels > 1 || r->stats_enabled || rp.technique != 0 || rp.illumType >= 2 ||
        r->raycast.useAO || rp.showEss || rp.imgEss)
        return VRHIP_OK;   // (the launcher uses it in the default kernels only)
    const VolView v = make_vol_view(r, r->vols[r->timestep].dev);
    const size_t bytes = (size_t)v.fp_nbx * v.fp_nby * ((size_t)(r->res[2] + 4u) >> 2) * 64u * 8u *
                         fmt_bytes(r->format);
    if (bytes > r->fp_max_bytes) return VRHIP_OK;
    // a renderer that shares an owner's voxels reads the owner's footprint volume of the same time step
    // (69 GB at 2048^3: one copy per GPU, not one per frame in flight); the owner builds it with its
    // first frame, which in every driver here comes before the twins' frames
    if (r->vol_owner) {
        const vrhip_renderer *o = r->vol_owner;
        if (o->fp && o->fp_valid && o->fp_timestep == r->timestep) {
            r->fp_use = o->fp;
            r->fp_active = true;
        }
        return VRHIP_OK;   // (no copy of its own: the plain layout until the owner has one)
    }
    if (!r->fp_valid || r->fp_timestep != r->timestep) {
        // A time series that moves on with every frame would rebuild 8x the volume per frame (40 ms at
        // 2048^3 against 0.6 ms for the frame from the plain layout): with more than one time step the
        // footprint volume is built for a step only once three frames in a row have shown it.
        if (r->vols.size() > 1) {
            if (r->fp_candidate != r->timestep) { r->fp_candidate = r->timestep; r->fp_candidate_frames = 0; }
            if (++r->fp_candidate_frames < 3u) return VRHIP_OK;
        }
        if (bytes == r->fp_failed_bytes) return VRHIP_OK;   // this allocation failed before: plain layout
        // renderers that share this one's volumes may be reading the old one on their own streams
        if (r->fp)
            for (vrhip_renderer *sh : r->sharers) VR_HIP(r, hipStreamSynchronize(sh->stream));
        r->fp_valid = false;
        if (bytes > r->fp_cap) {
            if (r->fp) {
                VR_HIP(r, hipStreamSynchronize(r->stream));
                VR_HIP(r, hipFree(r->fp));
            }
            r->fp = nullptr;
            r->fp_cap = 0;
            if (hipMalloc(&r->fp, bytes) != hipSuccess) {   // not enough HBM left: plain layout
                (void)hipGetLastError();
                r->fp = nullptr;
                r->fp_failed_bytes = bytes;   // (reset by a new volume or a new cap)
                return VRHIP_OK;
            }
            r->fp_cap = bytes;
        }
        VolView fv = v;
        fv.fp = r->fp;
        VR_HIP(r, vr_launch_build_footprint(fv, r->format, r->stream));
        VR_HIP(r, hipStreamSynchronize(r->stream));   // sharers render on other streams
        r->fp_valid = true;
        r->fp_timestep = r->timestep;
    }
    r->fp_use = r->fp;
    r->fp_active = true;
    return VRHIP_OK;
}

// Recompute the ESS skip bitmap when bricks, TF, prefix sum or timestep changed.
int ensure_skipmap(vrhip_renderer *r)
{
    if (!r->use_ess || !r->skip_dirty) return VRHIP_OK;
    const size_t n = (size_t)r->brick_tex[0] * r->brick_tex[1] * r->brick_tex[2];
    const uint32_t words = (uint32_t)(2 * ((n + 63) / 64));
    if (words + 1 > r->skip_cap) {
        VR_HIP(r, hipStreamSynchronize(r->stream));
        if (r->skip_bits) VR_HIP(r, hipFree(r->skip_bits));
        r->skip_bits = nullptr;
        VR_HIP(r, hipMalloc((void **)&r->skip_bits, ((size_t)words + 1) * sizeof(uint32_t)));
        if (r->near_bits) VR_HIP(r, hipFree(r->near_bits));
        if (r->near_scratch) VR_HIP(r, hipFree(r->near_scratch));
        r->near_bits = nullptr;
        r->near_scratch = nullptr;
        VR_HIP(r, hipMalloc((void **)&r->near_bits, ((size_t)words + 1) * sizeof(uint32_t)));
        VR_HIP(r, hipMalloc((void **)&r->near_scratch, 2 * n));
        r->skip_cap = words + 1;
    }
    r->skip_words = words;
    VR_HIP(r, vr_launch_skipmap(make_brick_view(r, r->vols[r->timestep].bricks), r->format,
                                inv_max_of(r->format), make_tf_view(r), r->skip_bits, words,
                                r->stream));
    if (r->cull_radius)
        VR_HIP(r, vr_launch_skip_near(make_brick_view(r, r->vols[r->timestep].bricks), r->skip_bits, words,
                                      r->cull_radius, r->near_scratch, r->near_bits, r->stream));
    r->skip_dirty = false;
    ++r->skip_version;
    return VRHIP_OK;
}

// The (min, max) of one cell grid of the current time step (geometry: g.cx.., g.shift), built with the
// separable streaming kernels where they apply.
static int build_cell_minmax(vrhip_renderer *r, VolumeSlot &s, const CellView &g, float2 *out)
{
    // scratch for the separable build (one (min, max) per cell column and voxel slice), held only
    // while it runs
    void *records = nullptr;
    const size_t n_rec = (size_t)g.cx * g.cy * r->res[2];
    if (g.shift <= 4 && r->nb[0] <= 1600u && !getenv("VRHIP_CELLS_PER_WAVE") &&
        hipMalloc(&records, n_rec * vr_cell_record_bytes(r->format)) != hipSuccess) {
        (void)hipGetLastError();
        records = nullptr;
    }
    hipError_t e = vr_launch_cell_minmax(make_vol_view(r, s.dev), r->format, g, out, r->stream, records);
    if (records) {
        if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
        (void)hipFree(records);
    }
    VR_HIP(r, e);
    return VRHIP_OK;
}

// (Re)build what the coming frame needs of the cell grids when the volume, the timestep or the TF
// changed: the opacity bounds (path tracer; cells of 2^shift voxels, at most 256 per axis: shift 3
// up to 2048^3) and / or the empty bits (ray caster; cells of 2^eshift voxels, at most 512 per axis:
// shift 2 up to 2048^3, 16 MiB of bits).
int ensure_cells(vrhip_renderer *r, bool need_bound, bool need_empty)
{
    if (r->pt_dirty) {
        r->cells_have_bound = r->cells_have_empty = false;
        r->pt_dirty = false;
    }
    if ((!need_bound || r->cells_have_bound) && (!need_empty || r->cells_have_empty)) return VRHIP_OK;
    VolumeSlot &s = r->vols[r->timestep];
    int shift, eshift;
    cell_shifts(r, &shift, &eshift);
    CellView g = r->cells;
    g.shift = shift;
    g.cx = (int)((r->res[0] + (1u << shift) - 1) >> shift);
    g.cy = (int)((r->res[1] + (1u << shift) - 1) >> shift);
    g.cz = (int)((r->res[2] + (1u << shift) - 1) >> shift);
    g.eshift = eshift;
    g.ecx = (int)((r->res[0] + (1u << eshift) - 1) >> eshift);
    g.ecy = (int)((r->res[1] + (1u << eshift) - 1) >> eshift);
    g.ecz = (int)((r->res[2] + (1u << eshift) - 1) >> eshift);
    const size_t n_cells = (size_t)g.cx * g.cy * g.cz, n_fine = (size_t)g.ecx * g.ecy * g.ecz;
    CellView fine = g;   // the fine grid's geometry for the kernels that take one grid
    fine.shift = eshift; fine.cx = g.ecx; fine.cy = g.ecy; fine.cz = g.ecz;
    if (!r->cell_sparse) VR_HIP(r, hipMalloc((void **)&r->cell_sparse, 13 * 4096 * sizeof(float)));

    // ---- (min, max) per cell: a property of the voxels, kept per time step.  A renderer that shares
    // another one's voxels (vrhip_share_volumes) reads the owner's tables where the owner has built them.
    const bool one_grid = eshift == shift;
    const float2 *fine_mm = nullptr, *coarse_mm = nullptr;
    if (r->vol_owner && r->timestep < r->vol_owner->vols.size()) {
        const VolumeSlot &os = r->vol_owner->vols[r->timestep];
        if (os.dev == s.dev) {
            bool use = false;
            if (os.fine_minmax_valid && !s.fine_minmax_valid) { fine_mm = os.fine_minmax; use = true; }
            if (os.pt_minmax_valid && !s.pt_minmax_valid) { coarse_mm = os.pt_minmax; use = true; }
            // (what the owner's stream has written must be complete before this renderer's stream reads it)
            if (use && r->vol_owner->stream != r->stream) VR_HIP(r, hipStreamSynchronize(r->vol_owner->stream));
        }
    }
    if (need_empty && !one_grid && !s.fine_minmax_valid && !fine_mm) {
        if (!s.fine_minmax) VR_HIP(r, hipMalloc((void **)&s.fine_minmax, n_fine * sizeof(float2)));
        int rc = build_cell_minmax(r, s, fine, s.fine_minmax);
        if (rc) return rc;
        s.fine_minmax_valid = true;
    }
    if (s.fine_minmax_valid) fine_mm = s.fine_minmax;
    if ((need_bound || one_grid) && !s.pt_minmax_valid && !coarse_mm) {
        if (!s.pt_minmax) VR_HIP(r, hipMalloc((void **)&s.pt_minmax, n_cells * sizeof(float2)));
        if (fine_mm && !one_grid) {
            VR_HIP(r, vr_launch_cell_reduce(fine_mm, g, s.pt_minmax, r->stream));
        } else {
            int rc = build_cell_minmax(r, s, g, s.pt_minmax);
            if (rc) return rc;
        }
        s.pt_minmax_valid = true;
    }
    if (s.pt_minmax_valid) coarse_mm = s.pt_minmax;

    // ---- with the transfer function: bounds, empty bits
    // (the macro cells' bounds -- CellView::cbound -- sit behind the cells' in the same allocation)
    g.ccx = (g.cx + (1 << kLeapShift) - 1) >> kLeapShift;
    g.ccy = (g.cy + (1 << kLeapShift) - 1) >> kLeapShift;
    g.ccz = (g.cz + (1 << kLeapShift) - 1) >> kLeapShift;
    const size_t n_macro = (size_t)g.ccx * g.ccy * g.ccz;
    if (need_bound && !r->cells_have_bound) {
        if (n_cells + n_macro > r->cell_cap) {
            VR_HIP(r, hipStreamSynchronize(r->stream));
            if (r->cell_bound) VR_HIP(r, hipFree(r->cell_bound));
            r->cell_bound = nullptr;
            r->cell_cap = 0;
            VR_HIP(r, hipMalloc((void **)&r->cell_bound, (n_cells + n_macro) * sizeof(float)));
            r->cell_cap = n_cells + n_macro;
        }
        VR_HIP(r, vr_launch_cell_bounds(coarse_mm, g, inv_max_of(r->format), make_tf_view(r),
                                        r->cell_sparse, r->cell_bound, nullptr, r->stream));
        g.bound = r->cell_bound;
        VR_HIP(r, vr_launch_cell_coarse_bounds(g, r->cell_bound + n_cells, r->stream));
        // how far the macro cells around one are free too (CellView::cdist)
        if (2 * (size_t)kLeapLevels * n_macro > r->cell_dist_cap) {
            VR_HIP(r, hipStreamSynchronize(r->stream));
            if (r->cell_dist) VR_HIP(r, hipFree(r->cell_dist));
            r->cell_dist = nullptr;
            r->cell_dist_cap = 0;
            VR_HIP(r, hipMalloc((void **)&r->cell_dist, 2 * (size_t)kLeapLevels * n_macro));
            r->cell_dist_cap = 2 * (size_t)kLeapLevels * n_macro;
        }
        VR_HIP(r, vr_launch_cell_leap_radius(g, r->cell_bound + n_cells, r->cell_dist, &r->cell_dist_table, r->stream));
        r->cells_have_bound = true;
    }
    g.bound = r->cells_have_bound ? r->cell_bound : nullptr;
    g.cbound = (r->cells_have_bound && r->pt_leap) ? r->cell_bound + n_cells : nullptr;
    g.cdist = (g.cbound && r->pt_leap_far) ? r->cell_dist_table : nullptr;
    if (need_empty && !r->cells_have_empty) {
        if (n_fine > r->empty_cap) {
            VR_HIP(r, hipStreamSynchronize(r->stream));
            if (r->cell_empty) VR_HIP(r, hipFree(r->cell_empty));
            r->cell_empty = nullptr;
            r->empty_cap = 0;
            VR_HIP(r, hipMalloc((void **)&r->cell_empty, ((n_fine + 63) / 64) * 2 * sizeof(uint32_t)));
            r->empty_cap = n_fine;
        }
        VR_HIP(r, vr_launch_cell_bounds(one_grid ? coarse_mm : fine_mm, fine, inv_max_of(r->format),
                                        make_tf_view(r), r->cell_sparse, nullptr, r->cell_empty, r->stream));
        r->cells_have_empty = true;
        g.empty = r->cell_empty;
        // the empty bits per ESS brick, for the march kernel (ESS bricks of >= 4 voxels per axis); only
        // the opt-in kernels read them (VRHIP_MARCH, leap stepping), so the default path does not pay the
        // 0.2 ms per transfer-function change
        g.bmask = nullptr;
        g.bex = g.bey = g.bez = 0;
        if (r->bricks_valid && (r->march || r->march_micro)) {
            int lg[3];
            bool ok = true;
            for (int i = 0; i < 3; ++i) {
                lg[i] = 0;
                while ((1u << lg[i]) < r->brick_edge[i]) ++lg[i];
                ok = ok && lg[i] >= 2 && r->brick_tex[i] <= 256u;
            }
            if (ok) {
                const size_t nb = (size_t)r->brick_tex[0] * r->brick_tex[1] * r->brick_tex[2];
                if (nb > r->bmask_cap) {
                    VR_HIP(r, hipStreamSynchronize(r->stream));
                    if (r->cell_bmask) VR_HIP(r, hipFree(r->cell_bmask));
                    r->cell_bmask = nullptr;
                    r->bmask_cap = 0;
                    VR_HIP(r, hipMalloc((void **)&r->cell_bmask, nb * sizeof(unsigned long long)));
                    r->bmask_cap = nb;
                }
                g.bex = lg[0]; g.bey = lg[1]; g.bez = lg[2];
                VR_HIP(r, vr_launch_cell_bmask(make_vol_view(r, s.dev), g, (int)r->brick_tex[0], (int)r->brick_tex[1],
                                               (int)r->brick_tex[2], r->cell_bmask, r->stream));
                g.bmask = r->cell_bmask;
            }
        }
    }
    if (!r->cells_have_empty) { g.empty = nullptr; g.bmask = nullptr; }
    r->cells = g;
    return VRHIP_OK;
}

// Build (or reuse) the centre-first queue of 8x8 wave tiles.  tile_ids == nullptr: the whole
// frame, `out` in frame layout; else the listed tiles, `out` compact [n][tile_h][tile_w].
int ensure_queue(vrhip_renderer *r, uint32_t W, uint32_t H, uint32_t tile_w, uint32_t tile_h,
                 const uint32_t *tile_ids, uint32_t n_tiles, uint32_t n_frames = 1,
                 uint32_t frame_stride = 0)
{
    std::vector<uint32_t> key = {W, H, tile_w, tile_h, tile_ids ? 1u : 0u, n_frames, frame_stride};
    if (tile_ids) key.insert(key.end(), tile_ids, tile_ids + n_tiles);
    if (key == r->queue_key && r->queue_dev) return VRHIP_OK;

    struct Item { float d2; WaveTile wt; };
    std::vector<Item> items;
    const float cx = 0.5f * (float)W, cy = 0.5f * (float)H;
    auto push = [&](uint32_t px0, uint32_t py0, uint32_t out_base) {
        if (px0 >= W || py0 >= H) return;
        Item it;
        float dx = (float)px0 + 4.f - cx, dy = (float)py0 + 4.f - cy;
        it.d2 = dx * dx + dy * dy;
        it.wt.tx8 = (uint16_t)(px0 / 8);
        it.wt.ty8 = (uint16_t)(py0 / 8);
        it.wt.out_base = out_base;
        items.push_back(it);
    };
    if (!tile_ids) {
        for (uint32_t y = 0; y < H; y += 8)
            for (uint32_t x = 0; x < W; x += 8) push(x, y, y * W + x);
    } else {
        const uint32_t tiles_x = (W + tile_w - 1) / tile_w;
        for (uint32_t k = 0; k < n_tiles; ++k) {
            const uint32_t tx = tile_ids[k] % tiles_x, ty = tile_ids[k] / tiles_x;
            for (uint32_t y = 0; y < tile_h; y += 8)
                for (uint32_t x = 0; x < tile_w; x += 8)
                    push(tx * tile_w + x, ty * tile_h + y, (k * tile_h + y) * tile_w + x);
        }
    }
    std::stable_sort(items.begin(), items.end(),
                     [](const Item &a, const Item &b) { return a.d2 < b.d2; });
    // a batch of frames: every patch once per frame (frame index in the upper bits of ty8 / tx8), the
    // frames' outputs one after the other
    const uint32_t frame_pixels = frame_stride ? frame_stride : (tile_ids ? n_tiles * tile_w * tile_h : W * H);
    std::vector<WaveTile> q(items.size() * n_frames);
    for (size_t i = 0; i < items.size(); ++i)
        for (uint32_t f = 0; f < n_frames; ++f) {
            WaveTile wt = items[i].wt;
            wt_set_frame(wt, f);
            wt.out_base += f * frame_pixels;
            q[i * n_frames + f] = wt;
        }

    VR_HIP(r, hipStreamSynchronize(r->stream));
    if (q.size() > r->queue_cap) {
        if (r->queue_dev) VR_HIP(r, hipFree(r->queue_dev));
        r->queue_dev = nullptr;
        if (r->live) VR_HIP(r, hipFree(r->live));
        r->live = nullptr;
        VR_HIP(r, hipMalloc((void **)&r->live, q.size() * sizeof(LiveTile)));
        VR_HIP(r, hipMalloc((void **)&r->queue_dev, q.size() * sizeof(WaveTile)));
        r->queue_cap = (uint32_t)q.size();
    }
    if (!q.empty())
        VR_HIP(r, hipMemcpy(r->queue_dev, q.data(), q.size() * sizeof(WaveTile),
                            hipMemcpyHostToDevice));
    r->queue_n = (uint32_t)q.size();
    r->queue_key.swap(key);
    r->queue_frames = n_frames;
    ++r->queue_version;
    // continuation buffer of the two-phase march: worst case every ray is suspended
    const size_t need = q.size() * 64;
    if (need > r->cont_cap) {
        if (r->cont) VR_HIP(r, hipFree(r->cont));
        r->cont = nullptr;
        r->cont_cap = 0;
        VR_HIP(r, hipMalloc((void **)&r->cont, need * sizeof(ContRec)));
        if (r->order) VR_HIP(r, hipFree(r->order));
        r->order = nullptr;
        VR_HIP(r, hipMalloc((void **)&r->order, need * sizeof(uint32_t)));
        if (r->live_rays) VR_HIP(r, hipFree(r->live_rays));
        r->live_rays = nullptr;
        VR_HIP(r, hipMalloc((void **)&r->live_rays, (size_t)kLiveLists * live_list_cap((uint32_t)q.size()) * sizeof(ContRec)));
        r->cont_cap = need;
    }
    return VRHIP_OK;
}

void fill_launch(vrhip_renderer *r, uint32_t width, uint32_t height, uint32_t out_stride,
                 RaycastLaunch *a)
{
    std::memset(a, 0, sizeof *a);
    const VolumeSlot &s = r->vols[r->timestep];
    a->vol = make_render_view(r, s);
    a->bricks = make_brick_view(r, s.bricks);
    a->tf = make_tf_view(r);
    a->skip.bits = r->skip_bits;
    a->skip.n_words = r->skip_words;
    a->skip.in_lds = ((size_t)r->skip_words + 1) * sizeof(uint32_t) <= kSkipLdsMaxBytes ? 1u : 0u;
    if (getenv("VRHIP_SKIP_GLOBAL")) a->skip.in_lds = 0;   // experiments: bitmap from L2/HBM
    a->skip.near_bits = r->cull_radius ? r->near_bits : nullptr;
    a->skip.near_r = r->cull_radius;
    a->frame.W = width;
    a->frame.H = height;
    // padded NDRange of the reference (volumerendercl.cpp:513-514): a full extra group
    // when the size is already a multiple of 8; the camera is derived from it.
    a->frame.gsx = width + (8u - width % 8u);
    a->frame.gsy = height + (8u - height % 8u);
    {
        const float gsx = (float)a->frame.gsx, gsy = (float)a->frame.gsy;
        float aspect = gsy / gsx;
        const float other = gsx / gsy;
        aspect = other < aspect ? other : aspect;   // vmin(aspect, other), vr_device_math.h
        a->frame.ray_aspect = aspect;
        a->frame.ray_psx = 2.f / gsx;
        a->frame.ray_psy = 2.f / gsy;
    }
    a->frame.queue = r->queue_dev;
    a->frame.n_wave_tiles = r->queue_n;
    a->frame.out_stride = out_stride;
    uint32_t *const ctrl = r->queue_head + (size_t)r->ctrl_sel * kControlWords;
    a->frame.queue_head = ctrl;
    a->frame.next_ctrl = r->queue_head + (size_t)(r->ctrl_sel ^ 1u) * kControlWords;
    a->frame.patch_class = nullptr;   // (launch_timed: ensure_patch_classes)
    a->frame.set_frames = 1;
    a->frame.cont = r->cont;
    a->frame.cont_count = ctrl + 1;
    a->frame.cont_head = ctrl + 2;
    a->frame.round_budget = r->cont ? r->round_budget : 0;
    a->frame.refill_min = r->refill_min;
    a->frame.live = r->prepass ? r->live : nullptr;
    a->frame.live_rays = (r->prepass && r->ray_list) ? r->live_rays : nullptr;
    a->frame.march = r->march ? 1u : 0u;
    a->frame.lds_stage = r->lds_stage;
    a->frame.march_micro = r->march_micro;
    a->frame.march_fill = r->march_fill;
    a->frame.live_count = ctrl + 3;
    a->frame.live_list_count = ctrl + kLiveBase;
    a->frame.draw_count = ctrl + kDrawBase;
    a->frame.cost = r->sort_cont ? r->cost : nullptr;
    a->frame.order = r->sort_cont && r->cost ? r->order : nullptr;
    a->frame.sort_ws = ctrl + 4;
    a->frame.fb = r->fb;
    if (r->render.imgEss && r->render.technique == 0) {
        a->frame.hit_in = r->hit_in;
        a->frame.hit_status = r->hit_status;
        a->frame.hit_any = r->hit_any;
        a->frame.hit_w = r->hit_w;
        a->frame.hit_h = r->hit_h;
    }
    a->frame.env = r->env;
    a->frame.env_w = r->env_w;
    a->frame.env_h = r->env_h;
    // showEss needs the position of every ray's last sample: it is kept in registers, not in the
    // continuation records, so the march runs in a single phase
    if (r->render.showEss) a->frame.round_budget = 0;
    a->cam = r->cam;
    a->render = r->render;
    a->raycast = r->raycast;
    a->pathtrace = r->pathtrace;
    a->cells = r->cells;
    if (!r->pt_cull) { a->cells.bound = a->cells.cbound = nullptr; a->cells.cdist = nullptr; }
    // the empty bits are those of TF(channel 0): not what a CL_RG / CL_RGBA sample's opacity is
    if (!ray_skip_empty(r)) { a->cells.empty = nullptr; a->cells.bmask = nullptr; }
    a->format = r->format;
    a->use_ess = r->use_ess ? 1 : 0;
    a->instr = r->stats_enabled ? 1 : 0;
    // Three waves per SIMD -- one workgroup of 12 waves per CU that share one transfer function and one skip bitmap
    // in LDS, 168 VGPRs per lane (vr_raycast.hip kWavesWide) -- where waves wait more than they issue:
    //  * volumes whose ESS bricks are too small for the empty-run lookahead (ray_skip_empty): the march waits for its
    //    fetches instead of stepping over empty cells (256^3: -12 % per frame, -6 % one frame at a time);
    //  * launch sets of several frames (the throughput schedule, a rank's tile share): enough rays for a third wave
    //    per SIMD, and no chain of one frame's longest rays that a third wave would stretch (2048^3, 16 frames per
    //    set and two renderers: "shells" -6 %, "haze" -8 %, 1024^3 USHORT -8 %; both phases -- one of them alone
    //    gains half of it or nothing).
    // One frame at a time with the lookahead: two waves ("shells" +10 % with three, the longest rays' chain).
    // VRHIP_OCC=2|3 forces both phases, VRHIP_OCC_P1 / VRHIP_OCC_P2 one of them (tools/ab_env.sh).
    const bool three = r->use_ess && (!ray_skip_empty(r) || r->queue_frames >= 4u);
    a->occ3 = r->occ_force ? (r->occ_force == 3) : three;
    a->occ3_split = r->occ_force_split ? (r->occ_force_split == 3) : three;
    a->stats = r->stats_dev;
    a->num_cus = r->num_cus;
}

// FrameView::patch_class for this launch (vr_patch_class_kernel), recomputed only when something it depends on
// has changed: camera, rendering / ray-cast parameters (the jitter seed does not count), frame geometry and
// work queue, skip bitmaps.  A static camera pays it once.
int ensure_patch_classes(vrhip_renderer *r, RaycastLaunch *a)
{
    a->frame.patch_class = nullptr;
    a->frame.set_frames = r->queue_frames;
    const vrhip_rendering_params &rp = a->render;
    if (!r->use_patch_classes || rp.technique != 0 || !a->use_ess || a->instr != 0 || !a->frame.live || !a->skip.near_bits ||
        rp.useGradient || a->frame.env || rp.showEss || rp.imgEss || rp.iteration != 0 || a->frame.lds_stage)
        return VRHIP_OK;
    const uint32_t n_patches = r->queue_n / r->queue_frames;
    std::vector<uint8_t> key;
    auto put = [&key](const void *p, size_t n) { key.insert(key.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
    vrhip_rendering_params rk = rp;
    rk.seed = 0;   // (frames of one camera differ in the seed only; iteration is 0 here)
    put(&a->cam, sizeof a->cam);
    put(&rk, sizeof rk);
    put(&a->raycast, sizeof a->raycast);
    const uint32_t misc[10] = {a->frame.W, a->frame.H, a->frame.gsx, a->frame.gsy, r->queue_version, r->skip_version,
                               a->skip.near_r, n_patches, r->queue_frames, (uint32_t)a->bricks.bw};
    put(misc, sizeof misc);
    if (key != r->patch_class_key) {
        if (n_patches > r->patch_class_cap) {
            VR_HIP(r, hipStreamSynchronize(r->stream));
            if (r->patch_class) VR_HIP(r, hipFree(r->patch_class));
            r->patch_class = nullptr;
            r->patch_class_cap = 0;
            VR_HIP(r, hipMalloc((void **)&r->patch_class, n_patches));
            r->patch_class_cap = n_patches;
        }
        r->patch_class_key.clear();
        VR_HIP(r, vr_launch_patch_classes(*a, n_patches, r->queue_frames, r->patch_class, r->stream));
        r->patch_class_key.swap(key);
        if (getenv("VRHIP_DEBUG")) {   // how many patches skip their ray set-up
            std::vector<uint8_t> h(n_patches);
            VR_HIP(r, hipMemcpyAsync(h.data(), r->patch_class, n_patches, hipMemcpyDeviceToHost, r->stream));
            VR_HIP(r, hipStreamSynchronize(r->stream));
            size_t n1 = 0;
            for (uint8_t v : h) n1 += v;
            fprintf(stderr, "[vrhip] patch classes: %zu of %u patches are background for every jitter\n", n1, n_patches);
        }
    }
    a->frame.patch_class = r->patch_class;
    return VRHIP_OK;
}

int launch_timed(vrhip_renderer *r, const RaycastLaunch &a)
{
    if (a.instr) VR_HIP(r, hipMemsetAsync(r->stats_dev, 0, sizeof(DevStats), r->stream));
    // control words: the block of this set was zeroed by the first kernel of the previous set (memset
    // only for the first set, or after something else has used the block)
    if (!r->ctrl_clean[r->ctrl_sel])
        VR_HIP(r, hipMemsetAsync(a.frame.queue_head, 0, kControlWords * sizeof(uint32_t), r->stream));
    r->ctrl_clean[r->ctrl_sel] = false;
    RaycastLaunch b = a;
    {
        const int rc = ensure_patch_classes(r, &b);   // (before the frame's timing starts: once per camera)
        if (rc) return rc;
    }
    // The frame's events are bound to its launches (hipExtLaunchKernelGGL: the first launch's start, phase 1's end, the
    // last launch's end) -- a RECORDED event is a marker packet of its own between two launches, ~6 us of GPU time
    // each; VRHIP_EVENT_BIND=0 records them (A/B), =1 binds the ends only.  What the launchers did not bind -- no
    // launch at all, an experiment's launcher -- is recorded here.
    bool start_bound = false, stop_bound = false;
    b.bind_events = r->event_bind;
    if (r->frame_timing) {
        b.stop_event = r->ev1;
        b.stop_bound = &stop_bound;
        if (r->event_bind >= 2) {
            b.start_event = r->ev0;
            b.start_bound = &start_bound;
        } else {
            VR_HIP(r, hipEventRecord(r->ev0, r->stream));
            start_bound = true;
        }
    }
    b.mid_event = r->phase_timing && r->frame_timing ? r->evm : nullptr;
    r->phase_timed = r->phase_timing && r->frame_timing;
    if (b.frame.hit_in) {
        VR_HIP(r, hipMemsetAsync(r->hit_any, 0, (size_t)r->hit_w * r->hit_h, r->stream));
        b.hit_out = r->hit_out;
    }
    std::memset(&r->last_info, 0, sizeof r->last_info);
    r->last_info.frames = b.frame.seeds ? r->queue_frames : 1u;
    b.info = &r->last_info;
    r->have_info = false;
    VR_HIP(r, vr_launch_frame(b, r->stream));
    r->have_info = true;
    // the first kernel of this set has zeroed the other block (VR_ZERO_NEXT_CTRL) -- unless the set was empty and
    // nothing ran: then the other block keeps whatever it held, and the next set clears it with a memset
    r->ctrl_sel ^= 1u;
    r->ctrl_clean[r->ctrl_sel] = b.frame.n_wave_tiles != 0;
    if (r->frame_timing && !start_bound) VR_HIP(r, hipEventRecord(r->ev0, r->stream));
    if (r->frame_timing && !stop_bound) VR_HIP(r, hipEventRecord(r->ev1, r->stream));
    r->timed = r->frame_timing;
    if (b.frame.hit_in) std::swap(r->hit_in, r->hit_out);   // runRaycast, volumerendercl.cpp:524-530
    return VRHIP_OK;
}

int prepare_render(vrhip_renderer *r, uint32_t width, uint32_t height, uint32_t tile_w,
                   uint32_t tile_h, const uint32_t *tile_ids, uint32_t n_tiles, uint32_t n_frames = 1,
                   uint32_t frame_stride = 0)
{
    int rc = check_renderable(r, width, height);
    if (rc) return rc;
    if (tile_ids) {
        VR_REQUIRE(r, tile_w && tile_h && tile_w % 16 == 0 && tile_h % 16 == 0, VRHIP_ERR_INVALID,
                   "Tile size must be a positive multiple of 16.");
        const uint32_t nt = ((width + tile_w - 1) / tile_w) * ((height + tile_h - 1) / tile_h);
        for (uint32_t i = 0; i < n_tiles; ++i)
            VR_REQUIRE(r, tile_ids[i] < nt, VRHIP_ERR_INVALID, "Tile id out of range.");
    }
    rc = ensure_fb(r, width, height);
    if (rc) return rc;
    r->fp_active = false;
    if (r->render.imgEss) {
        rc = ensure_hit_images(r, width, height);
        if (rc) return rc;
    }
    if (r->render.technique == 0) {
        rc = ensure_skipmap(r);
        if (rc) return rc;
        rc = ensure_footprint(r);
        if (rc) return rc;
    }
    if (r->render.technique == 1 ? r->pt_cull : ray_skip_empty(r)) {
        rc = ensure_cells(r, r->render.technique == 1, r->render.technique != 1);
        if (rc) return rc;
    }
    return ensure_queue(r, width, height, tile_w, tile_h, tile_ids, n_tiles, n_frames, frame_stride);
}

int count_touched_impl(vrhip_renderer *r, uint32_t width, uint32_t height, uint32_t tile_w,
                       uint32_t tile_h, const uint32_t *tile_ids, uint32_t n_tiles,
                       uint64_t *microbricks_touched, uint8_t *bitmap_host, size_t bitmap_bytes,
                       bool fetched_only = false)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    int rc = prepare_render(r, width, height, tile_w, tile_h, tile_ids, n_tiles);
    if (rc) return rc;
    const size_t mb = (size_t)((r->res[0] + 3) / 4) * ((r->res[1] + 3) / 4) * ((r->res[2] + 3) / 4);
    const size_t words = (mb + 31) / 32;
    VR_REQUIRE(r, !bitmap_host || bitmap_bytes == (mb + 7) / 8, VRHIP_ERR_INVALID,
               "vrhip_count_touched: bitmap size mismatch");
    uint32_t *bits = nullptr;
    VR_HIP(r, hipMalloc((void **)&bits, words * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(bits, 0, words * sizeof(uint32_t), r->stream);
    RaycastLaunch a;
    fill_launch(r, width, height, tile_ids ? tile_w : width, &a);
    a.instr = fetched_only ? 3 : 2;
    a.touched = bits;
    // the instrumented pass must not disturb the accumulate buffer: render into a scratch
    float4 *scratch = nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&scratch, (size_t)width * height * sizeof(float4));
    if (e == hipSuccess && r->render.iteration != 0)
        e = hipMemcpyAsync(scratch, r->fb, (size_t)width * height * sizeof(float4),
                           hipMemcpyDeviceToDevice, r->stream);
    a.frame.fb = scratch;
    if (e == hipSuccess) e = hipMemsetAsync(r->stats_dev, 0, sizeof(DevStats), r->stream);
    a.frame.round_budget = 0;   // the traffic pass runs single-phase (no speculative touches)
    // image-order ESS: the pass skips what the next frame will skip, and leaves the hit images alone
    if (a.frame.hit_in && e == hipSuccess)
        e = hipMemsetAsync(r->hit_any, 0, (size_t)r->hit_w * r->hit_h, r->stream);
    a.frame.next_ctrl = nullptr;   // (this pass does not take part in the alternation of the control blocks)
    if (e == hipSuccess) e = hipMemsetAsync(a.frame.queue_head, 0, kControlWords * sizeof(uint32_t), r->stream);
    r->ctrl_clean[r->ctrl_sel] = false;
    if (e == hipSuccess) e = vr_launch_frame(a, r->stream);
    std::vector<uint32_t> host(words);
    if (e == hipSuccess)
        e = hipMemcpyAsync(host.data(), bits, words * sizeof(uint32_t), hipMemcpyDeviceToHost,
                           r->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    (void)hipFree(bits);
    if (scratch) (void)hipFree(scratch);
    if (e != hipSuccess)
        return fail(r, VRHIP_ERR_HIP,
                    std::string("ERROR: vrhip_count_touched (") + hipGetErrorString(e) + ")");
    uint64_t cnt = 0;
    for (size_t i = 0; i < words; ++i) cnt += (uint64_t)__builtin_popcount(host[i]);
    if (microbricks_touched) *microbricks_touched = cnt;
    if (bitmap_host) std::memcpy(bitmap_host, host.data(), bitmap_bytes);
    return VRHIP_OK;
}

__global__ __launch_bounds__(256) void vr_assemble_kernel(const float4 *staging, const uint32_t *slot_of_tile,
                                                          uint32_t W, uint32_t H, uint32_t tw, uint32_t th,
                                                          uint32_t tiles_x, float4 *frame)
{
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const uint32_t slot = slot_of_tile[(y / th) * tiles_x + x / tw];
    frame[(size_t)y * W + x] = staging[((size_t)slot * th + (y % th)) * tw + (x % tw)];
}

// Multi-GPU, rank 0: the frames of a batch straight from the ranks' (sparse) gather messages, one thread per
// pixel -- tiles of one colour from their single pixel, the others from the message's whole tiles
// (TileDriver, tiles.py: message = [maxc slot numbers | S pixels | maxc whole tiles], S = frames x cap).
constexpr uint32_t kMaxGatherRanks = 64;
struct GatherMsgs { const float *p[kMaxGatherRanks]; };

__global__ __launch_bounds__(256) void vr_assemble_batch_kernel(GatherMsgs msgs, const int32_t *pos, const uint32_t *rank_slot,
                                                                uint32_t S, uint32_t cap, uint32_t maxc, uint32_t W, uint32_t H,
                                                                uint32_t tw, uint32_t th, uint32_t tiles_x, float4 *frames)
{
    // one workgroup per (tile, frame): the tile's rank, slot and position are looked up once, and a thread's
    // column inside the tile is fixed where the tile's width divides the workgroup (16 .. 256 pixels)
    const uint32_t t = blockIdx.x, f = blockIdx.y;
    const uint32_t rs = rank_slot[t];
    const uint32_t rank = rs >> 16, row = f * cap + (rs & 0xffffu);
    const float *m = msgs.p[rank];
    const int32_t p = pos[(size_t)rank * S + row];
    const uint32_t x0 = (t % tiles_x) * tw, y0 = (t / tiles_x) * th;
    float4 *dst = frames + ((size_t)f * H + y0) * W + x0;
    const uint32_t w = min(tw, W - x0), h = min(th, H - y0);   // (ragged right / bottom tiles)
    if (p < 0) {
        const float4 v = reinterpret_cast<const float4 *>(m + maxc)[row];
        if (256u % tw == 0u) {
            const uint32_t lx = threadIdx.x % tw;
            if (lx < w)
                for (uint32_t ly = threadIdx.x / tw; ly < h; ly += 256u / tw) dst[(size_t)ly * W + lx] = v;
        } else {
            for (uint32_t i = threadIdx.x; i < tw * th; i += 256u) {
                const uint32_t ly = i / tw, lx = i - ly * tw;
                if (lx < w && ly < h) dst[(size_t)ly * W + lx] = v;
            }
        }
        return;
    }
    const float4 *src = reinterpret_cast<const float4 *>(m + maxc + 4u * (size_t)S) + (size_t)p * th * tw;
    if (256u % tw == 0u) {
        const uint32_t lx = threadIdx.x % tw;
        if (lx < w)
            for (uint32_t ly = threadIdx.x / tw; ly < h; ly += 256u / tw) dst[(size_t)ly * W + lx] = src[ly * tw + lx];
    } else {
        for (uint32_t i = threadIdx.x; i < tw * th; i += 256u) {
            const uint32_t ly = i / tw, lx = i - ly * tw;
            if (lx < w && ly < h) dst[(size_t)ly * W + lx] = src[i];
        }
    }
}

// ---- the sparse gather message of a batch, packed on the GPU (vrhip_pack_tiles; the C++ host's TileGather):
// [spad slot numbers | one pixel per slot | the whole tiles], spad = n_slots rounded up to 4 -- the format
// vr_assemble_batch_kernel reads with maxc = spad (tiles.py packs the same with torch ops and maxc = the ranks'
// largest count).

// one workgroup per slot: is any pixel of the tile different (bit for bit) from its first?  Also the slot's pixel.
__global__ __launch_bounds__(256) void vr_pack_flags_kernel(const uint4 *tiles, uint32_t P, int32_t *flags, uint4 *uni)
{
    const uint32_t s = blockIdx.x;
    const uint4 *t = tiles + (size_t)s * P;
    const uint4 first = t[0];
    bool diff = false;
    for (uint32_t i = threadIdx.x; i < P; i += 256u) {
        const uint4 v = t[i];
        diff = diff || v.x != first.x || v.y != first.y || v.z != first.z || v.w != first.w;
    }
    __shared__ uint32_t any;
    if (threadIdx.x == 0) any = 0u;
    __syncthreads();
    if (__ballot(diff) && (threadIdx.x & 63u) == 0u) atomicOr(&any, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        flags[s] = any ? 1 : 0;
        uni[s] = first;
    }
}

// one workgroup: flags -> position among the whole tiles (or -1), the slot list and the count
__global__ __launch_bounds__(1024) void vr_pack_scan_kernel(int32_t *flags_pos, uint32_t S, int32_t *slots, uint32_t *count)
{
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < S; base += 1024u) {
        const uint32_t s = base + threadIdx.x;
        const bool f = s < S && flags_pos[s] != 0;
        const unsigned long long m = __ballot(f);
        const uint32_t below = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_sums[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t off = carry;
        for (uint32_t w = 0; w < wave; ++w) off += wave_sums[w];
        if (s < S) {
            flags_pos[s] = f ? (int32_t)(off + below) : -1;
            if (f) slots[off + below] = (int32_t)s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (uint32_t w = 0; w < 16u; ++w) t += wave_sums[w];
            carry += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = carry;
}

// one workgroup per slot: a whole tile to its place in the message
__global__ __launch_bounds__(256) void vr_pack_copy_kernel(const uint4 *tiles, uint32_t P, const int32_t *pos, uint4 *out)
{
    const int32_t p = pos[blockIdx.x];
    if (p < 0) return;
    const uint4 *t = tiles + (size_t)blockIdx.x * P;
    uint4 *o = out + (size_t)p * P;
    for (uint32_t i = threadIdx.x; i < P; i += 256u) o[i] = t[i];
}

// root: pos[rank][row] = -1, then the position of every listed slot
__global__ __launch_bounds__(256) void vr_msg_pos_fill_kernel(int32_t *pos, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) pos[i] = -1;
}
struct GatherCounts { uint32_t c[kMaxGatherRanks]; };
__global__ __launch_bounds__(256) void vr_msg_pos_scatter_kernel(GatherMsgs msgs, GatherCounts counts, uint32_t S, int32_t *pos)
{
    const uint32_t rank = blockIdx.y, i = blockIdx.x * 256u + threadIdx.x;
    if (i >= counts.c[rank]) return;
    const uint32_t slot = (uint32_t)reinterpret_cast<const int32_t *>(msgs.p[rank])[i];
    if (slot < S) pos[(size_t)rank * S + slot] = (int32_t)i;
}

} // namespace

extern "C" {

int vrhip_abi_version(void) { return VRHIP_ABI_VERSION; }

int vrhip_pack_tiles(vrhip_renderer *r, void *hip_stream, const float *tiles_dev, uint32_t n_slots, uint32_t tile_pixels,
                     int32_t *scratch_dev, float *msg_dev, uint32_t *count_dev)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, tiles_dev && scratch_dev && msg_dev && count_dev && n_slots && tile_pixels &&
                      ((uintptr_t)tiles_dev & 15u) == 0 && ((uintptr_t)msg_dev & 15u) == 0,
               VRHIP_ERR_INVALID, "vrhip_pack_tiles: invalid argument (buffers must be 16-byte aligned)");
    if (set_device(r)) return VRHIP_ERR_HIP;
    hipStream_t st = (hipStream_t)hip_stream;
    const uint32_t spad = (n_slots + 3u) / 4u * 4u;
    hipLaunchKernelGGL(vr_pack_flags_kernel, dim3(n_slots), dim3(256), 0, st, (const uint4 *)tiles_dev, tile_pixels,
                       scratch_dev, (uint4 *)(msg_dev + spad));
    hipLaunchKernelGGL(vr_pack_scan_kernel, dim3(1), dim3(1024), 0, st, scratch_dev, n_slots, (int32_t *)msg_dev, count_dev);
    hipLaunchKernelGGL(vr_pack_copy_kernel, dim3(n_slots), dim3(256), 0, st, (const uint4 *)tiles_dev, tile_pixels,
                       (const int32_t *)scratch_dev, (uint4 *)(msg_dev + spad + 4u * (size_t)n_slots));
    VR_HIP(r, hipGetLastError());
    return VRHIP_OK;
}

int vrhip_message_positions(vrhip_renderer *r, void *hip_stream, const float *const *msgs_dev, const uint32_t *counts_host,
                            uint32_t world, uint32_t n_slots, int32_t *pos_dev)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, msgs_dev && counts_host && pos_dev && world >= 1 && world <= kMaxGatherRanks && n_slots,
               VRHIP_ERR_INVALID, "vrhip_message_positions: invalid argument");
    if (set_device(r)) return VRHIP_ERR_HIP;
    hipStream_t st = (hipStream_t)hip_stream;
    GatherMsgs g;
    GatherCounts c;
    uint32_t maxc = 0;
    for (uint32_t i = 0; i < kMaxGatherRanks; ++i) {
        g.p[i] = i < world ? msgs_dev[i] : nullptr;
        c.c[i] = i < world ? counts_host[i] : 0u;
        VR_REQUIRE(r, c.c[i] <= n_slots && (i >= world || g.p[i]), VRHIP_ERR_INVALID,
                   "vrhip_message_positions: a count exceeds the number of slots, or a message is NULL");
        maxc = std::max(maxc, c.c[i]);
    }
    const uint32_t n = world * n_slots;
    hipLaunchKernelGGL(vr_msg_pos_fill_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, pos_dev, n);
    if (maxc)
        hipLaunchKernelGGL(vr_msg_pos_scatter_kernel, dim3((maxc + 255u) / 256u, world), dim3(256), 0, st, g, c, n_slots, pos_dev);
    VR_HIP(r, hipGetLastError());
    return VRHIP_OK;
}

int vrhip_create(int device_id, vrhip_renderer **out)
{
    if (!out) return fail(nullptr, VRHIP_ERR_INVALID, "vrhip_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, VRHIP_ERR_HIP,
                    std::string("ERROR: no HIP device available (") + hipGetErrorString(e) + ")");
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, VRHIP_ERR_INVALID, "vrhip_create: device id out of range");
    vrhip_renderer *r = new vrhip_renderer();
    r->device = device_id;
    hipDeviceProp_t prop;
    // the frame's timing events order nothing for the host (results leave by stream-ordered copies that fence for
    // themselves): no system-scope fence -- a write-back of the L2 -- when one of them completes.  VRHIP_EVENT_FENCE=1: A/B
    const unsigned ev_flags = getenv("VRHIP_EVENT_FENCE") && atoi(getenv("VRHIP_EVENT_FENCE")) ? hipEventDefault
                                                                                                 : hipEventDisableSystemFence;
    if ((e = hipSetDevice(device_id)) != hipSuccess ||
        (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&r->ev0, ev_flags)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&r->ev1, ev_flags)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&r->evm, ev_flags)) != hipSuccess ||
        (e = hipEventCreate(&r->evb0)) != hipSuccess ||
        (e = hipEventCreate(&r->evb1)) != hipSuccess ||
        (e = hipMalloc((void **)&r->stats_dev, sizeof(DevStats))) != hipSuccess ||
        (e = hipMalloc((void **)&r->queue_head, 2 * kControlWords * sizeof(uint32_t))) != hipSuccess) {
        std::string msg = std::string("ERROR: vrhip_create (") + hipGetErrorString(e) + ")";
        delete r;
        return fail(nullptr, VRHIP_ERR_HIP, msg);
    }
    r->stream = r->own_stream;
    // the opt-in experiment kernels exist in A/B builds only (tools/mkvariant.sh NAME -DVR_EXPERIMENTS): asking
    // the product library for them fails loudly instead of silently rendering with the default kernels
    if (!vr_experiments_built() && (getenv("VRHIP_MARCH") || getenv("VRHIP_LDS_STAGE") || getenv("VRHIP_MARCH_MICRO"))) {
        vrhip_destroy(r);
        return fail(nullptr, VRHIP_ERR_UNSUPPORTED,
                    "VRHIP_MARCH / VRHIP_LDS_STAGE / VRHIP_MARCH_MICRO need a library built with -DVR_EXPERIMENTS "
                    "(tools/mkvariant.sh experiments -DVR_EXPERIMENTS; VRHIP_LIB_PATH selects it)");
    }
    if (const char *b = getenv("VRHIP_ROUND_BUDGET")) r->round_budget = (uint32_t)atoi(b);   // tuning
    if (getenv("VRHIP_NO_PREPASS")) r->prepass = false;        // experiments: phase 1 walks every patch
    if (getenv("VRHIP_NO_RAYLIST")) r->ray_list = false;       // experiments: phase 1 on live patches
    if (getenv("VRHIP_MARCH")) r->march = true;                // experiments / A-B: the decoupled march kernel
    if (const char *e = getenv("VRHIP_LDS_STAGE")) r->lds_stage = (uint32_t)atoi(e);
    if (const char *e = getenv("VRHIP_CULL_RADIUS")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 64) r->cull_radius = (uint32_t)v;
    }
    if (const char *e = getenv("VRHIP_MARCH_MICRO")) r->march_micro = (uint32_t)atoi(e);
    if (const char *e = getenv("VRHIP_MARCH_FILL")) r->march_fill = (uint32_t)atoi(e);
    if (getenv("VRHIP_NO_SORT")) r->sort_cont = false;         // experiments: phase 2 in append order
    if (getenv("VRHIP_NO_PATCH_CLASS")) r->use_patch_classes = false;   // A/B: every patch sets up its rays
    auto occ_env = [](const char *name, int dflt) {
        const char *e = getenv(name);
        return e ? (atoi(e) == 3 ? 3 : atoi(e) == 2 ? 2 : 0) : dflt;
    };
    if (const char *e = getenv("VRHIP_FRAME_TIMING")) r->frame_timing = atoi(e) != 0;   // A/B (tools/ab_env.sh)
    if (const char *e = getenv("VRHIP_EVENT_BIND")) r->event_bind = atoi(e);
    r->occ_force = r->occ_force_split = occ_env("VRHIP_OCC", 0);
    r->occ_force = occ_env("VRHIP_OCC_P1", r->occ_force);
    r->occ_force_split = occ_env("VRHIP_OCC_P2", r->occ_force_split);
    if (getenv("VRHIP_PT_NO_CULL")) r->pt_cull = false;        // experiments: no opacity-bound culling
    if (getenv("VRHIP_PT_NO_LEAP")) r->pt_leap = false;        // A/B: every tracking step on its own
    if (getenv("VRHIP_PT_NO_FAR_LEAP")) r->pt_leap_far = false;
    if (getenv("VRHIP_NO_EMPTY_SKIP")) r->skip_empty = false;  // experiments: no empty-run skipping
    if (getenv("VRHIP_EMPTY_SKIP")) r->skip_empty_force = true;   // ... or everywhere
    if (getenv("VRHIP_NO_FOOTPRINT")) r->use_fp = false;       // plain volume layout only
    if (const char *e = getenv("VRHIP_REFILL_MIN")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 16) r->refill_min = (uint32_t)v;
    }
    if (const char *e = getenv("VRHIP_FOOTPRINT_MAX_GB")) {
        const double gb = atof(e);
        if (gb >= 0.0) r->fp_max_bytes = (size_t)(gb * 1073741824.0);   // (fp_failed_bytes starts at 0)
    }
    r->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    r->devname = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    *out = r;
    return VRHIP_OK;
}

void vrhip_destroy(vrhip_renderer *r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipStreamSynchronize(r->stream);
    vrhip_clear_volumes(r);
    if (r->tff) (void)hipFree(r->tff);
    if (r->prefix) (void)hipFree(r->prefix);
    if (r->fb) (void)hipFree(r->fb);
    if (r->stats_dev) (void)hipFree(r->stats_dev);
    if (r->queue_dev) (void)hipFree(r->queue_dev);
    if (r->queue_head) (void)hipFree(r->queue_head);
    if (r->cont) (void)hipFree(r->cont);
    if (r->cost) (void)hipFree(r->cost);
    for (uint8_t *p : {r->hit_in, r->hit_out, r->hit_status, r->hit_any})
        if (p) (void)hipFree(p);
    if (r->env) (void)hipFree(r->env);
    if (r->seeds_dev) (void)hipFree(r->seeds_dev);
    if (r->fp) (void)hipFree(r->fp);
    if (r->live) (void)hipFree(r->live);
    if (r->order) (void)hipFree(r->order);
    if (r->live_rays) (void)hipFree(r->live_rays);
    if (r->skip_bits) (void)hipFree(r->skip_bits);
    if (r->near_bits) (void)hipFree(r->near_bits);
    if (r->near_scratch) (void)hipFree(r->near_scratch);
    if (r->patch_class) (void)hipFree(r->patch_class);
    if (r->cell_bound) (void)hipFree(r->cell_bound);
    if (r->cell_dist) (void)hipFree(r->cell_dist);
    if (r->cell_empty) (void)hipFree(r->cell_empty);
    if (r->cell_sparse) (void)hipFree(r->cell_sparse);
    if (r->cell_bmask) (void)hipFree(r->cell_bmask);
    if (r->ev0) (void)hipEventDestroy(r->ev0);
    if (r->ev1) (void)hipEventDestroy(r->ev1);
    if (r->evm) (void)hipEventDestroy(r->evm);
    if (r->evb0) (void)hipEventDestroy(r->evb0);
    if (r->evb1) (void)hipEventDestroy(r->evb1);
    if (r->own_stream) (void)hipStreamDestroy(r->own_stream);
    delete r;
}

const char *vrhip_last_error(const vrhip_renderer *r)
{
    return r ? r->err.c_str() : g_create_error.c_str();
}

int vrhip_device_name(const vrhip_renderer *r, char *buf, size_t buf_len)
{
    if (!r || !buf || !buf_len) return VRHIP_ERR_INVALID;
    std::snprintf(buf, buf_len, "%s", r->devname.c_str());
    return VRHIP_OK;
}

int vrhip_set_stream(vrhip_renderer *r, void *hip_stream, int use_own)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_HIP(r, hipStreamSynchronize(r->stream));
    r->stream = use_own ? r->own_stream : (hipStream_t)hip_stream;
    return VRHIP_OK;
}

static int download_cells_impl(vrhip_renderer *r, bool fine, float *out_minmax, size_t n_floats, uint32_t dims[3],
                               uint32_t *shift)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, !r->vols.empty() && r->timestep < r->vols.size() && r->vols[r->timestep].dev,
               VRHIP_ERR_NODATA, "No volume data is loaded.");
    VR_REQUIRE(r, r->tff && r->tff_n, VRHIP_ERR_NODATA, "No transfer function set.");
    if (set_device(r)) return VRHIP_ERR_HIP;
    r->pt_dirty = true;
    // (only the grid asked for: the coarse one is then built directly unless the fine one exists already,
    // in which case it is reduced from it -- the tests take both routes)
    int rc = ensure_cells(r, !fine, fine);
    if (rc) return rc;
    const CellView &g = r->cells;
    const bool second = fine && g.eshift != g.shift;
    const int cx = second ? g.ecx : g.cx, cy = second ? g.ecy : g.cy, cz = second ? g.ecz : g.cz;
    const size_t n_cells = (size_t)cx * cy * cz;
    if (dims) { dims[0] = (uint32_t)cx; dims[1] = (uint32_t)cy; dims[2] = (uint32_t)cz; }
    if (shift) *shift = (uint32_t)(second ? g.eshift : g.shift);
    if (!out_minmax) return VRHIP_OK;
    VR_REQUIRE(r, n_floats == 2 * n_cells, VRHIP_ERR_INVALID, "vrhip_download_cells: size mismatch");
    VR_HIP(r, hipStreamSynchronize(r->stream));
    const VolumeSlot &s = r->vols[r->timestep];
    const float2 *src = second ? (s.fine_minmax_valid ? s.fine_minmax : nullptr) : (s.pt_minmax_valid ? s.pt_minmax : nullptr);
    if (!src && r->vol_owner && r->timestep < r->vol_owner->vols.size()) {   // tables read from the volumes' owner
        const VolumeSlot &os = r->vol_owner->vols[r->timestep];
        src = second ? (os.fine_minmax_valid ? os.fine_minmax : nullptr) : (os.pt_minmax_valid ? os.pt_minmax : nullptr);
    }
    VR_REQUIRE(r, src, VRHIP_ERR_NODATA, "vrhip_download_cells: no cell grid");
    VR_HIP(r, hipMemcpy(out_minmax, src, n_cells * sizeof(float2), hipMemcpyDeviceToHost));
    return VRHIP_OK;
}

int vrhip_download_cells(vrhip_renderer *r, float *out_minmax, size_t n_floats, uint32_t dims[3],
                         uint32_t *shift)
{
    return download_cells_impl(r, false, out_minmax, n_floats, dims, shift);
}

int vrhip_download_empty_cells(vrhip_renderer *r, float *out_minmax, size_t n_floats, uint32_t dims[3],
                               uint32_t *shift)
{
    return download_cells_impl(r, true, out_minmax, n_floats, dims, shift);
}

int vrhip_assemble_frame(vrhip_renderer *r, const float *staging_dev, const uint32_t *slot_of_tile_dev,
                         uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                         float *frame_dev)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, staging_dev && slot_of_tile_dev && frame_dev && width && height && tile_w && tile_h,
               VRHIP_ERR_INVALID, "vrhip_assemble_frame: invalid argument");
    if (set_device(r)) return VRHIP_ERR_HIP;
    const uint32_t tiles_x = (width + tile_w - 1) / tile_w;
    hipLaunchKernelGGL(vr_assemble_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, r->stream,
                       (const float4 *)staging_dev, slot_of_tile_dev, width, height, tile_w, tile_h, tiles_x,
                       (float4 *)frame_dev);
    VR_HIP(r, hipGetLastError());
    return VRHIP_OK;
}

int vrhip_assemble_batch(vrhip_renderer *r, void *hip_stream, const float *const *msgs_dev, uint32_t world,
                         uint32_t n_frames, uint32_t cap, uint32_t maxc, const int32_t *pos_dev,
                         const uint32_t *rank_slot_of_tile_dev, uint32_t width, uint32_t height, uint32_t tile_w,
                         uint32_t tile_h, float *frames_dev)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, msgs_dev && pos_dev && rank_slot_of_tile_dev && frames_dev && world >= 1 && world <= kMaxGatherRanks &&
                      n_frames && cap && cap <= 65536u && width && height && tile_w && tile_h && n_frames <= 65535u,
               VRHIP_ERR_INVALID, "vrhip_assemble_batch: invalid argument");
    if (set_device(r)) return VRHIP_ERR_HIP;
    GatherMsgs g;
    for (uint32_t i = 0; i < kMaxGatherRanks; ++i) g.p[i] = i < world ? msgs_dev[i] : nullptr;
    for (uint32_t i = 0; i < world; ++i)
        VR_REQUIRE(r, g.p[i] && ((uintptr_t)g.p[i] & 15u) == 0 && maxc % 4u == 0, VRHIP_ERR_INVALID,
                   "vrhip_assemble_batch: messages must be 16-byte aligned, maxc a multiple of 4");
    const uint32_t tiles_x = (width + tile_w - 1) / tile_w;
    const uint32_t tiles_y = (height + tile_h - 1) / tile_h;
    hipLaunchKernelGGL(vr_assemble_batch_kernel, dim3(tiles_x * tiles_y, n_frames), dim3(256), 0,
                       (hipStream_t)hip_stream, g, pos_dev, rank_slot_of_tile_dev, n_frames * cap, cap, maxc, width,
                       height, tile_w, tile_h, tiles_x, (float4 *)frames_dev);
    VR_HIP(r, hipGetLastError());
    return VRHIP_OK;
}

int vrhip_get_stream(const vrhip_renderer *r, void **hip_stream)
{
    if (!r || !hip_stream) return VRHIP_ERR_INVALID;
    *hip_stream = (void *)r->stream;
    return VRHIP_OK;
}

int vrhip_upload_volume(vrhip_renderer *r, const void *host_voxels, const uint32_t res[3],
                        int format, uint32_t timestep)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, host_voxels, VRHIP_ERR_INVALID, "vrhip_upload_volume: NULL voxel pointer");
    if (set_device(r)) return VRHIP_ERR_HIP;
    VolumeSlot *s;
    int rc = prepare_slot(r, res, format, timestep, &s);
    if (rc) return rc;
    VR_HIP(r, copy_volume(r, s->dev, const_cast<void *>(host_voxels), true, false, r->stream));
    return VRHIP_OK;
}

int vrhip_upload_volume_channels(vrhip_renderer *r, const void *host_voxels, const uint32_t res[3],
                                 int format, int channels, uint32_t timestep)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (channels == 1) return vrhip_upload_volume(r, host_voxels, res, format, timestep);
    VR_REQUIRE(r, host_voxels, VRHIP_ERR_INVALID, "vrhip_upload_volume: NULL voxel pointer");
    VR_REQUIRE(r, channels == 2 || channels == 4, VRHIP_ERR_INVALID,
               "Unknown or invalid volume color format.");   // volumerendercl.cpp:711
    if (set_device(r)) return VRHIP_ERR_HIP;
    VolumeSlot *s;
    int rc = prepare_slot(r, res, format, timestep, &s, channels);
    if (rc) return rc;
    // interleaved texels -> one planar array per channel, each re-tiled like a CL_R volume
    const size_t n = (size_t)res[0] * res[1] * res[2], bpv = fmt_bytes(format);
    std::vector<unsigned char> plane(n * bpv);
    const unsigned char *src = static_cast<const unsigned char *>(host_voxels);
    for (int c = 0; c < channels; ++c) {
        for (size_t i = 0; i < n; ++i)
            std::memcpy(&plane[i * bpv], src + (i * (size_t)channels + (size_t)c) * bpv, bpv);
        VR_HIP(r, copy_volume(r, c == 0 ? s->dev : s->chan[c - 1], plane.data(), true, false,
                              r->stream));
    }
    return VRHIP_OK;
}

int vrhip_upload_volume_device(vrhip_renderer *r, const void *dev_voxels, const uint32_t res[3],
                               int format, uint32_t timestep)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, dev_voxels, VRHIP_ERR_INVALID, "vrhip_upload_volume_device: NULL pointer");
    if (set_device(r)) return VRHIP_ERR_HIP;
    VolumeSlot *s;
    int rc = prepare_slot(r, res, format, timestep, &s);
    if (rc) return rc;
    VR_HIP(r, copy_volume(r, s->dev, const_cast<void *>(dev_voxels), true, true, r->stream));
    return VRHIP_OK;
}

int vrhip_synth_volume(vrhip_renderer *r, int kind, const uint32_t res[3], int format,
                       uint32_t timestep)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, kind == 0 || kind == 1, VRHIP_ERR_INVALID, "vrhip_synth_volume: kind must be 0/1");
    if (set_device(r)) return VRHIP_ERR_HIP;
    VolumeSlot *s;
    int rc = prepare_slot(r, res, format, timestep, &s);
    if (rc) return rc;
    VR_HIP(r, vr_launch_synth(kind, make_vol_view(r, s->dev), format, r->stream));
    VR_HIP(r, hipStreamSynchronize(r->stream));
    return VRHIP_OK;
}

int vrhip_download_volume(vrhip_renderer *r, uint32_t timestep, void *host_dst, size_t bytes)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, timestep < r->vols.size() && r->vols[timestep].dev, VRHIP_ERR_NODATA,
               "No volume data is loaded.");
    VR_REQUIRE(r, host_dst && bytes == volume_bytes(r), VRHIP_ERR_INVALID,
               "vrhip_download_volume: size mismatch");
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_HIP(r, copy_volume(r, r->vols[timestep].dev, host_dst, false, false, r->stream));
    return VRHIP_OK;
}

int vrhip_downsample_volume(vrhip_renderer *r, uint32_t timestep, int factor, void *host_dst,
                            size_t bytes, uint32_t out_res[3])
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, timestep < r->vols.size() && r->vols[timestep].dev, VRHIP_ERR_NODATA,
               "No volume data is loaded.");
    VR_REQUIRE(r, factor >= 2, VRHIP_ERR_INVALID, "Factor must be greater or equal 2.");
    int lo[3], vpc[3];
    for (int i = 0; i < 3; ++i) {
        lo[i] = (int)std::ceil((double)r->res[i] / (double)factor);        // volumerendercl.cpp:245-251
        vpc[i] = (int)std::ceil((float)r->res[i] / (float)lo[i]);           // volumeraycast.cl:974-975
        if (out_res) out_res[i] = (uint32_t)lo[i];
    }
    VR_REQUIRE(r, lo[0] >= 64, VRHIP_ERR_INVALID,
               "Could not create down-sampled volume data set, because the resolution would be "
               "smaller than the minimum (64x64x64).");   // :253-259
    if (!host_dst) return VRHIP_OK;
    const size_t need = (size_t)lo[0] * lo[1] * lo[2] * fmt_bytes(r->format);
    VR_REQUIRE(r, bytes == need, VRHIP_ERR_INVALID, "vrhip_downsample_volume: size mismatch");
    if (set_device(r)) return VRHIP_ERR_HIP;
    void *dev = nullptr;
    VR_HIP(r, hipMalloc(&dev, need));
    hipError_t e = vr_launch_downsample(make_vol_view(r, r->vols[timestep].dev), r->format, lo, vpc,
                                        dev, r->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(host_dst, dev, need, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    (void)hipFree(dev);
    if (e != hipSuccess)
        return fail(r, VRHIP_ERR_HIP,
                    std::string("ERROR: vrhip_downsample_volume (") + hipGetErrorString(e) + ")");
    return VRHIP_OK;
}

int vrhip_clear_volumes(vrhip_renderer *r)
{
    if (!r) return VRHIP_ERR_INVALID;
    (void)hipSetDevice(r->device);
    (void)hipStreamSynchronize(r->stream);
    // renderers that render from these voxels (vrhip_share_volumes) stop doing so before they are freed
    detach_sharers(r);
    for (VolumeSlot &s : r->vols) {
        if (!s.borrowed) {
            if (s.dev) (void)hipFree(s.dev);
            for (void *c : s.chan)
                if (c) (void)hipFree(c);
            if (s.bricks) (void)hipFree(s.bricks);
        }
        if (s.pt_minmax) (void)hipFree(s.pt_minmax);
        if (s.fine_minmax) (void)hipFree(s.fine_minmax);
    }
    r->vols.clear();
    r->fp_valid = false;
    r->fp_failed_bytes = 0;
    if (r->vol_owner) {   // a sharer leaves its owner's list
        std::vector<vrhip_renderer *> &v = r->vol_owner->sharers;
        v.erase(std::remove(v.begin(), v.end(), r), v.end());
        r->vol_owner = nullptr;
    }

    r->bricks_valid = false;
    r->skip_dirty = true;
    r->pt_dirty = true;
    r->format = -1;
    r->channels = 1;
    r->res[0] = r->res[1] = r->res[2] = 0;
    r->timestep = 0;
    return VRHIP_OK;
}

int vrhip_set_round_budget(vrhip_renderer *r, uint32_t rounds)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, rounds <= 100000u, VRHIP_ERR_INVALID, "vrhip_set_round_budget: out of range");
    r->round_budget = rounds;
    return VRHIP_OK;
}

int vrhip_share_volumes(vrhip_renderer *r, vrhip_renderer *owner)
{
    if (!r || !owner || r == owner) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, r->device == owner->device, VRHIP_ERR_INVALID,
               "vrhip_share_volumes: both renderers must live on the same device.");
    VR_REQUIRE(r, !owner->vols.empty(), VRHIP_ERR_NODATA, "No volume data is loaded.");
    for (const VolumeSlot &s : owner->vols)
        VR_REQUIRE(r, s.dev && !s.borrowed, VRHIP_ERR_INVALID,
                   "vrhip_share_volumes: the owner must hold its own, complete volumes.");
    int rc = vrhip_clear_volumes(r);
    if (rc) return rc;
    // what the owner has written must be complete before this renderer's stream reads it
    if (hipStreamSynchronize(owner->stream) != hipSuccess)
        return fail(r, VRHIP_ERR_HIP, "ERROR: vrhip_share_volumes (stream synchronize)");
    std::memcpy(r->res, owner->res, sizeof r->res);
    r->format = owner->format;
    r->channels = owner->channels;
    set_layout(r);
    for (int i = 0; i < 3; ++i) {
        r->brick_edge[i] = owner->brick_edge[i];
        r->brick_res[i] = owner->brick_res[i];
        r->brick_tex[i] = owner->brick_tex[i];
        r->raycast.brickRes[i] = owner->raycast.brickRes[i];
    }
    r->vols.resize(owner->vols.size());
    for (size_t t = 0; t < owner->vols.size(); ++t) {
        VolumeSlot &d = r->vols[t];
        const VolumeSlot &s = owner->vols[t];
        d.dev = s.dev;
        for (int c = 0; c < 3; ++c) d.chan[c] = s.chan[c];
        d.bricks = s.bricks;
        d.borrowed = true;
    }
    r->bricks_valid = owner->bricks_valid;
    r->timestep = owner->timestep < r->vols.size() ? owner->timestep : 0;
    r->fp_valid = false;   // (its own footprint volume is not used while it shares: ensure_footprint)
    r->vol_owner = owner;
    owner->sharers.push_back(r);
    r->skip_dirty = true;
    r->pt_dirty = true;
    return VRHIP_OK;
}

int vrhip_set_timestep(vrhip_renderer *r, uint32_t timestep)
{
    if (!r) return VRHIP_ERR_INVALID;
    // volumerendercl.cpp:1169-1170: silently ignored when out of range
    if (!r->vols.empty() && timestep >= r->vols.size()) return VRHIP_OK;
    if (r->timestep != timestep) {
        r->skip_dirty = r->pt_dirty = true;
        r->fp_valid = false;
    }
    r->timestep = timestep;
    return VRHIP_OK;
}

int vrhip_get_resolution(const vrhip_renderer *r, uint32_t res_xyzt[4])
{
    if (!r || !res_xyzt) return VRHIP_ERR_INVALID;
    res_xyzt[0] = r->res[0];
    res_xyzt[1] = r->res[1];
    res_xyzt[2] = r->res[2];
    res_xyzt[3] = r->vols.empty() ? 1u : (uint32_t)r->vols.size();
    return VRHIP_OK;
}

int vrhip_set_transfer_function(vrhip_renderer *r, const uint8_t *rgba8, uint32_t n_entries)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, rgba8 && n_entries > 0, VRHIP_ERR_INVALID, "Empty transfer function.");
    VR_REQUIRE(r, n_entries <= 4096, VRHIP_ERR_UNSUPPORTED,
               "Transfer functions above 4096 entries are not supported.");
    if (set_device(r)) return VRHIP_ERR_HIP;
    // CL_UNORM_INT8 -> float: c / 255.0f (OpenCL 1.2 spec 8.3.1.1)
    std::vector<float4> table(n_entries);
    for (uint32_t i = 0; i < n_entries; ++i) {
        table[i].x = (float)rgba8[4 * i + 0] / 255.0f;
        table[i].y = (float)rgba8[4 * i + 1] / 255.0f;
        table[i].z = (float)rgba8[4 * i + 2] / 255.0f;
        table[i].w = (float)rgba8[4 * i + 3] / 255.0f;
    }
    VR_HIP(r, hipStreamSynchronize(r->stream));
    if (r->tff_n != n_entries) {
        if (r->tff) VR_HIP(r, hipFree(r->tff));
        r->tff = nullptr;
        r->tff_n = 0;
        VR_HIP(r, hipMalloc((void **)&r->tff, n_entries * sizeof(float4)));
        r->tff_n = n_entries;
    }
    VR_HIP(r, hipMemcpy(r->tff, table.data(), n_entries * sizeof(float4), hipMemcpyHostToDevice));
    r->skip_dirty = true;
    r->pt_dirty = true;
    return VRHIP_OK;
}

int vrhip_set_tff_prefix_sum(vrhip_renderer *r, const uint32_t *prefix, uint32_t n)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, prefix && n > 0, VRHIP_ERR_INVALID, "Empty prefix sum.");
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_HIP(r, hipStreamSynchronize(r->stream));
    if (r->prefix_n != n) {
        if (r->prefix) VR_HIP(r, hipFree(r->prefix));
        r->prefix = nullptr;
        r->prefix_n = 0;
        VR_HIP(r, hipMalloc((void **)&r->prefix, n * sizeof(uint32_t)));
        r->prefix_n = n;
    }
    VR_HIP(r, hipMemcpy(r->prefix, prefix, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    r->skip_dirty = true;
    r->pt_dirty = true;
    return VRHIP_OK;
}

int vrhip_build_bricks(vrhip_renderer *r)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, !r->vols.empty(), VRHIP_ERR_NODATA, "No volume data is loaded.");
    if (set_device(r)) return VRHIP_ERR_HIP;
    // brick size / grid: volumerendercl.cpp:620-636
    for (int i = 0; i < 3; ++i) {
        uint32_t e = round_pow2(r->res[i] / 64u);
        r->brick_edge[i] = e > 1u ? e : 1u;
        r->brick_res[i] = (float)r->res[i] / (float)r->brick_edge[i];
        r->brick_tex[i] = (uint32_t)std::ceil((double)r->brick_res[i]);
        r->raycast.brickRes[i] = r->brick_res[i];   // :630-631
    }
    // The bricks depend on the voxels alone, and renderers created with vrhip_share_volumes keep
    // the pointer: an allocation is made once per slot and lives as long as the voxels do, and a
    // slot whose voxels have not changed since its last build is left alone (the reference
    // rebuilds on every setTransferFunction, :877 -- same values).
    for (size_t t = 0; t < r->vols.size(); ++t) {
        VolumeSlot &s = r->vols[t];
        if (!s.dev) return fail(r, VRHIP_ERR_NODATA,
                                "Error loading timeseries data: size mismatch.");   // :227
        if (s.borrowed) {   // shared voxels come with their bricks (they depend on nothing else)
            const vrhip_renderer *o = r->vol_owner;
            VR_REQUIRE(r, o && t < o->vols.size() && o->vols[t].bricks && o->vols[t].bricks_built, VRHIP_ERR_NODATA,
                       "vrhip_build_bricks: the owner of the shared volumes has to build its bricks first.");
            s.bricks = o->vols[t].bricks;
            continue;
        }
        if (!s.bricks) {
            VR_HIP(r, hipMalloc(&s.bricks, bricks_bytes(r)));
            s.bricks_built = false;
        }
    }
    bool any = false;
    for (const VolumeSlot &s : r->vols) any = any || (!s.borrowed && !s.bricks_built);
    if (any) {   // (vrhip_last_bricks_seconds keeps reporting the last build that did something)
        VR_HIP(r, hipEventRecord(r->evb0, r->stream));
        for (VolumeSlot &s : r->vols) {
            if (s.borrowed || s.bricks_built) continue;
            VolView v = make_vol_view(r, s.dev);
            VR_HIP(r, vr_launch_build_bricks(v, r->format, r->brick_tex, s.bricks, r->stream));
            s.bricks_built = true;
        }
        VR_HIP(r, hipEventRecord(r->evb1, r->stream));
        r->bricks_timed = true;
    }
    VR_HIP(r, hipStreamSynchronize(r->stream));   // reference: _queueCL.finish() (:670)
    r->bricks_valid = true;
    r->skip_dirty = true;
    r->pt_dirty = true;
    return VRHIP_OK;
}

int vrhip_get_brick_info(const vrhip_renderer *r, uint32_t tex[3], float brick_res[3],
                         uint32_t edge[3])
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, r->bricks_valid, VRHIP_ERR_NODATA, "ESS bricks not built.");
    for (int i = 0; i < 3; ++i) {
        if (tex) tex[i] = r->brick_tex[i];
        if (brick_res) brick_res[i] = r->brick_res[i];
        if (edge) edge[i] = r->brick_edge[i];
    }
    return VRHIP_OK;
}

int vrhip_download_bricks(vrhip_renderer *r, uint32_t timestep, void *host_dst, size_t bytes)
{
    if (!r) return VRHIP_ERR_INVALID;
    VR_REQUIRE(r, r->bricks_valid && timestep < r->vols.size() && r->vols[timestep].bricks,
               VRHIP_ERR_NODATA, "ESS bricks not built.");
    VR_REQUIRE(r, host_dst && bytes == bricks_bytes(r), VRHIP_ERR_INVALID,
               "vrhip_download_bricks: size mismatch");
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_HIP(r, hipMemcpyAsync(host_dst, r->vols[timestep].bricks, bytes, hipMemcpyDeviceToHost,
                             r->stream));
    VR_HIP(r, hipStreamSynchronize(r->stream));
    return VRHIP_OK;
}

double vrhip_last_bricks_seconds(const vrhip_renderer *r)
{
    if (!r || !r->bricks_timed) return 0.0;
    float ms = 0.f;
    if (hipEventSynchronize(r->evb1) != hipSuccess) return 0.0;
    if (hipEventElapsedTime(&ms, r->evb0, r->evb1) != hipSuccess) return 0.0;
    return (double)ms * 1e-3;
}

int vrhip_set_camera_params(vrhip_renderer *r, const vrhip_camera_params *p)
{
    if (!r || !p) return VRHIP_ERR_INVALID;
    r->cam = *p;
    return VRHIP_OK;
}

int vrhip_set_rendering_params(vrhip_renderer *r, const vrhip_rendering_params *p)
{
    if (!r || !p) return VRHIP_ERR_INVALID;
    r->render = *p;
    return VRHIP_OK;
}

int vrhip_set_raycast_params(vrhip_renderer *r, const vrhip_raycast_params *p)
{
    if (!r || !p) return VRHIP_ERR_INVALID;
    r->raycast = *p;
    return VRHIP_OK;
}

int vrhip_set_pathtrace_params(vrhip_renderer *r, const vrhip_pathtrace_params *p)
{
    if (!r || !p) return VRHIP_ERR_INVALID;
    r->pathtrace = *p;
    return VRHIP_OK;
}

int vrhip_set_object_ess(vrhip_renderer *r, int enabled)
{
    if (!r) return VRHIP_ERR_INVALID;
    r->use_ess = enabled != 0;
    return VRHIP_OK;
}

int vrhip_render_frame(vrhip_renderer *r, uint32_t width, uint32_t height, float *out_rgba,
                       int out_is_device)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    int rc = prepare_render(r, width, height, 0, 0, nullptr, 0);
    if (rc) return rc;
    RaycastLaunch a;
    fill_launch(r, width, height, width, &a);
    if (out_rgba && out_is_device) a.frame.out = (float4 *)out_rgba;
    rc = launch_timed(r, a);
    if (rc) return rc;
    if (out_rgba && !out_is_device) {
        VR_HIP(r, hipMemcpyAsync(out_rgba, r->fb, (size_t)width * height * sizeof(float4),
                                 hipMemcpyDeviceToHost, r->stream));
        VR_HIP(r, hipStreamSynchronize(r->stream));
    }
    return VRHIP_OK;
}

int vrhip_render_tiles(vrhip_renderer *r, uint32_t width, uint32_t height, uint32_t tile_w,
                       uint32_t tile_h, const uint32_t *tile_ids, uint32_t n_tiles,
                       float *out_tiles_dev)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_REQUIRE(r, out_tiles_dev, VRHIP_ERR_INVALID, "vrhip_render_tiles: NULL output");
    if (n_tiles == 0) return VRHIP_OK;
    VR_REQUIRE(r, tile_ids, VRHIP_ERR_INVALID, "vrhip_render_tiles: NULL tile list");
    int rc = prepare_render(r, width, height, tile_w, tile_h, tile_ids, n_tiles);
    if (rc) return rc;
    RaycastLaunch a;
    fill_launch(r, width, height, tile_w, &a);
    a.frame.out = (float4 *)out_tiles_dev;
    return launch_timed(r, a);
}

int vrhip_render_batch(vrhip_renderer *r, uint32_t width, uint32_t height, uint32_t tile_w,
                       uint32_t tile_h, const uint32_t *tile_ids, uint32_t n_tiles,
                       const uint32_t *seeds, uint32_t n_frames, float *out_dev,
                       uint32_t out_frame_stride)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_REQUIRE(r, out_dev && seeds, VRHIP_ERR_INVALID, "vrhip_render_batch: NULL argument");
    VR_REQUIRE(r, n_frames >= 1 && n_frames <= kMaxBatchFrames, VRHIP_ERR_INVALID,
               "vrhip_render_batch: 1..256 frames per batch");
    VR_REQUIRE(r, height <= (8u << kFrameShift) && width <= (8u << kFrameShift), VRHIP_ERR_INVALID,
               "Invalid output image size.");
    VR_REQUIRE(r, !tile_ids || n_tiles > 0, VRHIP_ERR_INVALID, "vrhip_render_batch: empty tile list");
    // the frames of a batch are independent: nothing that chains frames, nothing that draws from
    // rendering_params.seed outside the ray set-up
    const vrhip_rendering_params &rp = r->render;
    VR_REQUIRE(r, rp.technique == 0 && rp.iteration == 0 && !rp.imgEss && !r->raycast.useAO,
               VRHIP_ERR_UNSUPPORTED,
               "vrhip_render_batch: ray caster only, iteration 0, no image-order ESS, no ambient occlusion");
    const uint32_t packed = tile_ids ? n_tiles * tile_w * tile_h : width * height;
    VR_REQUIRE(r, out_frame_stride == 0 || out_frame_stride >= packed, VRHIP_ERR_INVALID,
               "vrhip_render_batch: frame stride smaller than a frame");
    VR_REQUIRE(r, (unsigned long long)n_frames * (out_frame_stride ? out_frame_stride : packed) <= 0xffffffffull,
               VRHIP_ERR_INVALID, "vrhip_render_batch: the frames of a batch must hold fewer than 2^32 pixels together");
    int rc = prepare_render(r, width, height, tile_w, tile_h, tile_ids, n_tiles, n_frames,
                            out_frame_stride);
    if (rc) return rc;
    if (!r->seeds_dev) VR_HIP(r, hipMalloc((void **)&r->seeds_dev, kMaxBatchFrames * sizeof(uint32_t)));
    VR_HIP(r, hipMemcpyAsync(r->seeds_dev, seeds, n_frames * sizeof(uint32_t), hipMemcpyHostToDevice,
                             r->stream));
    RaycastLaunch a;
    fill_launch(r, width, height, tile_ids ? tile_w : width, &a);
    a.frame.out = (float4 *)out_dev;
    a.frame.seeds = r->seeds_dev;
    return launch_timed(r, a);
}

double vrhip_last_kernel_seconds(const vrhip_renderer *r)
{
    if (!r || !r->timed) return 0.0;
    float ms = 0.f;
    if (hipEventSynchronize(r->ev1) != hipSuccess) return 0.0;
    if (hipEventElapsedTime(&ms, r->ev0, r->ev1) != hipSuccess) return 0.0;
    return (double)ms * 1e-3;
}

int vrhip_last_phase_seconds(const vrhip_renderer *r, double *phase1, double *phase2)
{
    if (!r || !r->timed || !r->phase_timed) return VRHIP_ERR_NODATA;
    float a = 0.f, b = 0.f;
    if (hipEventSynchronize(r->ev1) != hipSuccess) return VRHIP_ERR_HIP;
    if (hipEventElapsedTime(&a, r->ev0, r->evm) != hipSuccess) return VRHIP_ERR_HIP;
    if (hipEventElapsedTime(&b, r->evm, r->ev1) != hipSuccess) return VRHIP_ERR_HIP;
    if (phase1) *phase1 = (double)a * 1e-3;
    if (phase2) *phase2 = (double)b * 1e-3;
    return VRHIP_OK;
}

int vrhip_last_launch_info(const vrhip_renderer *r, vrhip_launch_info *out)
{
    if (!r || !out) return VRHIP_ERR_INVALID;
    if (!r->have_info) return fail(r, VRHIP_ERR_NODATA, "vrhip_last_launch_info: nothing has been rendered yet");
    *out = r->last_info;
    return VRHIP_OK;
}

int vrhip_download_cost_map(vrhip_renderer *r, uint16_t *out, size_t n)
{
    if (!r || !out) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_REQUIRE(r, r->cost && n == (size_t)r->fb_w * r->fb_h, VRHIP_ERR_NODATA,
               "vrhip_download_cost_map: no cost map of this size (render a frame first)");
    VR_HIP(r, hipStreamSynchronize(r->stream));
    VR_HIP(r, hipMemcpy(out, r->cost, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return VRHIP_OK;
}

int vrhip_set_phase_timing(vrhip_renderer *r, int enabled)
{
    if (!r) return VRHIP_ERR_INVALID;
    r->phase_timing = enabled != 0;
    return VRHIP_OK;
}

int vrhip_set_frame_timing(vrhip_renderer *r, int enabled)
{
    if (!r) return VRHIP_ERR_INVALID;
    r->frame_timing = enabled != 0;
    if (!r->frame_timing) r->timed = r->phase_timed = false;
    return VRHIP_OK;
}

int vrhip_set_stats_enabled(vrhip_renderer *r, int enabled)
{
    if (!r) return VRHIP_ERR_INVALID;
    r->stats_enabled = enabled != 0;
    return VRHIP_OK;
}

int vrhip_get_stats(const vrhip_renderer *r, vrhip_stats *out)
{
    if (!r || !out) return VRHIP_ERR_INVALID;
    if (hipSetDevice(r->device) != hipSuccess) return VRHIP_ERR_HIP;
    DevStats s;
    VR_HIP(r, hipStreamSynchronize(r->stream));
    VR_HIP(r, hipMemcpy(&s, r->stats_dev, sizeof s, hipMemcpyDeviceToHost));
    out->samples_taken = s.v[0];
    out->samples_nominal = s.v[1];
    out->samples_shaded = s.v[2];
    out->bricks_visited = s.v[3];
    out->bricks_skipped = s.v[4];
    out->rays_hit = s.v[5];
    return VRHIP_OK;
}

int vrhip_set_environment_map(vrhip_renderer *r, const float *rgba, uint32_t width,
                              uint32_t height)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_HIP(r, hipStreamSynchronize(r->stream));
    if (r->env) VR_HIP(r, hipFree(r->env));
    r->env = nullptr;
    r->env_w = r->env_h = 0;
    // the kernel only samples maps wider than one texel (:655); createEnvironmentMap("")
    // installs a 1x1 white one, i.e. none
    if (!rgba || width <= 1 || height == 0) return VRHIP_OK;
    VR_REQUIRE(r, width <= 16384 && height <= 16384, VRHIP_ERR_INVALID,
               "Environment map too large.");
    const size_t bytes = (size_t)width * height * sizeof(float4);
    VR_HIP(r, hipMalloc((void **)&r->env, bytes));
    VR_HIP(r, hipMemcpy(r->env, rgba, bytes, hipMemcpyHostToDevice));
    r->env_w = width;
    r->env_h = height;
    return VRHIP_OK;
}

int vrhip_reset_image_ess(vrhip_renderer *r)
{
    if (!r) return VRHIP_ERR_INVALID;
    r->hit_w = r->hit_h = 0;   // re-created (and re-initialised) by the next imgEss frame
    return VRHIP_OK;
}

int vrhip_get_image_ess(vrhip_renderer *r, uint32_t width, uint32_t height, uint8_t *hit_in,
                        uint8_t *hit_out)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_REQUIRE(r, width > 0 && height > 0 && width <= 16384 && height <= 16384, VRHIP_ERR_INVALID,
               "Invalid output image size.");
    int rc = ensure_hit_images(r, width, height);
    if (rc) return rc;
    const size_t n = (size_t)r->hit_w * r->hit_h;
    VR_HIP(r, hipStreamSynchronize(r->stream));
    if (hit_in) VR_HIP(r, hipMemcpy(hit_in, r->hit_in, n, hipMemcpyDeviceToHost));
    if (hit_out) VR_HIP(r, hipMemcpy(hit_out, r->hit_out, n, hipMemcpyDeviceToHost));
    return VRHIP_OK;
}

int vrhip_set_image_ess(vrhip_renderer *r, uint32_t width, uint32_t height, const uint8_t *hit_in,
                        const uint8_t *hit_out)
{
    if (!r) return VRHIP_ERR_INVALID;
    if (set_device(r)) return VRHIP_ERR_HIP;
    VR_REQUIRE(r, width > 0 && height > 0 && width <= 16384 && height <= 16384, VRHIP_ERR_INVALID,
               "Invalid output image size.");
    int rc = ensure_hit_images(r, width, height);
    if (rc) return rc;
    const size_t n = (size_t)r->hit_w * r->hit_h;
    VR_HIP(r, hipStreamSynchronize(r->stream));
    if (hit_in) VR_HIP(r, hipMemcpy(r->hit_in, hit_in, n, hipMemcpyHostToDevice));
    if (hit_out) VR_HIP(r, hipMemcpy(r->hit_out, hit_out, n, hipMemcpyHostToDevice));
    return VRHIP_OK;
}

int vrhip_count_touched(vrhip_renderer *r, uint32_t width, uint32_t height,
                        uint64_t *microbricks_touched, uint8_t *bitmap_host, size_t bitmap_bytes)
{
    return count_touched_impl(r, width, height, 0, 0, nullptr, 0, microbricks_touched,
                              bitmap_host, bitmap_bytes);
}

int vrhip_count_fetched(vrhip_renderer *r, uint32_t width, uint32_t height, uint64_t *microbricks_fetched)
{
    if (r && r->render.technique != 1)
        return fail(r, VRHIP_ERR_UNSUPPORTED, "vrhip_count_fetched: technique 1 (path tracer) only");
    return count_touched_impl(r, width, height, 0, 0, nullptr, 0, microbricks_fetched, nullptr, 0, true);
}

int vrhip_count_touched_tiles(vrhip_renderer *r, uint32_t width, uint32_t height,
                              uint32_t tile_w, uint32_t tile_h, const uint32_t *tile_ids,
                              uint32_t n_tiles, uint64_t *microbricks_touched)
{
    if (r && (!tile_ids || !n_tiles))
        return fail(r, VRHIP_ERR_INVALID, "vrhip_count_touched_tiles: empty tile list");
    return count_touched_impl(r, width, height, tile_w, tile_h, tile_ids, n_tiles,
                              microbricks_touched, nullptr, 0);
}

} // extern "C"
