// vr_internal.h -- descriptors shared by the kernels and the C-ABI implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "../../include/vrhip.h"

// Scalar field in HBM (DESIGN.md "Data layout"): 4x4x4-voxel MICRO-BRICKS of 64 contiguous
// voxels (x fastest inside a brick) -- one 64-byte line for UCHAR -- stored x-fastest over the
// brick grid.  Every line a ray bundle pulls in is a compact 3-D neighbourhood, whatever the
// view direction.  Element index of voxel (x, y, z):
//   (((z>>2)*nby + (y>>2))*nbx + (x>>2))*64 + (z&3)*16 + (y&3)*4 + (x&3)
struct VolView {
    const void *data;
    int w, h, d;
    float fw, fh, fd;
    float inv_max;              // UNORM scale: 1/255, 1/65535, 1
    uint32_t nbx, nby, nbz;     // micro-brick grid = ceil(res / 4)
    uint32_t ystride;           // elements per brick row:   nbx * 64
    unsigned long long zstride; // elements per brick slice: nbx * nby * 64
    // CL_RG / CL_RGBA volumes (volumerendercl.cpp:697-705): one planar micro-bricked array per
    // channel.  `data` is channel 0 -- the .x every single-channel reader of the kernel looks at
    // (bricks, gradients, path tracer); chan[0..2] are channels 1..3.
    const void *chan[3];
    int channels;               // 1, 2 or 4
    // Footprint volume (optional, nullptr = absent): entry (ex, ey, ez), ex in [0, w] etc., holds
    // the 8 voxels a trilinear fetch with low-corner texel ix = ex - 1 reads, edge clamping
    // applied: value j = dx + 2 dy + 4 dz is voxel (clamp(ix + dx), clamp(iy + dy), clamp(iz + dz)).
    // One 8 / 16 / 32-byte load per fetch instead of eight voxel loads; 8x the volume's bytes, laid
    // out in 4x4x4 micro-bricks of entries like the volume itself.
    const void *fp;
    uint32_t fp_nbx, fp_nby;    // micro-brick grid of the (w+1) x (h+1) x (d+1) entries
};

__host__ __device__ inline unsigned long long vr_fp_index(const VolView &v, int ex, int ey, int ez)
{
    return ((unsigned long long)(ez >> 2) * v.fp_nby + (unsigned long long)(ey >> 2)) * v.fp_nbx * 64ull +
           (unsigned long long)(((ex >> 2) << 6) + ((ez & 3) << 4) + ((ey & 3) << 2) + (ex & 3));
}

__host__ __device__ inline unsigned long long vr_voxel_index(const VolView &v, int x, int y, int z)
{
    return (unsigned long long)(z >> 2) * v.zstride + (unsigned long long)(y >> 2) * v.ystride +
           (unsigned long long)(((x >> 2) << 6) + ((z & 3) << 4) + ((y & 3) << 2) + (x & 3));
}

// min/max brick grid (generateBricks): (min,max) pairs in the volume's type, x fastest.
struct BrickView {
    const void *data;
    int bw, bh, bd;
};

struct TfView {
    const float4 *tff;     // RGBA8 table converted with c / 255.0f
    uint32_t tff_n;
    const uint32_t *prefix;
    uint32_t prefix_n;
};

// ESS decision per brick, precomputed from (bricks, TF, prefix sum): bit = 1 when the
// reference's test `TF(max).a < 1e-6 && prefix[min] == prefix[max]` (volumeraycast.cl:
// 777-787) holds.  x-fastest bit index; word n_words holds the decision for the (0,0) value
// defined for out-of-range cells.
struct SkipView {
    const uint32_t *bits;
    uint32_t n_words;
    uint32_t in_lds;       // bitmap is staged in LDS by every workgroup
    // Patch culling in the DDA pre-pass: bit = 1 when some brick within `near_r` bricks (Chebyshev,
    // per axis) of this one is NOT skipped.  An 8x8 patch whose rays stay -- by a conservative
    // bound -- inside bricks with bit 0 cannot meet a brick to sample: its rays are finished without
    // the walk.  nullptr / 0: off.
    const uint32_t *near_bits;
    uint32_t near_r;
};

// One 8x8-pixel patch (= one wave64) of the work queue.
struct WaveTile {
    uint16_t tx8, ty8;     // patch origin / 8 in the frame, bits 0..10; bits 11..15 of ty8 and of tx8: frame of the batch
    uint32_t out_base;     // index of its first pixel in FrameView::out
};
// Several independent frames (jitter seeds) in one set of launches: the work queue holds every
// patch once per frame, the frame index rides in the upper bits of WaveTile::ty8 (low five bits) and
// WaveTile::tx8 (high five bits) -- patch rows and columns need 11 bits up to 16384 pixels -- and of
// ContRec::state, FrameView::seeds gives each frame its seed.  32 whole frames fill a GPU; a rank's tile
// share of an 8-rank split needs 256 frames per set for the same work per launch.
constexpr uint32_t kFrameShift = 11, kMaxBatchFrames = 256;
constexpr uint32_t kPatchMask = (1u << kFrameShift) - 1u, kFrameLowBits = 16u - kFrameShift;
__host__ __device__ inline uint32_t wt_row(const WaveTile &w) { return w.ty8 & kPatchMask; }
__host__ __device__ inline uint32_t wt_col(const WaveTile &w) { return w.tx8 & kPatchMask; }
__host__ __device__ inline uint32_t wt_frame(const WaveTile &w)
{
    return (uint32_t)(w.ty8 >> kFrameShift) | ((uint32_t)(w.tx8 >> kFrameShift) << kFrameLowBits);
}
__host__ __device__ inline void wt_set_frame(WaveTile &w, uint32_t f)
{
    w.ty8 = (uint16_t)((w.ty8 & kPatchMask) | ((f & ((1u << kFrameLowBits) - 1u)) << kFrameShift));
    w.tx8 = (uint16_t)((w.tx8 & kPatchMask) | ((f >> kFrameLowBits) << kFrameShift));
}

// A patch with at least one ray that reaches a non-skipped brick (DDA pre-pass), and which rays.
struct LiveTile {
    WaveTile wt;
    uint32_t mask_lo, mask_hi;   // bit = lane (8 * row + column) of the patch
};

// Dynamic state of a suspended ray (two-phase march): everything else is recomputed from the
// pixel.  64 bytes.
struct ContRec {
    uint32_t pix;          // gx | gy << 16
    uint32_t out_index;    // pixel index in FrameView::out
    int32_t state;
    float t, t_exit, alpha, r0, r1, r2;
    int32_t cx, cy, cz;
    float tv0, tv1, tv2;
    uint32_t pad;
};

struct FrameView {
    uint32_t W, H;         // frame size in pixels
    uint32_t gsx, gsy;     // padded launch size the reference derives the camera from
    // what make_ray derives from it for every pixel alike (:614-627), computed once on the host with
    // the same IEEE operations: min(gsy / gsx, gsx / gsy), 2 / gsx, 2 / gsy
    float ray_aspect, ray_psx, ray_psy;
    const WaveTile *queue; // n_wave_tiles entries, processed in order (centre first)
    uint32_t n_wave_tiles;
    uint32_t out_stride;   // pixels per row of `out`
    uint32_t *queue_head;  // zeroed before each launch
    // the control words (kControlWords from queue_head on) of the NEXT set of launches: the renderer
    // alternates between two blocks, and the first kernel of a set zeroes the other block -- nothing else
    // touches it until the next set starts -- instead of a memset launch per frame.  nullptr: nothing to zero
    uint32_t *next_ctrl;
    // Patch classes (vr_patch_class_kernel, once per camera / transfer function / tile set, not per frame):
    // class 1 = whatever the jitter, every ray of the patch hits the box and none can meet a brick that is
    // not skipped, and the background is one colour -- the pre-pass writes 64 times (background, alpha 0)
    // for it without setting up a single ray.  patch_class[q / set_frames] for work item q; nullptr: off
    const uint8_t *patch_class;
    uint32_t set_frames;
    float4 *fb;            // W*H frame / accumulate buffer (always written)
    float4 *out;           // optional second destination (device), frame or tile layout
    // Two-phase march: rays still alive after `round_budget` sample rounds of phase 1 (0 =
    // unlimited) are appended to `cont` and resumed by the split kernel, 16 lanes per ray.
    ContRec *cont;
    uint32_t *cont_count;  // zeroed before each launch
    uint32_t *cont_head;   // zeroed before each launch
    uint32_t round_budget;
    const uint32_t *seeds; // per frame of a batch (WaveTile frame index), or nullptr: rendering_params.seed
    uint32_t refill_min;   // phase 2: idle ray slots of a wave before they take new rays (0 = all 16)
    // DDA pre-pass (ESS, un-instrumented): a light, high-occupancy kernel walks every ray to its
    // first non-skipped brick; rays that never reach one get their (background) pixel there and
    // phase 1 only visits the patches listed in `live`.  nullptr: phase 1 walks the whole queue.
    LiveTile *live;
    uint32_t *live_count;  // zeroed before each launch (the list of live PATCHES, FrameView::live)
    uint32_t *live_list_count;   // kLiveLists counters, kLiveStride words apart (the lists of live RAYS, live_rays)
    uint32_t *draw_count;        // kDrawCounters counters, kLiveStride words apart (the persistent kernels' draws)
    // Default kernels: the pre-pass lists the live RAYS instead (with the DDA state they have
    // reached; live_count counts them) and phase 1 runs on that list, one lane per ray, lanes
    // refilled as rays end (vr_raycast_rays_kernel).  nullptr: the patch list above.
    ContRec *live_rays;
    // 1: the default modes march the ray list with vr_march_kernel (stepping / dense evaluation /
    // compositing as stages of a round) instead of the two-phase march; a schedule, not a result
    uint32_t march;
    uint32_t lds_stage;   // experiment (VRHIP_LDS_STAGE): 1 = patch kernel with LDS-staged voxel boxes, 2 = same without
    uint32_t march_micro, march_fill;   // tuning (0 = built-in): micro-steps per round, queue fill that ends stage A
    // Phase-2 scheduling: `cost` keeps, per pixel, the phase-2 rounds the pixel's ray needed in the
    // previous frame.  Suspended rays are sorted by it, longest first (counting sort into
    // `order`), so that the longest chains start first and the 16 rays of a group are alike.
    // Purely a schedule: the image does not depend on it.
    uint16_t *cost;        // W*H, or nullptr
    uint32_t *order;       // permutation of the suspended rays, or nullptr (append order)
    uint32_t *sort_ws;     // kSortBins counts + kSortBins cursors, zeroed before each launch
    // Image-order ESS (rendering_params.imgEss, volumeraycast.cl:659-670, :912-925): one texel per
    // 8x8 work-group, (W/8 + 1) x (H/8 + 1) of them (volumerendercl.cpp:482-488).
    const uint8_t *hit_in; // last frame's hit image
    uint8_t *hit_status;   // this frame: what each group's work-items did (HIT_*, vr_raycast.hip)
    uint8_t *hit_any;      // this frame: 1 = a work-item that reached the end changed its pixel;
                           // zeroed before each launch
    uint32_t hit_w, hit_h;
    // Environment map (createEnvironmentMap, volumerendercl.cpp:1121-1150): float RGBA texels,
    // nullptr when none is set (the reference's 1x1 white map is never sampled, :655)
    const float4 *env;
    uint32_t env_w, env_h;
};
constexpr uint32_t kSortBins = 256;
// The pre-pass's ray list is kLiveLists lists, work item q appending to list q % kLiveLists through a counter of its
// own, each counter in a cache line of its own (kLiveStride words apart): waves that run side by side walk neighbouring
// work items, so their appends meet at kLiveLists addresses instead of one (one counter: 193 us of a 20-frame set's
// 580 us of pre-pass).  Phase 1 reads the lists interleaved, 64 rays from each in turn (vr_raycast_rays_kernel), which
// keeps the queue's order: all lists grow at the same pace.  List k sits at live_rays + k * live_list_cap(items).
#ifdef VR_EXPERIMENTS   // (the opt-in march kernel of the A/B builds reads one list)
constexpr uint32_t kLiveLists = 1, kLiveStride = 32;
#else
constexpr uint32_t kLiveLists = 8, kLiveStride = 32;
#endif
constexpr uint32_t kLiveBase = 4 + 2 * kSortBins;        // first list counter, in control words
// ... and the persistent kernels draw their work through kDrawCounters counters instead of one (FrameView::draw_count, a
// cache line each): counter k hands out the work units G + kDrawCounters j + k behind the G units the waves own by
// position; a wave starts at counter (its number mod kDrawCounters) and moves on when one has run out.
constexpr uint32_t kDrawCounters = 8;
constexpr uint32_t kDrawBase = kLiveBase + kLiveLists * kLiveStride;
constexpr uint32_t kControlWords = kDrawBase + kDrawCounters * kLiveStride;   // queue head, cont count, cont head, live-tile count, sort_ws, list counters, draw counters
__host__ __device__ inline uint32_t live_list_cap(uint32_t n_items) { return ((n_items + kLiveLists - 1u) / kLiveLists) * 64u; }
// first kernel of a set of launches, first workgroup: the control words of the next set (FrameView::next_ctrl)
#define VR_ZERO_NEXT_CTRL(fr)                                                                              \
    do {                                                                                                   \
        if (blockIdx.x == 0 && (fr).next_ctrl)                                                             \
            for (uint32_t i_ = threadIdx.x; i_ < kControlWords; i_ += blockDim.x) (fr).next_ctrl[i_] = 0u; \
    } while (0)

// Cell grid: the volume cut into cells of 2^shift voxels per axis.  For every cell the renderer
// keeps (min, max) of the voxels a trilinear fetch whose low-corner texel lies in the cell -- or
// within one texel of it -- can read: voxels [(c << shift) - 1, ((c + 1) << shift) + 1] per axis.
// Combined with the transfer function this gives, per cell (vr_cells.hip):
//  * bound: an upper bound of the opacity any such fetch can map to.  Path tracer: a tracking
//    step whose bound is below the walk's acceptance threshold cannot be accepted, so its voxel
//    loads are skipped.
//  * empty: 1 bit, set when that opacity is exactly 0.  Ray caster: such a sample composites to
//    exactly nothing (volumeraycast.cl:864-879 with alpha == 0), so runs of them are stepped over
//    with the reference's own t sequence and without touching the volume.
// Neither changes any pixel: both only skip work whose result is known.
// The two live on grids of their own: the bounds on cells of 2^shift voxels (8 up to 2048^3: the path
// tracer's table stays cache-sized), the empty bits on cells of 2^eshift voxels (4 up to 2048^3:
// fewer fetches at the rims of a structure that can only return opacity 0).  A coarse cell's extent
// is the union of the extents of the fine cells inside it, so the coarse (min, max) are reduced
// from the fine ones where both exist.
struct CellView {
    const float *bound;     // cx * cy * cz floats, x fastest; nullptr = feature off
    const uint32_t *empty;  // 1 bit per cell of the ecx * ecy * ecz grid, x fastest; nullptr = feature off
    int cx, cy, cz;
    int shift;
    int ecx, ecy, ecz;
    int eshift;
    // The empty bits once more, per ESS brick (the march kernel keeps the words of the bricks a ray
    // is in in registers): brick (bx, by, bz) of the bw x bh x bd brick grid, x fastest, is cut into
    // 4 x 4 x 4 sub-blocks of (edge / 4) voxels; bit i + 4 j + 16 k of its word is set when every
    // cell that overlaps sub-block (i, j, k) is empty.  nullptr: not available (brick edge < 4).
    const unsigned long long *bmask;
    int bex, bey, bez;      // log2 of the brick edge per axis (>= 2)
    // The bounds once more, per macro cell of 4 x 4 x 4 cells (kLeapShift): the maximum of its cells' bounds -- what
    // lets a tracking walk leap over all its steps inside a macro cell at once (vr_pathtrace.hip).  ccx * ccy * ccz
    // floats, x fastest; nullptr = no leaps.
    const float *cbound;
    int ccx, ccy, ccz;
    // ... and how far the macro cells around one are free as well, for kLeapLevels thresholds tau_j = j / 8, j = 1..7:
    // cdist[(j - 1) * ccx * ccy * ccz + C] = 0 when macro cell C has a bound >= tau_j, else 1 + the largest R <= kLeapRadius
    // such that every macro cell of the grid within Chebyshev distance R of C has a bound < tau_j.  A walk with
    // threshold thr reads level j = floor(8 thr) (tau_j <= thr) and may leap through that whole cube.  nullptr: leaps
    // stay inside one macro cell.
    const uint8_t *cdist;
};

struct DevStats {
    unsigned long long v[6]; // order of vrhip_stats
};

struct RaycastLaunch {
    VolView vol;
    BrickView bricks;
    TfView tf;
    SkipView skip;
    FrameView frame;
    vrhip_camera_params cam;
    vrhip_rendering_params render;
    vrhip_raycast_params raycast;
    vrhip_pathtrace_params pathtrace;
    CellView cells;
    int format;            // vrhip_format
    int use_ess;
    int instr;             // 0 none, 1 stats, 2 stats + touched bitmap
    int occ3;              // phase 1 on the ray list at three waves per SIMD (vr_raycast.hip kWavesWide): a schedule, not a result
    int occ3_split;        // the same for phase 2
    DevStats *stats;
    uint32_t *touched;
    int num_cus;
    hipEvent_t mid_event;  // optional: the end of phase 1
    // optional: the end of the frame's last launch.  Both events are BOUND to a kernel's own completion signal
    // (hipExtLaunchKernelGGL's stopEvent) rather than recorded behind it: a recorded event is a marker packet of its
    // own between two launches (~6 us of GPU time each).  *stop_bound tells the caller that the launchers have done
    // so (false: nothing was launched, or bind_events is off -- the caller records the event itself).
    hipEvent_t stop_event;
    bool *stop_bound;
    // optional, likewise: the start of the frame's first launch (hipExtLaunchKernelGGL's startEvent)
    hipEvent_t start_event;
    bool *start_bound;
    int bind_events;       // 0: record events behind the launches instead (VRHIP_EVENT_BIND=0, A/B)
    uint8_t *hit_out;      // imgEss: this frame's hit image (resolved after the march), or nullptr
    vrhip_launch_info *info;   // optional: the launchers record what they launched (vrhip_last_launch_info)
};

// hipLaunchKernelGGL, or -- with an event -- the launch whose completion the event is bound to
template <typename F, typename... Args>
inline void vr_launch_kernel(F k, dim3 grid, dim3 block, size_t lds, hipStream_t stream, hipEvent_t start,
                             hipEvent_t stop, Args... args)
{
    if (start || stop)
        hipExtLaunchKernelGGL(k, grid, block, (uint32_t)lds, stream, start, stop, 0u, args...);
    else
        hipLaunchKernelGGL(k, grid, block, lds, stream, args...);
}

hipError_t vr_launch_raycast(const RaycastLaunch &a, hipStream_t stream);
// FrameView::patch_class for the n_patches patches of a set of set_frames frames (work item p * set_frames = patch
// p of frame 0); the launch's camera, parameters, skip bitmaps and queue must be those of the frames to come
hipError_t vr_launch_patch_classes(const RaycastLaunch &a, uint32_t n_patches, uint32_t set_frames, uint8_t *cls,
                                   hipStream_t stream);
// 1 when vr_raycast.hip was built with the opt-in experiment kernels (-DVR_EXPERIMENTS: A/B builds only)
int vr_experiments_built();
// fills the footprint volume vol.fp of `vol`: (w+1)(h+1)(d+1) entries rounded up to
// whole micro-bricks, 8 values of the volume's type each; vr_bricks.hip
hipError_t vr_launch_build_footprint(const VolView &vol, int format, hipStream_t stream);
// technique 1 (Woodcock-tracking path tracer), one sample per pixel; vr_pathtrace.hip
hipError_t vr_launch_pathtrace(const RaycastLaunch &a, hipStream_t stream);
// cell grid (vr_cells.hip): per-cell (min,max) of the raw voxel values incl. halo; then opacity
// bound + empty bit from (min,max) and the transfer function.  sparse_scratch: 13 * 4096 floats.
// records: d * cy * cx float2 of scratch for the separable streaming build (cells of <= 16 voxels,
// micro-brick rows of <= 2048 bricks: the x classes of a row live in LDS), or nullptr: the
// one-wave-per-cell kernel
// records: scratch of vr_cell_record_bytes(format) * grid.cx * grid.cy * vol.d bytes for the separable
// streaming build (nullptr, or cells of more than 16 voxels: the one-wave-per-cell kernel)
inline size_t vr_cell_record_bytes(int format) { return format == VRHIP_UCHAR ? 2 : format == VRHIP_USHORT ? 4 : 8; }
hipError_t vr_launch_cell_minmax(const VolView &vol, int format, const CellView &grid,
                                 float2 *minmax, hipStream_t stream, void *records = nullptr);
hipError_t vr_launch_cell_bounds(const float2 *minmax, const CellView &grid, float inv_max,
                                 const TfView &tf, float *sparse_scratch, float *bound,
                                 uint32_t *empty_bits, hipStream_t stream);
// (min, max) of the coarse grid `grid` (cx.., shift) from those of the fine one (ecx.., eshift < shift)
hipError_t vr_launch_cell_reduce(const float2 *fine, const CellView &grid, float2 *coarse, hipStream_t stream);
// CellView::cbound from CellView::bound (grid.cx/cy/cz, grid.ccx/ccy/ccz set)
#ifndef VR_LEAP_SHIFT
#define VR_LEAP_SHIFT 2
#endif
constexpr int kLeapShift = VR_LEAP_SHIFT;
hipError_t vr_launch_cell_coarse_bounds(const CellView &grid, float *cbound, hipStream_t stream);
constexpr int kLeapLevels = 7, kLeapRadius = 15;
// CellView::cdist from `cbound`: `dist` holds 2 x kLeapLevels x ccx*ccy*ccz bytes (two buffers the erosion passes
// alternate between); returns in *result the buffer that holds the table at the end
hipError_t vr_launch_cell_leap_radius(const CellView &grid, const float *cbound, uint8_t *dist, const uint8_t **result,
                                      hipStream_t stream);
// CellView::bmask from CellView::empty for the bw x bh x bd brick grid (grid.bex.. set)
hipError_t vr_launch_cell_bmask(const VolView &vol, const CellView &grid, int bw, int bh, int bd,
                                unsigned long long *bmask, hipStream_t stream);
// the frame launch for a.render.technique
inline hipError_t vr_launch_frame(const RaycastLaunch &a, hipStream_t stream)
{
    return a.render.technique == 1 ? vr_launch_pathtrace(a, stream) : vr_launch_raycast(a, stream);
}

// skip bitmap from bricks + TF + prefix
hipError_t vr_launch_skipmap(const BrickView &bricks, int format, float inv_max, const TfView &tf,
                             uint32_t *bits, uint32_t n_words, hipStream_t stream);
// SkipView::near_bits from the skip bitmap: scratch = 2 bytes per brick
hipError_t vr_launch_skip_near(const BrickView &bricks, const uint32_t *bits, uint32_t n_words, uint32_t radius,
                               uint8_t *scratch, uint32_t *near_bits, hipStream_t stream);

hipError_t vr_launch_build_bricks(const VolView &vol, int format, const uint32_t tex[3],
                                  void *bricks_out, hipStream_t stream);
// downsampling kernel (volumeraycast.cl:966-994): lo = low-res size, vpc = voxels per cell
hipError_t vr_launch_downsample(const VolView &vol, int format, const int lo[3], const int vpc[3],
                                void *out_dense, hipStream_t stream);
// synthetic field written straight into the micro-brick layout
hipError_t vr_launch_synth(int kind, const VolView &vol, int format, hipStream_t stream);
// dense x-fastest slices [z0, z0+nz) (device memory, `dense` points at slice z0) <-> bricks
hipError_t vr_launch_retile(const VolView &vol, int format, const void *dense, int z0, int nz,
                            bool to_bricks, hipStream_t stream);

// Blocks per CU of `kernel` (block_dim threads, `lds` bytes of dynamic LDS), after raising the
// kernel's dynamic LDS limit where needed.  Both are per device (hipFuncSetAttribute acts on the
// current device's copy of the function), so the answers are cached per (kernel address, device,
// LDS size), under a lock: renderers on several devices or host threads share this code.  The
// kernel's address is part of the key because the function-local statics exist once per pointer
// TYPE K, and many instantiations (ESS / INSTR / XS / FP variants) share one signature.
template <typename K>
hipError_t vr_prepare_kernel(K kernel, int block_dim, size_t lds, int *nb_out, const char *what, int num_cus)
{
    static std::mutex mu;
    static std::map<std::tuple<const void *, int, size_t>, int> cache;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple((const void *)kernel, dev, lds);
    const auto it = cache.find(key);
    if (it != cache.end()) { *nb_out = it->second; return hipSuccess; }
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block_dim, lds) != hipSuccess || nb < 1)
        nb = 1;
    if (const char *ev = getenv("VRHIP_BLOCKS_PER_CU")) {   // tuning / experiments
        const int v = atoi(ev);
        if (v > 0) nb = v;
    }
    if (getenv("VRHIP_DEBUG"))
        fprintf(stderr, "[vrhip] %s @%p: device %d, lds=%zu B, blocks/CU=%d, CUs=%d\n", what, (const void *)kernel, dev,
                lds, nb, num_cus);
    cache[key] = nb;
    *nb_out = nb;
    return hipSuccess;
}
