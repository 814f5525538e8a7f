// vr_internal.h -- descriptors shared by the kernels and the C-ABI implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vrhip.h"

// Scalar field in HBM. Layout (DESIGN.md "Data layout"): dense, x fastest, then y, z.
struct VolView {
    const void *data;
    int w, h, d;
    float fw, fh, fd;
    float inv_max;              // UNORM scale: 1/255, 1/65535, 1
    unsigned long long row;     // voxels per y step
    unsigned long long slice;   // voxels per z step
    int mbx, mby;               // micro-brick (4^3 voxels) grid, for the traffic bitmap
};

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
// 777-787) holds.  x-fastest bit index; `oob_skip` is the decision for the (0,0) value
// defined for out-of-range cells.
struct SkipView {
    const uint32_t *bits;
    uint32_t n_words;
    uint32_t oob_skip;
    uint32_t in_lds;       // bitmap is staged in LDS by every workgroup
};

// One 8x8-pixel patch (= one wave64) of the work queue.
struct WaveTile {
    uint16_t tx8, ty8;     // patch origin / 8 in the frame
    uint32_t out_base;     // index of its first pixel in FrameView::out
};

struct FrameView {
    uint32_t W, H;         // frame size in pixels
    uint32_t gsx, gsy;     // padded launch size the reference derives the camera from
    const WaveTile *queue; // n_wave_tiles entries, processed in order (centre first)
    uint32_t n_wave_tiles;
    uint32_t out_stride;   // pixels per row of `out`
    uint32_t *queue_head;  // zeroed before each launch
    float4 *fb;            // W*H frame / accumulate buffer (always written)
    float4 *out;           // optional second destination (device), frame or tile layout
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
    int format;            // vrhip_format
    int use_ess;
    int instr;             // 0 none, 1 stats, 2 stats + touched bitmap
    DevStats *stats;
    uint32_t *touched;
    int num_cus;
};

hipError_t vr_launch_raycast(const RaycastLaunch &a, hipStream_t stream);

// skip bitmap from bricks + TF + prefix
hipError_t vr_launch_skipmap(const BrickView &bricks, int format, float inv_max, const TfView &tf,
                             uint32_t *bits, uint32_t n_words, uint32_t *oob_skip_dev,
                             hipStream_t stream);

hipError_t vr_launch_build_bricks(const VolView &vol, int format, const uint32_t tex[3],
                                  void *bricks_out, hipStream_t stream);
hipError_t vr_launch_synth(int kind, void *dst, const uint32_t res[3], unsigned long long row,
                           unsigned long long slice, int format, hipStream_t stream);
