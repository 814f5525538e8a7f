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

struct FrameView {
    uint32_t W, H;         // frame size in pixels
    uint32_t gsx, gsy;     // padded launch size the reference derives the camera from
    uint32_t blocks_x;     // 16x16-pixel blocks per row (full-frame mode)
    // tile mode (tile_ids != nullptr): compact output [n_tiles][tile_h][tile_w]
    const uint32_t *tile_ids;
    uint32_t tile_w, tile_h, tiles_x, bpt_x, bpt;
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
    uint32_t n_blocks;
};

hipError_t vr_launch_raycast(const RaycastLaunch &a, hipStream_t stream);

hipError_t vr_launch_build_bricks(const VolView &vol, int format, const uint32_t tex[3],
                                  void *bricks_out, hipStream_t stream);
hipError_t vr_launch_synth(int kind, void *dst, const uint32_t res[3], int format,
                           hipStream_t stream);
