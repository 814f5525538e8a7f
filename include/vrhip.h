/*
 * vrhip.h -- C ABI of the MI355X-native volume ray-caster (libvrhip.so).
 *
 * This is the drop-in boundary underneath the reference's C++ class
 * `VolumeRenderCL` (/root/reference/src/core/volumerendercl.h:39-482): every entry
 * point below replaces one OpenCL-side action of that class and cites it.  Plain
 * pointers and sizes only; no C++/torch types.  All functions return VRHIP_OK (0)
 * or an error code; vrhip_last_error() gives the message (the host class turns it
 * into the std::runtime_error the reference throws, volumerendercl.cpp:83-89).
 *
 * Not thread-safe per renderer (like the reference's single in-order queue,
 * volumerendercl.cpp:140).  One renderer == one GPU.
 */
#ifndef VRHIP_H
#define VRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRHIP_ABI_VERSION 1

enum vrhip_status {
    VRHIP_OK = 0,
    VRHIP_ERR_INVALID = 1,     /* bad argument (reference: std::invalid_argument)          */
    VRHIP_ERR_HIP = 2,         /* HIP runtime failure (reference: cl::Error -> runtime_error) */
    VRHIP_ERR_NODATA = 3,      /* no volume / TF / bricks yet                               */
    VRHIP_ERR_UNSUPPORTED = 4  /* feature outside the hot path                  */
};

/* DatRawReader::data_format, src/io/datrawreader.h:41-48 */
enum vrhip_format { VRHIP_UCHAR = 0, VRHIP_USHORT = 1, VRHIP_FLOAT = 2 };

/* Kernel-argument structs: byte-identical to the reference's
 * camera_params / rendering_params / raycast_params / pathtrace_params
 * (src/core/volumerendercl.h:43-81 == src/kernel/volumeraycast.cl:540-582). */
typedef struct vrhip_camera_params {
    float viewMat[16];   /* row-major 4x4 (updateView, volumerendercl.cpp:379-390) */
    float bbox_bl[4];    /* cl_float3 in a 16-byte slot */
    float bbox_tr[4];
    uint32_t ortho;
    uint32_t _pad[7];
} vrhip_camera_params;   /* 128 bytes */

typedef struct vrhip_rendering_params {
    float backgroundColor[4];
    float modelScale[4]; /* cl_float3 in a 16-byte slot */
    uint32_t illumType;  /* 0 off, 1 central differences, 2 TF-opacity differences, 3 Sobel,
                            4 gradient magnitude through the TF, 5 cel (volumeraycast.cl:796-830) */
    uint32_t imgEss;
    uint32_t showEss;
    uint32_t useLinear;
    uint32_t useGradient;
    uint32_t technique;  /* 0 ray cast, 1 path tracing */
    uint32_t seed;
    uint32_t iteration;
} vrhip_rendering_params; /* 64 bytes */

typedef struct vrhip_raycast_params {
    float samplingRate;
    uint32_t useAO;
    uint32_t contours;
    uint32_t aerial;
    float brickRes[4];   /* volume_res / brick_edge; written by vrhip_build_bricks */
} vrhip_raycast_params;  /* 32 bytes */

typedef struct vrhip_pathtrace_params {
    float max_extinction;
} vrhip_pathtrace_params;

/* Work counters of the last instrumented render (SURVEY 8d "sample" definitions).
 * technique 0: samples = executions of the inner loop body (volumeraycast.cl:790-880), bricks =
 * DDA steps / steps skipped by ESS.  technique 1: samples_taken = tracking steps inside the
 * volume (sample_interaction :419-431); the two brick counters are reused for the opacity-bound
 * culling: bricks_visited = steps whose bound was consulted (their cell's or, in a leap, their
 * macro cell's), bricks_skipped = steps whose voxel fetch was skipped, samples_nominal = those of
 * them that were taken in leaps over a macro cell; samples_shaded stays 0. */
typedef struct vrhip_stats {
    uint64_t samples_taken;
    uint64_t samples_nominal;
    uint64_t samples_shaded;
    uint64_t bricks_visited;
    uint64_t bricks_skipped;
    uint64_t rays_hit;
} vrhip_stats;

typedef struct vrhip_renderer vrhip_renderer;

/* ---- life cycle: VolumeRenderCL::initialize (volumerendercl.cpp:95-159) -------- */
int vrhip_abi_version(void);
int vrhip_create(int device_id, vrhip_renderer **out);
void vrhip_destroy(vrhip_renderer *r);
/* message of the last failing call on `r` (or of vrhip_create when r == NULL) */
const char *vrhip_last_error(const vrhip_renderer *r);
/* getCurrentDeviceName (volumerendercl.cpp:1114-1117) */
int vrhip_device_name(const vrhip_renderer *r, char *buf, size_t buf_len);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream; NULL is the legacy
 * default stream) instead of the renderer's own stream; use_own != 0 restores the latter. */
int vrhip_set_stream(vrhip_renderer *r, void *hip_stream, int use_own);
/* The hipStream_t the renderer launches on right now (its own stream unless vrhip_set_stream
 * gave it another): render calls return without synchronising, so whoever consumes a frame on
 * another stream (a collective, a copy) orders that stream behind this one. */
int vrhip_get_stream(const vrhip_renderer *r, void **hip_stream);

/* ---- volume: volDataToCLmem (volumerendercl.cpp:690-759) ----------------------- */
/* Dense x-fastest scalar field (CL_R image) of `format`, host memory. */
int vrhip_upload_volume(vrhip_renderer *r, const void *host_voxels, const uint32_t res[3],
                        int format, uint32_t timestep);
/* CL_RG / CL_RGBA volumes (volumerendercl.cpp:697-705; kernel volumeraycast.cl:838-855):
 * `channels` interleaved values of `format` per voxel, 2 = RG (colour (r,0,0), opacity TF(|g|)),
 * 4 = RGBA (the voxel is the sample's colour and opacity), 1 = vrhip_upload_volume.  Everything
 * that reads `.x` in the reference -- bricks, gradients, the path tracer, down-sampling -- sees
 * channel 0.  Other counts: VRHIP_ERR_INVALID ("Unknown or invalid volume color format."). */
int vrhip_upload_volume_channels(vrhip_renderer *r, const void *host_voxels, const uint32_t res[3],
                                 int format, int channels, uint32_t timestep);
/* Same, source already in device memory (HBM). */
int vrhip_upload_volume_device(vrhip_renderer *r, const void *dev_voxels, const uint32_t res[3],
                               int format, uint32_t timestep);
/* Synthetic inputs of SURVEY 8(d), generated on the GPU: kind 0 sphere, 1 shells. */
int vrhip_synth_volume(vrhip_renderer *r, int kind, const uint32_t res[3], int format,
                       uint32_t timestep);
/* Copy timestep `t` back as a dense x-fastest array (bytes must equal its size). */
int vrhip_download_volume(vrhip_renderer *r, uint32_t timestep, void *host_dst, size_t bytes);
/* volumeDownsampling's device part (volumerendercl.cpp:238-300 + kernel `downsampling`,
 * volumeraycast.cl:966-994): low-res size = ceil(res / factor) per axis; the low-res volume of
 * timestep `t` is copied to host_dst as a dense x-fastest array in the volume's type.  Call with
 * host_dst == NULL to query out_res only.  factor < 2 -> VRHIP_ERR_INVALID ("Factor must be
 * greater or equal 2."); low-res x size < 64 -> VRHIP_ERR_INVALID (the reference's minimum). */
int vrhip_downsample_volume(vrhip_renderer *r, uint32_t timestep, int factor, void *host_dst,
                            size_t bytes, uint32_t out_res[3]);
/* Scheduling of the two-phase march (no effect on any pixel): sample rounds (4 samples each) a ray
 * marches with one lane in phase 1 before it is suspended and resumed with four lanes in phase 2;
 * 0 = single phase.  The default, 10, minimises the time of ONE frame (long rays get the short
 * 4-lane chains early); with several frames in flight (vrhip_share_volumes) latency is hidden by
 * the other frames and the leaner 1-lane march should keep its rays longer: 32 renders the 2048^3
 * headline at 0.38 ms per frame instead of 0.45 (four in flight), but 0.80 instead of 0.70 alone. */
int vrhip_set_round_budget(vrhip_renderer *r, uint32_t rounds);
/* Frames in flight: renderer `r` (same device) renders from `owner`'s voxels and ESS bricks
 * instead of holding copies -- everything else (transfer function, parameters, frame and scratch
 * buffers, footprint volume, stream) is its own, so two renderers on two streams can
 * have one frame each in flight over one 8 GiB volume.  The owner keeps a list of its sharers:
 * when it clears its volumes, is destroyed, or uploads a volume of another size or a new time step,
 * the sharers are detached first (they wait for their streams, drop the borrowed volumes and answer
 * "No volume data is loaded." until they share again); an upload into an existing time step of the
 * same size overwrites the shared voxels in place after the sharers' streams have drained, and the
 * sharers rebuild what they had derived from them (the owner calls vrhip_build_bricks first).  `r`
 * gives the volumes back with vrhip_clear_volumes or by uploading its own.  vrhip_build_bricks on `r`
 * takes the owner's bricks, and on the owner it never moves them: a brick grid is allocated once per
 * time step and rebuilt only after that step's voxels were uploaded again, so transfer-function edits
 * on either renderer are safe. */
int vrhip_share_volumes(vrhip_renderer *r, vrhip_renderer *owner);
int vrhip_clear_volumes(vrhip_renderer *r);
/* setTimestep (volumerendercl.cpp:1167-1174) */
int vrhip_set_timestep(vrhip_renderer *r, uint32_t timestep);
int vrhip_get_resolution(const vrhip_renderer *r, uint32_t res_xyzt[4]);

/* ---- transfer function + ESS bricks: setTransferFunction (volumerendercl.cpp:864-891) */
/* RGBA8 table upload only (:870-876); n_entries <= 4096. */
int vrhip_set_transfer_function(vrhip_renderer *r, const uint8_t *rgba8, uint32_t n_entries);
/* setTffPrefixSum (:898-916) */
int vrhip_set_tff_prefix_sum(vrhip_renderer *r, const uint32_t *prefix, uint32_t n);
/* generateBricks host part + kernel for every timestep (:614-684, volumeraycast.cl:932-961);
 * also stores raycast.brickRes like :627-631. */
int vrhip_build_bricks(vrhip_renderer *r);
int vrhip_get_brick_info(const vrhip_renderer *r, uint32_t tex[3], float brick_res[3],
                         uint32_t edge[3]);
/* (min,max) pairs in the volume's own type, x-fastest; for tests. */
int vrhip_download_bricks(vrhip_renderer *r, uint32_t timestep, void *host_dst, size_t bytes);
double vrhip_last_bricks_seconds(const vrhip_renderer *r);

/* ---- kernel arguments: setCameraArgs/.. (volumerendercl.cpp:406-449) ----------- */
int vrhip_set_camera_params(vrhip_renderer *r, const vrhip_camera_params *p);
int vrhip_set_rendering_params(vrhip_renderer *r, const vrhip_rendering_params *p);
int vrhip_set_raycast_params(vrhip_renderer *r, const vrhip_raycast_params *p);
int vrhip_set_pathtrace_params(vrhip_renderer *r, const vrhip_pathtrace_params *p);
/* setObjEss (volumerendercl.cpp:1006-1019): the reference recompiles with/without -DESS;
 * here it selects the kernel variant. Default on (:150). */
int vrhip_set_object_ess(vrhip_renderer *r, int enabled);

/* ---- render: runRaycast / runRaycastNoGL (volumerendercl.cpp:506-607) ---------- */
/* Renders the whole width x height frame (launch grid padded like :513-514).
 * out_rgba: width*height*4 floats, row 0 = top, or NULL to keep the frame on the GPU.
 * out_is_device != 0: out_rgba is device memory. */
int vrhip_render_frame(vrhip_renderer *r, uint32_t width, uint32_t height, float *out_rgba,
                       int out_is_device);
/* Image-tile decomposition for multi-GPU: renders n_tiles tiles of
 * tile_w x tile_h pixels (tile id = ty * ceil(width/tile_w) + tx) into the compact
 * DEVICE buffer out_tiles[n_tiles][tile_h][tile_w][4].  Pixels outside the frame are
 * left untouched.  tile_w, tile_h must be multiples of 16. */
int vrhip_render_tiles(vrhip_renderer *r, uint32_t width, uint32_t height, uint32_t tile_w,
                       uint32_t tile_h, const uint32_t *tile_ids, uint32_t n_tiles,
                       float *out_tiles_dev);
/* createEnvironmentMap (volumerendercl.cpp:1121-1150): float RGBA texels, row-major, width*height*4
 * floats (the host layer decodes the Radiance .hdr file).  The kernel samples it in place of the
 * background colour when it is wider than one texel (volumeraycast.cl:506-510, :655-656);
 * rgba == NULL or width <= 1 removes it. */
int vrhip_set_environment_map(vrhip_renderer *r, const float *rgba, uint32_t width,
                              uint32_t height);

/* ---- image-order ESS (rendering_params.imgEss; volumeraycast.cl:659-670, :912-925) --------
 * The renderer keeps the reference's two hit images, (width/8 + 1) x (height/8 + 1) texels of one
 * byte, creates them on the first imgEss frame of a size with updateOutputImg's initial contents
 * (volumerendercl.cpp:482-488) and swaps them after every imgEss frame (:524-530).  A tile render
 * only updates the texels of its own 8x8 work-groups: with several GPUs the caller merges the
 * images between frames (get on every rank, exchange, set; volumerenderercl_amd/tiles.py).
 * vrhip_reset_image_ess: what updateOutputImg does to them (re-created on the next frame).
 * get/set: hit_in = the image the next frame reads, hit_out = the one it writes; either may be
 * NULL.  Not supported together with technique 1. */
int vrhip_reset_image_ess(vrhip_renderer *r);
int vrhip_get_image_ess(vrhip_renderer *r, uint32_t width, uint32_t height, uint8_t *hit_in,
                        uint8_t *hit_out);
int vrhip_set_image_ess(vrhip_renderer *r, uint32_t width, uint32_t height, const uint8_t *hit_in,
                        const uint8_t *hit_out);
/* The cell grid behind empty-run skipping and the path tracer's culling (no reference counterpart;
 * DESIGN.md "Kernels"): per cell of 2^shift voxels the (min, max) of the raw voxel values a fetch in
 * or within one texel of the cell can read, as pairs of floats, x fastest.  Builds the grid of the
 * current time step if need be.  out == NULL: dims / shift only.  For tests. */
int vrhip_download_cells(vrhip_renderer *r, float *out_minmax, size_t n_floats, uint32_t dims[3],
                         uint32_t *shift);
/* The same for the finer grid the ray caster's empty bits live on (cells of 4 voxels up to 2048^3;
 * the grid above where the two coincide). */
int vrhip_download_empty_cells(vrhip_renderer *r, float *out_minmax, size_t n_floats, uint32_t dims[3],
                               uint32_t *shift);

/* Image-tile gather, root side: the frame from the gathered tiles.  `staging_dev` holds
 * tile slots of tile_w x tile_h RGBA float pixels (the peers' blocks as received, one after the
 * other); slot_of_tile_dev[t] is the slot of tile t (tiles numbered row-major over the frame).
 * One thread per pixel, enqueued on the renderer's stream; frame_dev = width x height x 4 floats. */
int vrhip_assemble_frame(vrhip_renderer *r, const float *staging_dev, const uint32_t *slot_of_tile_dev,
                         uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                         float *frame_dev);

/* Multi-GPU, rank 0: assemble the n_frames frames of a batch straight from the ranks' gather messages
 * (tiles.py TileDriver, sparse gather: a tile of one colour travels as one pixel).  msgs_dev: HOST array
 * of `world` (<= 64) device pointers, message r = [maxc floats: slot numbers of rank r's whole tiles |
 * S = n_frames x cap RGBA pixels, one per tile slot | maxc whole tiles of tile_w x tile_h RGBA pixels]
 * (16-byte aligned, maxc a multiple of 4); pos_dev[r * S + row] = -1 (one colour: that pixel) or the
 * position of row = frame x cap + slot among rank r's whole tiles; rank_slot_of_tile_dev[t] = rank << 16 |
 * slot of tile t (tiles numbered row-major over the frame).  One thread per pixel of frames_dev
 * [n_frames][height][width][4], enqueued on hip_stream (a hipStream_t; NULL = legacy default stream). */
int vrhip_assemble_batch(vrhip_renderer *r, void *hip_stream, const float *const *msgs_dev, uint32_t world,
                         uint32_t n_frames, uint32_t cap, uint32_t maxc, const int32_t *pos_dev,
                         const uint32_t *rank_slot_of_tile_dev, uint32_t width, uint32_t height, uint32_t tile_w,
                         uint32_t tile_h, float *frames_dev);

/* Multi-GPU, every rank: pack the n_slots tile slots of a batch (tile_pixels RGBA float pixels each, one after the
 * other in tiles_dev: a rank's output of vrhip_render_batch with a tile subset) into the sparse gather message
 * [spad slot numbers of the whole tiles (int32 bits) | n_slots pixels, one per slot | the whole tiles], spad =
 * n_slots rounded up to a multiple of 4: a tile whose pixels are all bit-identical travels as its one pixel.  Pure
 * data compression (no notion of a background).  msg_dev holds spad + 4 n_slots + 4 n_slots tile_pixels floats (the
 * worst case); count_dev receives the number c of whole tiles: the first spad + 4 n_slots + 4 c tile_pixels floats
 * are what has to travel.  scratch_dev: n_slots int32.  All device memory, 16-byte aligned; three small kernels on
 * hip_stream.  The root reads the messages with vrhip_message_positions + vrhip_assemble_batch (maxc = spad). */
int vrhip_pack_tiles(vrhip_renderer *r, void *hip_stream, const float *tiles_dev, uint32_t n_slots,
                     uint32_t tile_pixels, int32_t *scratch_dev, float *msg_dev, uint32_t *count_dev);
/* Multi-GPU, rank 0: pos_dev[rank * n_slots + row] (what vrhip_assemble_batch reads) from the slot lists at the
 * head of the `world` received messages; counts_host[rank] = that rank's number of whole tiles.  msgs_dev: HOST
 * array of device pointers. */
int vrhip_message_positions(vrhip_renderer *r, void *hip_stream, const float *const *msgs_dev,
                            const uint32_t *counts_host, uint32_t world, uint32_t n_slots, int32_t *pos_dev);

/* A batch of n_frames <= 256 INDEPENDENT frames (n_frames x pixels per frame < 2^32) -- same camera and parameters, frame f with jitter
 * seed seeds[f] (rendering_params.seed is not used) -- in ONE set of launches: the work queue holds
 * every patch once per frame, so a small tile share still fills the GPU and the latency chain of
 * a frame (pre-pass, phase-1 rounds, sort, the longest rays of phase 2) is paid once per batch.
 * tile_ids == NULL: whole frames, out_dev[n_frames][height][width][4]; else the tile subset like
 * vrhip_render_tiles, out_dev[n_frames][n_tiles][tile_h][tile_w][4]; out_frame_stride != 0 gives the
 * distance between the frames of out_dev in pixels (>= one frame).  DEVICE output only (the
 * renderer's own frame buffer is not meaningful afterwards).  Ray caster, iteration 0, no
 * image-order ESS, no ambient occlusion: anything else is VRHIP_ERR_UNSUPPORTED. */
int vrhip_render_batch(vrhip_renderer *r, uint32_t width, uint32_t height, uint32_t tile_w,
                       uint32_t tile_h, const uint32_t *tile_ids, uint32_t n_tiles,
                       const uint32_t *seeds, uint32_t n_frames, float *out_dev,
                       uint32_t out_frame_stride);
/* getLastExecTime (volumerendercl.cpp:1053-1056): HIP-event time of the last ray-cast
 * kernel launch, seconds. */
double vrhip_last_kernel_seconds(const vrhip_renderer *r);
/* The ray-cast pass is two back-to-back launches (budgeted march of every ray, then the
 * suspended long rays with 4 lanes per ray): HIP-event seconds of each. */
int vrhip_last_phase_seconds(const vrhip_renderer *r, double *phase1, double *phase2);
/* The event between the two phases costs a few microseconds of GPU time per frame (a barrier packet
 * between two launches): it is recorded only while this is enabled (default off; then
 * vrhip_last_phase_seconds answers VRHIP_ERR_NODATA). */
int vrhip_set_phase_timing(vrhip_renderer *r, int enabled);
/* The two events around a frame's launches (what vrhip_last_kernel_seconds reads; the reference times every frame
 * with CL profiling events, volumerendercl.cpp:545-551) cost GPU time of their own when frames follow each other
 * without a wait in between.  Default on; off: vrhip_last_kernel_seconds answers 0 and vrhip_last_phase_seconds
 * VRHIP_ERR_NODATA. */
int vrhip_set_frame_timing(vrhip_renderer *r, int enabled);

/* What the last render call launched (no reference counterpart; filled by the launchers themselves, not
 * re-derived): the tests and bench.py assert that the kernel instantiation and schedule they mean to check
 * are the ones that ran -- e.g. that the timed frames of the benchmark and the frames compared with the
 * oracle came out of the same kernels. */
typedef struct vrhip_launch_info {
    uint32_t technique;      /* 0 ray caster, 1 path tracer                                             */
    uint32_t frames;         /* frames of the launch set (vrhip_render_batch), else 1                   */
    uint32_t work_items;     /* 8x8 patches in the work queue, all frames                               */
    uint32_t prepass;        /* 1: vr_dda_prepass_kernel ran                                            */
    uint32_t ray_list;       /* 1: phase 1 = vr_raycast_rays_kernel on the pre-pass's ray list          */
    uint32_t phase1_waves;   /* waves per workgroup of the phase-1 kernel: 4, or 12 (three per SIMD)    */
    uint32_t phase2_waves;   /* the same for vr_raycast_split_kernel; 0: single phase                   */
    uint32_t round_budget;   /* phase-1 sample rounds per ray, 0 = single phase                         */
    uint32_t footprint;      /* 1: the kernels read the footprint volume                                */
    uint32_t empty_skip;     /* 1: the empty-run lookahead is on (cell grid handed to the kernels)      */
    uint32_t skip_in_lds;    /* 1: the ESS skip bitmap is staged in LDS (phase 1)                       */
    uint32_t instrumented;   /* 0 production kernels, 1 work counters, 2 / 3 + touched bitmap           */
    uint32_t extras;         /* 1: the variants with the rarer modes (illumType 2-5, AO, contours, ...) */
    uint32_t patch_classes;  /* 1: the pre-pass used per-patch classes                                  */
    uint32_t sorted_phase2;  /* 1: suspended rays were counting-sorted, longest first                   */
    uint32_t reserved[17];
} vrhip_launch_info;
/* VRHIP_ERR_NODATA before the first render call. */
int vrhip_last_launch_info(const vrhip_renderer *r, vrhip_launch_info *out);

/* The per-pixel cost map of the two-phase march (no reference counterpart; a schedule, not a result): for every
 * pixel of the last width x height frame the 4-lane rounds (16 samples each) its ray needed when it last reached
 * the 4-lane kernel -- the key suspended rays are sorted by.  n = width * height.  For tools (tools/costmap.py). */
int vrhip_download_cost_map(vrhip_renderer *r, uint16_t *out, size_t n);

/* ---- measurement helpers ------------------------------------------- */
/* When enabled, render calls run the instrumented kernel variant that accumulates
 * vrhip_stats (one atomic per wave). */
int vrhip_set_stats_enabled(vrhip_renderer *r, int enabled);
int vrhip_get_stats(const vrhip_renderer *r, vrhip_stats *out);
/* Untimed instrumentation pass over the current frame set-up: number of distinct
 * 4x4x4-voxel micro-bricks touched by at least one voxel fetch (compulsory traffic,
 * SURVEY 8d B_frame). Optionally returns the bitmap (ceil(res/4)^3 bits). */
int vrhip_count_touched(vrhip_renderer *r, uint32_t width, uint32_t height,
                        uint64_t *microbricks_touched, uint8_t *bitmap_host, size_t bitmap_bytes);
/* Path tracer (technique 1): the micro-bricks touched by the voxel fetches the PRODUCT issues -- tracking
 * steps the opacity bound of their cell cannot rule out (the others are rejections whatever the voxels
 * hold and are never fetched) plus the gradient at the primary interaction; speculative steps of a batch
 * excluded.  vrhip_count_touched counts the reference's fetch set (every step inside the volume).  The
 * difference is the traffic the culling removes; bench.py prices the kernel against THIS set. */
int vrhip_count_fetched(vrhip_renderer *r, uint32_t width, uint32_t height, uint64_t *microbricks_fetched);
/* Same for a tile subset (arguments as vrhip_render_tiles); also fills vrhip_get_stats. */
int vrhip_count_touched_tiles(vrhip_renderer *r, uint32_t width, uint32_t height,
                              uint32_t tile_w, uint32_t tile_h, const uint32_t *tile_ids,
                              uint32_t n_tiles, uint64_t *microbricks_touched);

/* Hash (16 hex digits) of the kernel sources this library was built from (the .hip and .h files of csrc/ and this header,
 * volumerenderercl_amd/_srchash.py): bench.py compares it with the tree's before it measures, so that a stale build
 * is never measured under the tree's name.  No reference counterpart. */
const char *vrhip_build_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* VRHIP_H */
