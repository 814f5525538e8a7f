/*
 * vr_oracle.h -- CPU restatement of the VolumeRendererCL ray-cast hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (libvrhip.so) never links,
 * loads or calls it.
 *
 * Parity status: the integer RNG (random.cl) and the .dat/.raw loader are pinned
 * against the reference itself, compiled from /root/reference with no stand-ins
 * (oracle/Makefile, target `ref` -> oracle/_ref/).  The ray-march itself
 * (volumeraycast.cl:589-926) needs the OpenCL C builtin library (read_imagef &
 * friends), which this image lacks; a builtin shim would be a stand-in for that
 * library, so the kernel is treated as unbuildable here and its restatement is
 * "PARITY UNPINNED" beyond: the RNG/loader pins above, the known-answer values
 * recorded in SURVEY.md App. E (tests/golden/survey_kats.json) and analytic
 * closed forms (tests/test_oracle_analytic.py).  See DESIGN.md "Oracle".
 *
 * Struct layouts follow SURVEY.md App. D == reference
 * src/kernel/volumeraycast.cl:540-582 / src/core/volumerendercl.h:43-81.
 */
#ifndef VR_ORACLE_H
#define VR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VRO_UCHAR = 0, VRO_USHORT = 1, VRO_FLOAT = 2 };

/* volumeraycast.cl:543-550 (128 B, viewMat row-major) */
typedef struct {
    float viewMat[16];
    float bbox_bl[4];
    float bbox_tr[4];
    uint32_t ortho;
    uint32_t _pad[7];
} vro_camera_params;

/* volumeraycast.cl:552-567 (64 B) */
typedef struct {
    float backgroundColor[4];
    float modelScale[4]; /* float3 in a 16-byte slot */
    uint32_t illumType;
    uint32_t imgEss;
    uint32_t showEss;
    uint32_t useLinear;
    uint32_t useGradient;
    uint32_t technique;
    uint32_t seed;
    uint32_t iteration;
} vro_rendering_params;

/* volumeraycast.cl:569-577 (32 B) */
typedef struct {
    float samplingRate;
    uint32_t useAO;
    uint32_t contours;
    uint32_t aerial;
    float brickRes[4]; /* float3 in a 16-byte slot: volume_res / brick_edge */
} vro_raycast_params;

/* volumeraycast.cl:579-582 */
typedef struct {
    float max_extinction;
} vro_pathtrace_params;

/* Everything the kernel reads besides the four parameter structs. */
typedef struct {
    const void *voxels;      /* dense x-fastest array, type per `format` (CL_R image) */
    uint32_t res[3];
    int32_t format;          /* VRO_UCHAR / VRO_USHORT / VRO_FLOAT */
    const void *bricks;      /* (min,max) pairs in the volume's own type, x-fastest   */
    uint32_t bricks_res[3];  /* brick image dims; may be 0 when use_ess == 0          */
    const uint8_t *tff;      /* RGBA8 transfer function                               */
    uint32_t tff_n;          /* entries (reference: 1024)                             */
    const uint32_t *prefix;  /* inclusive prefix sum of the alpha bytes               */
    uint32_t prefix_n;
    uint32_t channels;       /* 0/1 = CL_R; 2 = CL_RG, 4 = CL_RGBA: interleaved voxels, the
                              * bricks are those of channel 0 (generateBricks reads .x)      */
} vro_scene;

typedef struct {
    uint64_t samples_taken;    /* executions of the inner loop body (:790-880)        */
    uint64_t samples_nominal;  /* sum over hit rays of ceil(sampleDist/stepSize)      */
    uint64_t samples_shaded;   /* samples for which the 6-fetch gradient fired        */
    uint64_t bricks_visited;   /* DDA iterations (brick reads)                        */
    uint64_t bricks_skipped;   /* of which skipped as empty                           */
    uint64_t rays_hit;         /* rays that pass the bbox test                        */
} vro_stats;

/* random.cl:2-13, :22-28, :44-47 */
uint32_t vro_parallel_rng(uint32_t x);
uint32_t vro_parallel_rng3(uint32_t x, uint32_t y, uint32_t z);
float vro_map_uint_float(uint32_t v);

/* hybrid Tausworthe / LCG generator behind calcAO (volumeraycast.cl:50-80) and the box-edge test
 * behind showEss (:323-343): exported for the pin against the compiled reference kernel file */
uint32_t vro_ui_rand_step(uint32_t st[4], uint32_t p, int s1, int s2, int s3, uint32_t m);
uint32_t vro_lcg_step(uint32_t st[4], uint32_t a, uint32_t c);
float vro_hybrid_rand(uint32_t st[4]);
int vro_check_bounding_box(const float pos[3], const float voxLen[3], float b0, float b1);

/* volumeraycast.cl:122-142. Returns hit flag. */
int vro_intersect_bbox(const float orig[3], const float dir[3], const float lower[3],
                       const float upper[3], float *tnear, float *tfar);

/* The parity definition of native_powr: a fixed sequence of
 * IEEE fp32 operations, reproduced verbatim by the HIP kernel. */
float vro_powr(float x, float y);

/* atan2 / acos as used by get_environment_coords (volumeraycast.cl:506-510): fixed fp32
 * sequences shared with the HIP kernel. */
float vro_atan2f(float y, float x);
float vro_acosf(float x);

/* volumerendercl.cpp:39-54 */
uint32_t vro_round_pow2(uint32_t n);
/* volumerendercl.cpp:620-636: brick edge (voxels), raycast.brickRes, brick image dims */
void vro_brick_layout(const uint32_t res[3], uint32_t edge[3], float brick_res_f[3],
                      uint32_t tex[3]);
/* volumerendercl.cpp:347-362 */
void vro_calc_scaling(const uint32_t res[3], const double thickness[3], float model_scale[3]);
/* volumerendercl.cpp:879-884 */
void vro_prefix_sum(const uint8_t *tff_rgba, uint32_t n, uint32_t *prefix);

/* generateBricks kernel, volumeraycast.cl:932-961. `out` holds 2*tex[0]*tex[1]*tex[2]
 * values of the volume's own type. */
/* downsampling kernel + host size rule (volumeraycast.cl:966-994, volumerendercl.cpp:245-251);
 * out must hold ceil(res/factor)^3 elements of the volume's type. */
int vro_downsample(const void *voxels, const uint32_t res[3], int format, int factor, void *out,
                   uint32_t out_res[3]);
int vro_generate_bricks(const void *voxels, const uint32_t res[3], int format,
                        const uint32_t tex[3], void *out);

/* Padded launch size, volumerendercl.cpp:513-514. */
uint32_t vro_padded(uint32_t n);

/*
 * volumeRender kernel, volumeraycast.cl:589-926, for the tile [x0,x0+w) x [y0,y0+h)
 * of a W x H frame.  `out` is w*h*4 floats (tile-local, row-major, RGBA), `in_accum`
 * (same shape, may be NULL when iteration == 0) is the previous accumulate image.
 * `touched`, if non-NULL, is a bitmap over 4x4x4-voxel micro-bricks
 * (ceil(res/4) per axis, x-fastest, 1 bit each) that receives every voxel fetch.
 * `threads` <= 0 uses all OpenMP threads.  Returns 0, or <0 on bad arguments.
 */
int vro_render_tile(const vro_scene *scene, const vro_camera_params *cam,
                    const vro_rendering_params *render, const vro_raycast_params *raycast,
                    const vro_pathtrace_params *pathtrace, int use_ess, uint32_t W, uint32_t H,
                    uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const float *in_accum,
                    float *out, vro_stats *stats, uint8_t *touched, int threads);

/* Inter-frame inputs of volumeRender beyond the accumulate image. */
typedef struct {
    const uint8_t *hit_in;  /* imgEss: last frame's hit image, (W/8+1)*(H/8+1) bytes, row-major   */
    uint8_t *hit_out;       /* this frame's; only the work-groups of the tile are (maybe) written */
    const float *env_rgba;  /* environment map, float RGBA, env_w*env_h texels; NULL = none       */
    uint32_t env_w, env_h;
} vro_frame_extras;

/* vro_render_tile plus image-order ESS (volumeraycast.cl:659-670, :912-925; needs a tile of whole
 * 8x8 work-groups) and the environment map (:506-510, :655-656).  The caller swaps hit_in/hit_out
 * between frames like runRaycast does (volumerendercl.cpp:524-530). */
int vro_render_tile_ex(const vro_scene *scene, const vro_camera_params *cam,
                       const vro_rendering_params *render, const vro_raycast_params *raycast,
                       const vro_pathtrace_params *pathtrace, int use_ess, uint32_t W, uint32_t H,
                       uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const float *in_accum,
                       float *out, vro_stats *stats, uint8_t *touched, int threads,
                       const vro_frame_extras *ex);
/* Initial contents of the two hit images as updateOutputImg leaves them
 * (volumerendercl.cpp:482-488), each (W/8+1)*(H/8+1) bytes. */
void vro_hit_image_init(uint32_t W, uint32_t H, uint8_t *hit_in, uint8_t *hit_out);

/* Synthetic inputs of SURVEY 8(d): kind 0 = sphere, 1 = shells. */
int vro_synth_volume(int kind, const uint32_t res[3], int format, void *out);

int vro_num_threads(void);

/* LITERAL mode (CPU tests only, see vr_oracle.c): evaluate gradients, image reads and the
 * transcendental builtins the way the reference's source text reads instead of through the
 * parity definitions shared with the HIP kernel.  Process-wide switch; not thread-safe
 * against a render in progress. */
void vro_set_literal(int on);
int vro_get_literal(void);

#ifdef __cplusplus
}
#endif
#endif
