/*
 * vrhost.h -- C access to the host-side volume loader (libvrhost.so), the counterpart of the
 * reference's DatRawReader (/root/reference/src/io/datrawreader.h:38-184).  Pure host code:
 * no GPU needed.
 */
#ifndef VRHOST_H
#define VRHOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrdr vrdr;

typedef struct vrdr_info_t {
    uint32_t res[4];            /* x, y, z, time steps (Properties::volume_res)       */
    double thickness[3];        /* Properties::slice_thickness                        */
    int32_t format;             /* DatRawReader::data_format: 0 UCHAR 1 USHORT 2 FLOAT 3 DOUBLE */
    int32_t endianness;         /* 0 LITTLE, 1 BIG                                    */
    float min_value, max_value; /* Properties::min_value / max_value                  */
    uint64_t n_timesteps;
    uint64_t bytes_per_timestep;
    char channel_order[16];
} vrdr_info_t;

/* DatRawReader::read_files with dat_file_name = dat_file (raw_file: optional explicit .raw
 * name, i.e. Properties::raw_file_names = {raw_file}).  Returns 0, 1 (std::invalid_argument)
 * or 2 (std::runtime_error); vrdr_error() gives the exception text. */
int vrdr_load(const char *dat_file, const char *raw_file, vrdr **out);
const char *vrdr_error(void);
void vrdr_free(vrdr *h);
int vrdr_info(vrdr *h, vrdr_info_t *info);
/* DatRawReader::data()[t] -- normalised voxels as they go to the GPU */
const void *vrdr_data(vrdr *h, uint64_t t);
/* DatRawReader::getHistogram(t) */
int vrdr_histogram(vrdr *h, uint64_t t, double out[256]);


/* Radiance .hdr decoder behind createEnvironmentMap (reference: inc/hdr_loader.h:255-277,
 * load_hdr_float4): *pixels receives width*height*4 floats (RGB + alpha 0), to be released with
 * vrhost_free_pixels.  Returns 0, 1 (bad arguments) or 2 (file cannot be loaded). */
int vrhost_load_hdr(const char *file, float **pixels, uint32_t *width, uint32_t *height);
void vrhost_free_pixels(float *pixels);

/* Multi-GPU tile dealing of the headless host (csrc/host/tilegather.h; no reference counterpart: the
 * reference is single-GPU, volumerendercl.cpp:140): owner_out[t] = rank of tile t of the width x height
 * frame cut into tile x tile tiles (row-major, n_tiles = ceil(w / tile) * ceil(h / tile)); root_share: the
 * fraction of a peer's tiles rank 0 takes (1 = an equal share; it also assembles the frames).  Returns 0, or
 * 1 on bad arguments. */
int vrhost_deal_tiles(uint32_t width, uint32_t height, uint32_t tile, uint32_t ranks, double root_share,
                      uint32_t *owner_out, uint32_t n_tiles);

#ifdef __cplusplus
}
#endif
#endif
