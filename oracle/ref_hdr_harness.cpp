// ref_hdr_harness.cpp -- extern "C" harness around the REFERENCE's own Radiance .hdr loader
// (/root/reference/inc/hdr_loader.h, included where it lies; see Makefile target `ref`).
// TEST INFRASTRUCTURE ONLY; exists only where /root/reference exists.  Used by
// oracle/gen_hdr_golden.py to produce tests/golden/hdr/*.f32.
#include <cstdlib>   // the header calls malloc/calloc/free without including it

#include "inc/hdr_loader.h"

extern "C" {

// returns 1 on success (load_hdr_float4's own return value); *pixels must be freed with refhdr_free
int refhdr_load(const char *file, float **pixels, unsigned int *w, unsigned int *h)
{
    return load_hdr_float4(pixels, w, h, file) ? 1 : 0;
}

void refhdr_free(float *pixels) { free(pixels); }

}
