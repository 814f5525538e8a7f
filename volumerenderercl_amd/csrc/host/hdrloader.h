// hdrloader.h -- Radiance RGBE (.hdr) decoder of the host layer, behind createEnvironmentMap.
// Restates what the reference's loader does with a file (/root/reference/inc/hdr_loader.h:43-277,
// used at src/core/volumerendercl.cpp:1136-1141): header lines up to the resolution line, flat or
// run-length coded scanlines, RGBE -> float RGB with alpha 0.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace vrhost {

struct HdrImage {
    uint32_t width = 0, height = 0;
    std::vector<float> rgba;   // width * height * 4, row-major, alpha = 0 (hdr_loader.h:268)
};

// false when the file cannot be opened, the header is not a Radiance header or the pixel data
// is truncated / inconsistent (hdr_loader.h:255-277 returns false in the same cases).
bool load_hdr_float4(const std::string &file_name, HdrImage &out);

} // namespace vrhost
