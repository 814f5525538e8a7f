// vrhost_capi.cpp -- plain-C access to the host-side DatRawReader (libvrhost.so), used by the
// Python package (volumerenderercl_amd/datraw.py) and its tests.  Declared in include/vrhost.h.
#include "hdrloader.h"
#include <cstdlib>
#include <cstring>
#include "vrhost.h"

#include <cstring>
#include <stdexcept>
#include <string>

#include "datrawreader.h"

struct vrdr {
    DatRawReader reader;
};

static thread_local std::string g_error;

extern "C" {

int vrdr_load(const char *dat_file, const char *raw_file, vrdr **out)
{
    if (!out) return 1;
    *out = nullptr;
    vrdr *h = new vrdr();
    try {
        DatRawReader::Properties p;
        if (dat_file) p.dat_file_name = dat_file;
        if (raw_file && *raw_file) p.raw_file_names.push_back(raw_file);
        h->reader.read_files(p);
    } catch (const std::invalid_argument &e) {
        g_error = e.what();
        delete h;
        return 1;   // std::invalid_argument
    } catch (const std::exception &e) {
        g_error = e.what();
        delete h;
        return 2;   // std::runtime_error
    }
    *out = h;
    return 0;
}

const char *vrdr_error(void) { return g_error.c_str(); }

void vrdr_free(vrdr *h) { delete h; }

int vrdr_info(vrdr *h, vrdr_info_t *info)
{
    if (!h || !info || !h->reader.has_data()) return 1;
    const DatRawReader::Properties &p = h->reader.properties();
    for (int i = 0; i < 4; ++i) info->res[i] = p.volume_res[i];
    for (int i = 0; i < 3; ++i) info->thickness[i] = p.slice_thickness[i];
    info->format = int(p.format);
    info->endianness = int(p.endianness);
    info->min_value = p.min_value;
    info->max_value = p.max_value;
    info->n_timesteps = h->reader.data().size();
    info->bytes_per_timestep = h->reader.data().front().size();
    std::memset(info->channel_order, 0, sizeof info->channel_order);
    std::strncpy(info->channel_order, p.image_channel_order.c_str(), sizeof info->channel_order - 1);
    return 0;
}

const void *vrdr_data(vrdr *h, uint64_t t)
{
    if (!h || !h->reader.has_data() || t >= h->reader.data().size()) return nullptr;
    return h->reader.data()[t].data();
}

int vrhost_load_hdr(const char *file, float **pixels, uint32_t *width, uint32_t *height)
{
    if (!file || !pixels || !width || !height) return 1;
    vrhost::HdrImage img;
    if (!vrhost::load_hdr_float4(file, img)) return 2;
    float *p = static_cast<float *>(std::malloc(img.rgba.size() * sizeof(float)));
    if (!p) return 2;
    std::memcpy(p, img.rgba.data(), img.rgba.size() * sizeof(float));
    *pixels = p;
    *width = img.width;
    *height = img.height;
    return 0;
}

void vrhost_free_pixels(float *pixels) { std::free(pixels); }

int vrdr_histogram(vrdr *h, uint64_t t, double out[256])
{
    if (!h || !h->reader.has_data() || t >= h->reader.data().size()) return 1;
    const std::array<double, 256> &hist = h->reader.getHistogram(t);
    std::memcpy(out, hist.data(), sizeof(double) * 256);
    return 0;
}
}

// ---- tile dealing of the multi-GPU driver (tilegather.h), exported for the CPU test against tiles.py
#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#include "tilegather.h"

std::vector<unsigned int> vr_deal_tiles(size_t width, size_t height, size_t tile, size_t n, double root_share)
{
    const size_t tiles_x = (width + tile - 1) / tile, tiles_y = (height + tile - 1) / tile, nt = tiles_x * tiles_y;
    std::vector<unsigned int> owner(nt, 0u);
    if (n <= 1) return owner;
    std::vector<std::pair<long long, unsigned int>> order(nt);
    for (size_t t = 0; t < nt; ++t) {
        const long long dx = (2 * static_cast<long long>(t % tiles_x) + 1) * static_cast<long long>(tile) - static_cast<long long>(width);
        const long long dy = (2 * static_cast<long long>(t / tiles_x) + 1) * static_cast<long long>(tile) - static_cast<long long>(height);
        order[t] = {dx * dx + dy * dy, static_cast<unsigned int>(t)};
    }
    std::sort(order.begin(), order.end());
    // rounds of the deal 0 1 .. n-1 n-1 .. 1 0; of its two cards per round rank 0 takes 2 * root_share on
    // average (error diffusion, first / last card alternating) -- tiles.py deal_tiles, operation for operation
    const double share = std::min(1.0, std::max(0.0, root_share));
    std::vector<unsigned int> seq;
    seq.reserve(nt + 2 * n);
    double acc = 0.0;
    for (size_t cycle = 0; seq.size() < nt; ++cycle) {
        acc += 2.0 * share;
        const int take = static_cast<int>(std::floor(acc + 1e-9));
        acc -= take;
        const bool first = take == 2 || (take == 1 && cycle % 2 == 0);
        const bool last = take == 2 || (take == 1 && cycle % 2 == 1);
        if (first) seq.push_back(0u);
        for (size_t r = 1; r < n; ++r) seq.push_back(static_cast<unsigned int>(r));
        for (size_t r = n - 1; r >= 1; --r) seq.push_back(static_cast<unsigned int>(r));
        if (last) seq.push_back(0u);
    }
    for (size_t i = 0; i < nt; ++i) owner[order[i].second] = seq[i];
    return owner;
}

extern "C" int vrhost_deal_tiles(uint32_t width, uint32_t height, uint32_t tile, uint32_t ranks, double root_share,
                                 uint32_t *owner_out, uint32_t n_tiles)
{
    if (!owner_out || !tile || !ranks || !width || !height) return 1;
    const std::vector<unsigned int> owner = vr_deal_tiles(width, height, tile, ranks, root_share);
    if (owner.size() != n_tiles) return 1;
    std::copy(owner.begin(), owner.end(), owner_out);
    return 0;
}
