#include "hdrloader.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

namespace vrhost {
namespace {

struct Cursor {
    const std::vector<unsigned char> &b;
    size_t p = 0;
    explicit Cursor(const std::vector<unsigned char> &bytes) : b(bytes) {}
    int get() { return p < b.size() ? int(b[p++]) : -1; }
    // fgets semantics: up to and including '\n', at most 2047 characters; false at end of file
    bool line(std::string &out)
    {
        out.clear();
        if (p >= b.size()) return false;
        while (p < b.size() && out.size() < 2047) {
            const char c = char(b[p++]);
            out.push_back(c);
            if (c == '\n') break;
        }
        return true;
    }
    bool take(unsigned char *dst, size_t n)
    {
        if (b.size() - p < n) { p = b.size(); return false; }
        std::memcpy(dst, b.data() + p, n);
        p += n;
        return true;
    }
};

bool starts_with(const std::string &s, const char *prefix)
{
    return s.compare(0, std::strlen(prefix), prefix) == 0;
}

// the number after the first `axis` letter of the resolution line (hdr_loader.h:91-99; the
// letter must not be the first character, flips and axis order are ignored there too)
bool axis_size(const std::string &line, char axis, uint32_t &out)
{
    const size_t at = line.find(axis);
    if (at == std::string::npos || at == 0) return false;
    unsigned v = 0;
    if (std::sscanf(line.c_str() + at + 1, "%u", &v) != 1) return false;
    out = v;
    return true;
}

// hdr_loader.h:43-105
bool read_header(Cursor &in, uint32_t &w, uint32_t &h)
{
    std::string line;
    if (!in.line(line)) return false;
    // the reference rejects the first line only when neither of its first two characters matches
    if (line.size() > 1 ? (line[0] != '#' && line[1] != '?') : line[0] != '#') return false;
    for (;;) {
        if (!in.line(line)) return false;   // (the reference dereferences NULL here)
        if (line[0] == '#') continue;
        if (starts_with(line, "EXPOSURE=") || starts_with(line, "GAMMA=")) continue;   // parsed, unused
        if (starts_with(line, "FORMAT=")) {
            const std::string f = line.substr(7);
            if (!starts_with(f, "32-bit_rle_xyze") && !starts_with(f, "32-bit_rle_rgbe")) return false;
            continue;
        }
        if (line[0] == '-' || line[0] == '+') return axis_size(line, 'X', w) && axis_size(line, 'Y', h);
    }
}

bool flat_scanline(Cursor &in, unsigned char *rgbe, uint32_t n) { return in.take(rgbe, size_t(n) * 4); }

// hdr_loader.h:120-187
bool rle_scanline(Cursor &in, unsigned char *rgbe, uint32_t len)
{
    const int first = in.get();
    if (first < 0) return false;
    if (first != 2) {   // a flat scanline after all
        --in.p;
        return flat_scanline(in, rgbe, len);
    }
    const int g = in.get(), b = in.get(), c = in.get();
    rgbe[1] = (unsigned char)g;
    rgbe[2] = (unsigned char)b;
    if (rgbe[1] != 2 || (rgbe[2] & 128)) {   // flat, and its first pixel is already consumed
        rgbe[0] = 2;
        rgbe[3] = (unsigned char)c;
        return flat_scanline(in, rgbe + 4, len - 1);
    }
    if ((uint32_t(rgbe[2]) << 8 | uint32_t(c)) != len) return false;
    for (uint32_t comp = 0; comp < 4; ++comp) {
        uint32_t pos = 0;
        while (pos < len) {
            int count = in.get();
            if (count < 0) return false;
            if (count > 128) {   // run of one value
                count &= 127;
                const int value = in.get();
                if (value < 0 || pos + uint32_t(count) > len) return false;
                for (int j = 0; j < count; ++j) rgbe[(pos++) * 4 + comp] = (unsigned char)value;
            } else {             // literal values
                if (pos + uint32_t(count) > len) return false;
                for (int j = 0; j < count; ++j) {
                    const int value = in.get();
                    if (value < 0) return false;
                    rgbe[(pos++) * 4 + comp] = (unsigned char)value;
                }
            }
        }
    }
    return true;
}

// hdr_loader.h:190-209: (mantissa + 0.5) * 2^(exponent - 136), exponent 0 = black
void rgbe_to_rgb(const unsigned char *rgbe, float *rgb)
{
    if (rgbe[3] == 0) {
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
        return;
    }
    const uint32_t bits = (uint32_t(int(rgbe[3]) - 9) << 23) & 0x7f800000u;
    float scale;
    std::memcpy(&scale, &bits, sizeof scale);
    for (int i = 0; i < 3; ++i) rgb[i] = (float(rgbe[i]) + 0.5f) * scale;
}

} // namespace

bool load_hdr_float4(const std::string &file_name, HdrImage &out)
{
    std::ifstream f(file_name, std::ios::binary);
    if (!f) return false;
    const std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(f)),
                                           std::istreambuf_iterator<char>());
    Cursor in(bytes);
    uint32_t w = 0, h = 0;
    if (!read_header(in, w, h)) return false;
    if (w == 0 || h == 0 || uint64_t(w) * h > (1ull << 28)) return false;
    // very short and very long scanlines cannot be run-length coded (hdr_loader.h:217-218)
    const bool rle = !(w < 8 || w > 0x7fff);
    std::vector<float> px(size_t(w) * h * 4, 0.0f);
    std::vector<unsigned char> row(size_t(w) * 4);
    for (uint32_t y = 0; y < h; ++y) {
        if (!(rle ? rle_scanline(in, row.data(), w) : flat_scanline(in, row.data(), w))) return false;
        for (uint32_t x = 0; x < w; ++x) rgbe_to_rgb(&row[size_t(x) * 4], &px[(size_t(y) * w + x) * 4]);
    }
    out.width = w;
    out.height = h;
    out.rgba.swap(px);
    return true;
}

} // namespace vrhost
