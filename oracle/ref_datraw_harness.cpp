// ref_datraw_harness.cpp -- extern "C" harness around the REFERENCE's own DatRawReader
// (compiled from /root/reference/src/io/datrawreader.cpp where it lies; see Makefile
// target `ref`).  TEST INFRASTRUCTURE ONLY; exists only where /root/reference exists.
// Used by oracle/gen_loader_golden.py to produce tests/golden/loader/*.json.
#include "src/io/datrawreader.h"

#include <cstring>
#include <string>

extern "C" {

struct refdr_result {
    unsigned int res[4];
    double thickness[3];
    int format;
    int endianness;
    float min_value, max_value;
    unsigned long long n_timesteps;
    unsigned long long bytes_per_timestep;
    char channel_order[16];
};

static DatRawReader *g_reader = nullptr;
static std::string g_err;

// returns 0 on success; -1 with refdr_error() set when the reference throws
int refdr_load(const char *dat_file, refdr_result *out)
{
    try {
        delete g_reader;
        g_reader = new DatRawReader();
        DatRawReader::Properties p;
        p.dat_file_name = dat_file;
        g_reader->read_files(p);
        const DatRawReader::Properties &q = g_reader->properties();
        for (int i = 0; i < 4; ++i) out->res[i] = q.volume_res[i];
        for (int i = 0; i < 3; ++i) out->thickness[i] = q.slice_thickness[i];
        out->format = int(q.format);
        out->endianness = int(q.endianness);
        out->min_value = q.min_value;
        out->max_value = q.max_value;
        out->n_timesteps = g_reader->data().size();
        out->bytes_per_timestep = g_reader->data().front().size();
        std::memset(out->channel_order, 0, sizeof out->channel_order);
        std::strncpy(out->channel_order, q.image_channel_order.c_str(), 15);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

const char *refdr_error() { return g_err.c_str(); }

int refdr_copy_data(unsigned long long t, char *dst, unsigned long long n)
{
    if (!g_reader || !g_reader->has_data() || t >= g_reader->data().size()) return -1;
    const std::vector<char> &d = g_reader->data()[t];
    if (n > d.size()) n = d.size();
    std::memcpy(dst, d.data(), n);
    return 0;
}

int refdr_copy_histogram(unsigned long long t, double *dst)
{
    if (!g_reader || !g_reader->has_data()) return -1;
    const std::array<double, 256> &h = g_reader->getHistogram(t);
    std::memcpy(dst, h.data(), 256 * sizeof(double));
    return 0;
}
}
