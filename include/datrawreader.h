// datrawreader.h -- .dat/.raw volume loader with the interface of the reference's
// DatRawReader (/root/reference/src/io/datrawreader.h:38-184): same nested types, member
// names, exceptions and value-changing side effects (USHORT stretch to 65535, FLOAT division
// by the maximum, 256-bin histogram), so that "the same .dat/.raw input"
// reaches the renderer as the same bytes.  Written from the behaviour, not from the source.
#pragma once

#include <array>
#include <limits>
#include <string>
#include <vector>

class DatRawReader
{
public:
    enum data_format { UCHAR = 0, USHORT, FLOAT, DOUBLE, UNKNOWN_FORMAT };
    enum data_endianness { LITTLE = 0, BIG };

    struct Properties {
        std::string dat_file_name = "";
        std::vector<std::string> raw_file_names;
        size_t raw_file_size = 0;

        std::array<unsigned int, 4> volume_res = {{0, 0, 0, 1}};   // x, y, z, t
        std::array<double, 3> slice_thickness = {{1.0, 1.0, 1.0}};
        data_format format = UNKNOWN_FORMAT;
        data_endianness endianness = LITTLE;
        std::string node_file_name = "";
        std::string image_channel_order = "R";
        unsigned int time_series = {1u};
        float min_value = std::numeric_limits<float>::max();
        float max_value = std::numeric_limits<float>::min();

        const std::string to_string() const;
        const std::string get_format_string(const enum data_format f) const;
    };

    // Reads the .dat description (unless raw_file_names is preset) and every time step.
    // Throws std::invalid_argument for empty names, std::runtime_error for I/O problems.
    void read_files(Properties volume_properties);
    bool has_data() const;
    const std::vector<std::vector<char>> &data() const;   // throws when empty
    const Properties &properties() const;                 // throws when empty
    void clearData();
    const std::array<double, 256> &getHistogram(size_t timestep = 0);

private:
    void infer_volume_resolution(unsigned long long file_size);
    void read_dat(const std::string &dat_file_name);
    void read_raw(const std::string &raw_file_name);

    Properties _prop;
    std::vector<std::vector<char>> _raw_data;
    std::vector<std::array<double, 256>> _histograms;
};
