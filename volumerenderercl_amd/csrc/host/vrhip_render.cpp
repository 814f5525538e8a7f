// vrhip_render -- headless host: loads a .dat/.raw volume (or a synthetic field), sets camera
// and transfer function the way the reference's Qt widget does
// (/root/reference/src/qt/volumerenderwidget.cpp:916-938 TF table, :1079-1098 view matrix),
// renders through VolumeRenderCL and writes the float RGBA framebuffer to disk.  Replaces the
// Qt/OpenGL front end dropped by the north star.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <map>
#include <cctype>
#include <iterator>
#include <string>
#include <vector>

#include <memory>
#include <chrono>

#include <hip/hip_runtime_api.h>

#include "tilegather.h"
#include "volumerendercl.h"

namespace {

struct Stop { double pos; int c[4]; };

// QEasingCurve::valueForProgress for the three curves the GUI offers (mainwindow.cpp:945-957), with the
// operation order of Qt's src/3rdparty/easing/easing.cpp (easeNone, easeInOutQuad, easeInOutCubic)
double ease(const std::string &kind, double t)
{
    t = std::min(1.0, std::max(0.0, t));
    if (kind == "quad") {
        t *= 2.0;
        if (t < 1) return t * t / 2.0;
        --t;
        return -0.5 * (t * (t - 2) - 1);
    }
    if (kind == "cubic") {
        t *= 2.0;
        if (t < 1) return 0.5 * t * t * t;
        t -= 2.0;
        return 0.5 * (t * t * t + 2);
    }
    return t;
}

// updateTransferFunction (volumerenderwidget.cpp:916-938): 1024 entries, entry i = the key-value
// animation over the stops at time qRound(i/1024*8192) of 8192; QVariantAnimation interpolates QColor
// per channel as qBound(0, int(f + (t - f) * localProgress), 255) with localProgress = (progress -
// start) / (end - start) in double; then max(0, c - 3).  setKeyValueAt replaces an earlier key at the
// same position; missing end stops (the GUI always has them) repeat the nearest stop's colour.
std::vector<unsigned char> tff_from_stops(std::vector<Stop> in, int n = 1024, const std::string &easing = "linear")
{
    std::stable_sort(in.begin(), in.end(), [](const Stop &a, const Stop &b) { return a.pos < b.pos; });
    std::vector<Stop> stops;
    for (const Stop &s : in) {
        if (!stops.empty() && stops.back().pos == s.pos) stops.back() = s;
        else stops.push_back(s);
    }
    if (stops.front().pos > 0.0) { Stop s = stops.front(); s.pos = 0.0; stops.insert(stops.begin(), s); }
    if (stops.back().pos < 1.0) { Stop s = stops.back(); s.pos = 1.0; stops.push_back(s); }
    std::vector<unsigned char> out(size_t(n) * 4, 0);
    for (int i = 0; i < n; ++i) {
        const double time = std::floor(double(i) / n * 8192.0 + 0.5);
        const double p = ease(easing, time / 8192.0);
        size_t k = 0;
        while (k + 2 < stops.size() && p >= stops[k + 1].pos) ++k;
        const Stop &a = stops[k], &b = stops[k + 1];
        const double lp = b.pos == a.pos ? 0.0 : (p - a.pos) / (b.pos - a.pos);
        for (int c = 0; c < 4; ++c) {
            int v = int(a.c[c] + (b.c[c] - a.c[c]) * lp);
            v = std::min(255, std::max(0, v));
            out[size_t(i) * 4 + c] = static_cast<unsigned char>(std::max(0, v - 3));
        }
    }
    return out;
}

// raw TF text file: 4096 whitespace-separated integers (mainwindow.cpp:680-719)
std::vector<unsigned char> tff_from_raw_file(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Could not open transfer function file " + path);
    std::vector<unsigned char> out;
    double v;
    while (in >> v) out.push_back(static_cast<unsigned char>(std::min(255.0, std::max(0.0, v))));
    if (out.empty() || out.size() % 4) throw std::runtime_error("Invalid transfer function file " + path);
    return out;
}

// `.tff` gradient-stop file (MainWindow::readTff, mainwindow.cpp:583-620): `pos r g b a` per line
std::vector<unsigned char> tff_from_stops_file(const std::string &path, const std::string &easing)
{
    std::ifstream in(path);
    if (!in) throw std::invalid_argument("Could not open transfer function file " + path);
    std::vector<Stop> stops;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        double v[5];
        int n = 0;
        while (n < 5 && (ls >> v[n])) ++n;
        if (n < 5) continue;
        Stop st;
        st.pos = v[0];
        for (int c = 0; c < 4; ++c) st.c[c] = int(v[1 + c]);
        stops.push_back(st);
    }
    if (stops.empty()) throw std::invalid_argument("Empty transfer function file.");
    return tff_from_stops(stops, 1024, easing);
}

// The GUI's JSON state (MainWindow::loadCamState, mainwindow.cpp:374-410; VolumeRenderWidget::read,
// volumerenderwidget.cpp:1457-1480): a flat object of strings, numbers and booleans.
struct CamState {
    bool has_rot = false, has_tr = false;
    double q[4] = {1, 0, 0, 0}, t[3] = {0, 0, 2};
    std::map<std::string, double> num;
    std::map<std::string, bool> flag;
};

CamState read_cam_state(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::invalid_argument("Couldn't open state file " + path);
    std::string js((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    CamState st;
    size_t i = 0;
    auto skip = [&]() { while (i < js.size() && std::isspace((unsigned char)js[i])) ++i; };
    auto str = [&]() {
        std::string out;
        ++i;   // opening quote
        while (i < js.size() && js[i] != '"') {
            if (js[i] == '\\' && i + 1 < js.size()) ++i;
            out += js[i++];
        }
        ++i;
        return out;
    };
    skip();
    if (i >= js.size() || js[i] != '{') throw std::invalid_argument("Invalid state file " + path);
    ++i;
    for (;;) {
        skip();
        if (i >= js.size() || js[i] == '}') break;
        if (js[i] == ',') { ++i; continue; }
        if (js[i] != '"') throw std::invalid_argument("Invalid state file " + path);
        const std::string key = str();
        skip();
        if (i < js.size() && js[i] == ':') ++i;
        skip();
        if (i < js.size() && js[i] == '"') {
            std::istringstream vs(str());
            if (key == "camRotation") { st.has_rot = bool(vs >> st.q[0] >> st.q[1] >> st.q[2] >> st.q[3]); }
            else if (key == "camTranslation") { st.has_tr = bool(vs >> st.t[0] >> st.t[1] >> st.t[2]); }
        } else if (js.compare(i, 4, "true") == 0) { st.flag[key] = true; i += 4; }
        else if (js.compare(i, 5, "false") == 0) { st.flag[key] = false; i += 5; }
        else {
            size_t used = 0;
            st.num[key] = std::stod(js.substr(i), &used);
            i += used;
        }
    }
    return st;
}

// updateViewMatrix (volumerenderwidget.cpp:1079-1098): M = R(q) T(t) S(t.z), row-major
std::array<float, 16> view_matrix(const double q[4], const double t[3])
{
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                      {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                      {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    std::array<float, 16> m{};
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[r * 4 + c] = float(R[r][c] * t[2]);
        m[r * 4 + 3] = float(R[r][0] * t[0] + R[r][1] * t[1] + R[r][2] * t[2]);
    }
    m[15] = 1.f;
    return m;
}

void write_ppm(const std::string &path, const std::vector<float> &rgba, size_t w, size_t h)
{
    std::ofstream f(path, std::ios::binary);
    f << "P6\n" << w << " " << h << "\n255\n";
    std::vector<unsigned char> row(w * 3);
    for (size_t y = 0; y < h; ++y) {
        for (size_t x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                float v = rgba[(y * w + x) * 4 + c];
                row[x * 3 + c] = static_cast<unsigned char>(std::lround(std::min(1.f, std::max(0.f, v)) * 255.f));
            }
        f.write(reinterpret_cast<const char *>(row.data()), std::streamsize(row.size()));
    }
}

[[noreturn]] void usage()
{
    std::cerr <<
        "usage: vrhip_render (--dat FILE.dat | --synth sphere|shells N UCHAR|USHORT|FLOAT)\n"
        "         [--size W H] [--rotate AX AY AZ DEG] [--translate X Y Z] [--view M0..M15]\n"
        "         [--tf default|FILE] [--illum N] [--no-ess] [--ortho] [--nearest] [--rate R]\n"
        "         [--bg R G B] [--gradient-bg] [--seed S] [--frames N] [--device D] --out PREFIX\n"
        "         [--pathtrace] [--extinction E]   (technique 1; --frames = samples per pixel)\n"
        "         [--downsample FACTOR]            (volumeDownsampling: writes <dat>_<N>.raw/.dat, no frame)\n"
        "         [--state FILE.json] [--tf-stops FILE.tff]   (files saved by the reference GUI)\n"
        "         [--tf-easing linear|quad|cubic] [--dump-tf FILE]   (interpolation between the stops; --dump-tf\n"
        "                                           writes the RGBA8 table the options select and exits, no GPU)\n"
        "         [--contours] [--aerial] [--ao] [--show-ess] [--img-ess]   (--img-ess: state carried over --frames)\n"
        "         [--env FILE.hdr]                 (createEnvironmentMap: Radiance RGBE environment map)\n"
        "         [--ranks N [--tile T] [--loopback] [--force-gather]]   (image tiles over N GPUs, devices D .. D+N-1,\n"
        "                                           volume replicated, RCCL gather to the first; --loopback: N\n"
        "                                           renderers on device D, device copies instead of RCCL;\n"
        "                                           --force-gather: the root sends its tiles to itself over RCCL\n"
        "                                           too -- with --ranks 1 the transport runs on one GPU)\n"
        "         [--independent]                  (--frames N independent frames -- iteration 0 each, jitter seeds = the\n"
        "                                           first N outputs of the default-seeded mt19937 -- instead of the\n"
        "                                           running mean; all of them go to PREFIX.frames.rgba.f32)\n"
        "         [--frames-per-launch K [--frames-in-flight F] [--round-budget B] [--bench] [--root-share X]]\n"
        "                                          (the throughput path: the N independent frames in launch sets of K\n"
        "                                           (vrhip_render_batch), F renderers over one shared volume taking the\n"
        "                                           sets in turn (default 2; with --ranks: one per rank, K frames per\n"
        "                                           exchange, sparse messages, one exchange in flight), phase-1 round\n"
        "                                           budget B (default 48); --bench: one untimed set per renderer, then\n"
        "                                           the N frames timed with HIP events, nothing copied or written but\n"
        "                                           the last frame; --root-share: rank 0's share of a peer's tiles)\n"
        "writes PREFIX.rgba.f32 (W*H*4 float32, row 0 = top), PREFIX.ppm and prints one JSON line\n";
    std::exit(2);
}

} // namespace

int main(int argc, char **argv)
{
    std::string dat, synth_kind, synth_fmt = "UCHAR", tf = "default", out, env_file;
    unsigned synth_n = 0;
    size_t W = 1024, H = 1024;
    double q[4] = {1, 0, 0, 0}, tr[3] = {0, 0, 2};
    bool have_view = false, ess = true, ortho = false, linear = true, gradient_bg = false, pin = false;
    bool pathtrace = false;
    double extinction = 100.0;
    int downsample = 0;
    std::string state_file, tf_stops, tf_easing = "linear", dump_tf;
    bool contours = false, aerial = false, use_ao_flag = false, show_ess_flag = false, img_ess = false;
    std::array<float, 16> view{};
    unsigned illum = 1, seed = 0;
    int frames = 1, device = 0, ranks = 0;
    size_t tile = 64;
    bool loopback = false, force_gather = false;
    bool independent = false, bench = false;
    int frames_per_launch = 0, frames_in_flight = 2, round_budget = 48;
    double root_share = 1.0;
    double rate = 1.5;
    std::array<float, 4> bg = {{1, 1, 1, 1}};

    auto need = [&](int i, int n) { if (i + n >= argc) usage(); };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--dat") { need(i, 1); dat = argv[++i]; }
        else if (a == "--synth") { need(i, 3); synth_kind = argv[++i]; synth_n = unsigned(std::atoi(argv[++i])); synth_fmt = argv[++i]; }
        else if (a == "--size") { need(i, 2); W = size_t(std::atol(argv[++i])); H = size_t(std::atol(argv[++i])); }
        else if (a == "--rotate") {
            need(i, 4);
            double ax = std::atof(argv[++i]), ay = std::atof(argv[++i]), az = std::atof(argv[++i]);
            double half = std::atof(argv[++i]) * M_PI / 360.0, l = std::sqrt(ax * ax + ay * ay + az * az);
            q[0] = std::cos(half); q[1] = ax / l * std::sin(half); q[2] = ay / l * std::sin(half); q[3] = az / l * std::sin(half);
        }
        else if (a == "--translate") { need(i, 3); for (int k = 0; k < 3; ++k) tr[k] = std::atof(argv[++i]); }
        else if (a == "--view") { need(i, 16); for (int k = 0; k < 16; ++k) view[size_t(k)] = float(std::atof(argv[++i])); have_view = true; }
        else if (a == "--tf") { need(i, 1); tf = argv[++i]; }
        else if (a == "--illum") { need(i, 1); illum = unsigned(std::atoi(argv[++i])); }
        else if (a == "--no-ess") ess = false;
        else if (a == "--ortho") ortho = true;
        else if (a == "--nearest") linear = false;
        else if (a == "--gradient-bg") gradient_bg = true;
        else if (a == "--rate") { need(i, 1); rate = std::atof(argv[++i]); }
        else if (a == "--bg") { need(i, 3); for (int k = 0; k < 3; ++k) bg[size_t(k)] = float(std::atof(argv[++i])); }
        else if (a == "--seed") { need(i, 1); seed = unsigned(std::strtoul(argv[++i], nullptr, 10)); pin = true; }
        else if (a == "--frames") { need(i, 1); frames = std::atoi(argv[++i]); }
        else if (a == "--pathtrace") pathtrace = true;
        else if (a == "--downsample") { need(i, 1); downsample = std::atoi(argv[++i]); }
        else if (a == "--state") { need(i, 1); state_file = argv[++i]; }
        else if (a == "--tf-stops") { need(i, 1); tf_stops = argv[++i]; }
        else if (a == "--tf-easing") { need(i, 1); tf_easing = argv[++i]; }
        else if (a == "--dump-tf") { need(i, 1); dump_tf = argv[++i]; }
        else if (a == "--contours") contours = true;
        else if (a == "--aerial") aerial = true;
        else if (a == "--ao") use_ao_flag = true;
        else if (a == "--show-ess") show_ess_flag = true;
        else if (a == "--img-ess") img_ess = true;
        else if (a == "--env") { need(i, 1); env_file = argv[++i]; }
        else if (a == "--extinction") { need(i, 1); extinction = std::atof(argv[++i]); }
        else if (a == "--device") { need(i, 1); device = std::atoi(argv[++i]); }
        else if (a == "--ranks") { need(i, 1); ranks = std::atoi(argv[++i]); }
        else if (a == "--tile") { need(i, 1); tile = size_t(std::atol(argv[++i])); }
        else if (a == "--loopback") loopback = true;
        else if (a == "--force-gather") force_gather = true;
        else if (a == "--independent") independent = true;
        else if (a == "--bench") bench = true;
        else if (a == "--frames-per-launch") { need(i, 1); frames_per_launch = std::atoi(argv[++i]); }
        else if (a == "--frames-in-flight") { need(i, 1); frames_in_flight = std::atoi(argv[++i]); }
        else if (a == "--round-budget") { need(i, 1); round_budget = std::atoi(argv[++i]); }
        else if (a == "--root-share") { need(i, 1); root_share = std::atof(argv[++i]); }
        else if (a == "--out") { need(i, 1); out = argv[++i]; }
        else usage();
    }
    if (tf_easing != "linear" && tf_easing != "quad" && tf_easing != "cubic") usage();
    // the transfer-function table the options select (TransferFunctionWidget's default stops,
    // transferfunctionwidget.cpp:338-346, unless a file is given)
    auto make_table = [&]() -> std::vector<unsigned char> {
        return !tf_stops.empty() ? tff_from_stops_file(tf_stops, tf_easing)
               : tf == "default" ? tff_from_stops({{0.0, {0, 0, 0, 0}}, {0.1, {125, 125, 125, 0}}, {1.0, {0, 0, 0, 255}}},
                                                  1024, tf_easing)
                                 : tff_from_raw_file(tf);
    };
    if (!dump_tf.empty()) {   // front-end formula only: nothing below touches a GPU
        try {
            const std::vector<unsigned char> table = make_table();
            std::ofstream f(dump_tf, std::ios::binary);
            f.write(reinterpret_cast<const char *>(table.data()), std::streamsize(table.size()));
            return f ? 0 : 1;
        } catch (const std::exception &e) {
            std::cerr << e.what() << std::endl;
            return 1;
        }
    }
    if ((dat.empty() && synth_kind.empty()) || (out.empty() && !downsample)) usage();

    try {
        // everything the front end sets on a renderer; returns false when there is no frame to render
        auto setup = [&](VolumeRenderCL &vr, int dev) -> bool {
        vr.initialize(false, false, VENDOR_ANY, std::to_string(dev));
        if (!dat.empty()) {
            DatRawReader::Properties p;
            p.dat_file_name = dat;
            vr.loadVolumeData(p);
        } else {
            DatRawReader::data_format f = synth_fmt == "USHORT" ? DatRawReader::USHORT
                                          : synth_fmt == "FLOAT" ? DatRawReader::FLOAT : DatRawReader::UCHAR;
            vr.loadSyntheticVolume(synth_kind, synth_n, f);
        }
        if (downsample) {   // volumeDownsampling (volumerendercl.cpp:238-341) and nothing else
            const std::string base = vr.volumeDownsampling(0, downsample);
            std::printf("{\"downsampled\": \"%s\"}\n", base.c_str());
            return false;
        }
        bool use_ao = use_ao_flag, show_box = show_ess_flag;
        if (!state_file.empty()) {   // what the GUI's widgets would forward after loadCamState
            const CamState st = read_cam_state(state_file);
            if (st.has_rot) for (int k = 0; k < 4; ++k) q[k] = st.q[k];
            if (st.has_tr) for (int k = 0; k < 3; ++k) tr[k] = st.t[k];
            if (st.num.count("rayStepSize")) rate = st.num.at("rayStepSize");
            if (st.num.count("imgResFactor")) {
                W = size_t(std::floor(double(W) * st.num.at("imgResFactor")));
                H = size_t(std::floor(double(H) * st.num.at("imgResFactor")));
            }
            if (st.flag.count("useLerp")) linear = st.flag.at("useLerp");
            if (st.flag.count("useOrtho")) ortho = st.flag.at("useOrtho");
            if (st.flag.count("showContours")) contours = st.flag.at("showContours");
            if (st.flag.count("useAerial")) aerial = st.flag.at("useAerial");
            if (st.flag.count("useAO")) use_ao = st.flag.at("useAO");
            if (st.flag.count("showBox")) show_box = st.flag.at("showBox");
        }
        std::vector<unsigned char> table = make_table();
        vr.setTransferFunction(table);
        vr.setIllumination(illum);
        vr.setObjEss(ess);
        vr.setCamOrtho(ortho);
        vr.setLinearInterpolation(linear);
        vr.setUseGradient(gradient_bg);
        vr.setContours(contours);
        vr.setAerial(aerial);
        vr.setAmbientOcclusion(use_ao);
        if (show_box) vr.setShowESS(true);
        if (!env_file.empty()) vr.createEnvironmentMap(env_file);
        if (img_ess) vr.setImgEss(true);   // state carried from frame to frame (--frames N)
        vr.setBackground(bg);
        vr.updateSamplingRate(rate);
        if (pathtrace) {
            vr.setTechnique(VolumeRenderCL::TECH_PATHTRACE);
            vr.setExtinction(extinction);
        }
        if (pin) vr.setSeed(seed);
        vr.updateOutputImg(W, H, 0);
        vr.updateView(have_view ? view : view_matrix(q, tr));
        return true;
        };

        if (ranks > 0) {
            // image-tile decomposition over `ranks` GPUs: one renderer per rank, every
            // one with the whole volume and the same settings; jitter seeds follow the same
            // default-seeded mt19937 sequence in every renderer, frame by frame
            if (img_ess) throw std::runtime_error("image-order ESS state crosses tile borders: not with --ranks");
            std::vector<std::unique_ptr<VolumeRenderCL>> vrs;
            std::vector<VolumeRenderCL *> ptrs;
            std::vector<int> devs;
            for (int r = 0; r < ranks; ++r) {
                vrs.emplace_back(new VolumeRenderCL());
                devs.push_back(loopback ? device : device + r);
                if (!setup(*vrs.back(), devs.back())) return 0;
                ptrs.push_back(vrs.back().get());
            }
            const bool batched = frames_per_launch > 0;
            if (batched && (pathtrace || img_ess || use_ao_flag))
                throw std::runtime_error("--frames-per-launch: ray caster only, no image-order ESS, no ambient occlusion");
            const size_t K = batched ? size_t(std::min(256, std::max(1, frames_per_launch))) : 0;
            TileGather tg(ptrs, devs, W, H, tile, loopback, force_gather, root_share, K);
            std::vector<float> frame, all_frames;
            double secs = 0.0;
            const std::array<float, 16> the_view = have_view ? view : view_matrix(q, tr);
            if (batched) {
                // the throughput path: K independent frames per exchange, one exchange in flight
                for (auto &v : vrs) { v->setRoundBudget(unsigned(round_budget)); v->setFrameTiming(false); }
                std::vector<std::vector<unsigned int>> sets;
                std::vector<unsigned int> seeds = vrs[0]->drawSeeds(size_t(frames));
                for (size_t f0 = 0; f0 < seeds.size(); f0 += K)
                    sets.emplace_back(seeds.begin() + long(f0), seeds.begin() + long(std::min(seeds.size(), f0 + K)));
                if (bench) {   // untimed: buffers, work queues, cost maps, communicators
                    std::mt19937 warm(20261004u);
                    std::vector<unsigned int> ws(std::min(K, size_t(frames)));
                    for (int rep = 0; rep < 2; ++rep) {
                        for (auto &x : ws) x = static_cast<unsigned int>(warm());
                        tg.submitFrames(ws);
                    }
                    tg.collectFrames(nullptr);
                    tg.collectFrames(nullptr);
                }
                const auto t0 = std::chrono::steady_clock::now();
                for (size_t k = 0; k < sets.size(); ++k) {
                    tg.submitFrames(sets[k]);
                    if (tg.pending() == 2) {
                        tg.collectFrames(bench ? nullptr : &frame);
                        if (!bench) all_frames.insert(all_frames.end(), frame.begin(), frame.end());
                    }
                }
                while (tg.pending()) {
                    tg.collectFrames(&frame);
                    if (!bench) all_frames.insert(all_frames.end(), frame.begin(), frame.end());
                }
                secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                frame.erase(frame.begin(), frame.end() - long(W * H * 4));   // the last frame of the last batch
            } else {
                for (int f = 0; f < frames; ++f) {
                    if (independent)
                        for (auto &v : vrs) v->updateView(the_view);   // (resets the running mean: iteration 0)
                    secs += tg.renderFrame(frame);
                    if (independent) all_frames.insert(all_frames.end(), frame.begin(), frame.end());
                }
            }
            if (!all_frames.empty()) {
                std::ofstream af(out + ".frames.rgba.f32", std::ios::binary);
                af.write(reinterpret_cast<const char *>(all_frames.data()), std::streamsize(all_frames.size() * sizeof(float)));
            }
            std::ofstream raw(out + ".rgba.f32", std::ios::binary);
            raw.write(reinterpret_cast<const char *>(frame.data()), std::streamsize(frame.size() * sizeof(float)));
            write_ppm(out + ".ppm", frame, W, H);
            auto res = vrs[0]->getResolution();
            std::printf("{\"device\": \"%s\", \"volume\": [%u, %u, %u], \"width\": %zu, \"height\": %zu, "
                        "\"frames\": %d, \"ranks\": %d, \"tile\": %zu, \"transport\": \"%s\", "
                        "\"frames_per_exchange\": %zu, \"independent\": %s, \"sent_bytes_per_frame\": %.0f, "
                        "\"dense_bytes_per_frame\": %.0f, \"frame_ms\": %.4f, \"ms_per_frame\": %.4f, "
                        "\"clock\": \"host, first submit to last assembled batch\", \"out\": \"%s.rgba.f32\"}\n",
                        vrs[0]->getCurrentDeviceName().c_str(), res[0], res[1], res[2], W, H, frames, ranks, tile,
                        tg.transport().c_str(), K, (batched || independent) ? "true" : "false",
                        batched ? tg.sentBytes() / std::max(1, frames + (bench ? 2 * int(std::min(K, size_t(frames))) : 0)) : 0.0,
                        batched ? tg.denseBytes() / std::max(1, frames + (bench ? 2 * int(std::min(K, size_t(frames))) : 0)) : 0.0,
                        secs / frames * 1e3, secs / frames * 1e3, out.c_str());
            return 0;
        }

        VolumeRenderCL vr;
        if (!setup(vr, device)) return 0;
        std::vector<float> frame, all_frames;
        double kernel_s = 0.0;
        if (frames_per_launch > 0) {
            // ---- the throughput path on one GPU: F renderers over one shared volume take launch sets of <= K
            // independent frames in turn (what bench.py times; volumerenderwidget.cpp:464-486 is the one-frame-per-
            // paintGL caller this replaces for a caller that has many frames to render)
            if (pathtrace || img_ess || use_ao_flag)
                throw std::runtime_error("--frames-per-launch: ray caster only, no image-order ESS, no ambient occlusion");
            const size_t K = size_t(std::min(256, std::max(1, frames_per_launch)));
            // (a short run -- all the frames fit one launch set -- goes to ONE renderer as one set: two renderers with
            // half the frames each pay a set's ramp and tail side by side without hiding each other's; bench.py does
            // the same)
            const size_t F = size_t(frames) <= K ? 1 : size_t(std::max(1, frames_in_flight));
            std::vector<std::unique_ptr<VolumeRenderCL>> twins;
            std::vector<VolumeRenderCL *> lanes{&vr};
            vr.setRoundBudget(unsigned(round_budget));
            for (size_t j = 1; j < F; ++j) {
                twins.push_back(vr.shareVolumes());
                lanes.push_back(twins.back().get());
            }
            auto hip_ok = [](hipError_t e, const char *what) {
                if (e != hipSuccess) throw std::runtime_error(std::string("ERROR: ") + what + " (" + hipGetErrorString(e) + ")");
            };
            hip_ok(hipSetDevice(device), "hipSetDevice");
            std::vector<float *> blocks(F, nullptr);
            std::vector<hipStream_t> streams(F);
            for (size_t j = 0; j < F; ++j) {
                hip_ok(hipMalloc(reinterpret_cast<void **>(&blocks[j]), K * W * H * 4 * sizeof(float)), "hipMalloc frames");
                streams[j] = static_cast<hipStream_t>(lanes[j]->stream());
                lanes[j]->setFrameTiming(false);   // nobody reads a set's own time: the next set may start under its tail
            }
            const std::vector<unsigned int> seeds = vr.drawSeeds(size_t(frames));
            // launch sets of <= K frames, as many as a multiple of the renderers and all of (nearly) one size
            size_t n_sets = (size_t(frames) + K - 1) / K;
            n_sets = std::min(size_t(frames), (n_sets + F - 1) / F * F);
            std::vector<std::vector<unsigned int>> sets;
            for (size_t i = 0; i < n_sets; ++i) {
                const size_t lo = size_t(std::llround(double(i) * frames / double(n_sets)));
                const size_t hi = size_t(std::llround(double(i + 1) * frames / double(n_sets)));
                if (hi > lo) sets.emplace_back(seeds.begin() + long(lo), seeds.begin() + long(hi));
            }
            if (bench) {   // every renderer once, untimed: buffers, work queue, skip bitmap, cell grid, cost map
                std::mt19937 warm(20261004u);
                for (size_t j = 0; j < F; ++j) {
                    std::vector<unsigned int> ws(sets[0].size());
                    for (auto &x : ws) x = static_cast<unsigned int>(warm());
                    lanes[j]->renderFrames(W, H, ws, blocks[j]);
                }
                hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
            }
            hipEvent_t ev0, ev1;
            std::vector<hipEvent_t> done(F);
            hip_ok(hipEventCreate(&ev0), "hipEventCreate");
            hip_ok(hipEventCreate(&ev1), "hipEventCreate");
            for (auto &e : done) hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
            all_frames.reserve(bench ? 0 : size_t(frames) * W * H * 4);
            const auto t0 = std::chrono::steady_clock::now();
            hip_ok(hipEventRecord(ev0, streams[0]), "hipEventRecord");
            for (size_t j = 1; j < F; ++j) hip_ok(hipStreamWaitEvent(streams[j], ev0, 0), "hipStreamWaitEvent");
            for (size_t i = 0; i < sets.size(); ++i) {
                lanes[i % F]->renderFrames(W, H, sets[i], blocks[i % F]);
                if (!bench && ((i + 1) % F == 0 || i + 1 == sets.size())) {
                    // the frames of this round of sets to the host, in frame order, before their blocks are reused
                    hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
                    for (size_t k = i / F * F; k <= i; ++k) {
                        const size_t at = all_frames.size(), nfl = sets[k].size() * W * H * 4;
                        all_frames.resize(at + nfl);
                        hip_ok(hipMemcpy(all_frames.data() + at, blocks[k % F], nfl * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
                    }
                }
            }
            for (size_t j = 1; j < F; ++j) {
                hip_ok(hipEventRecord(done[j], streams[j]), "hipEventRecord");
                hip_ok(hipStreamWaitEvent(streams[0], done[j], 0), "hipStreamWaitEvent");
            }
            hip_ok(hipEventRecord(ev1, streams[0]), "hipEventRecord");
            hip_ok(hipEventSynchronize(ev1), "hipEventSynchronize");
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            float ms = 0.f;
            hip_ok(hipEventElapsedTime(&ms, ev0, ev1), "hipEventElapsedTime");
            // the last frame of the last set
            const size_t last = sets.size() - 1;
            frame.resize(W * H * 4);
            hip_ok(hipMemcpy(frame.data(), blocks[last % F] + (sets[last].size() - 1) * W * H * 4, frame.size() * sizeof(float),
                             hipMemcpyDeviceToHost), "hipMemcpy");
            vrhip_launch_info li;
            std::memset(&li, 0, sizeof li);
            (void)vrhip_last_launch_info(lanes[last % F]->handle(), &li);
            if (!all_frames.empty()) {
                std::ofstream af(out + ".frames.rgba.f32", std::ios::binary);
                af.write(reinterpret_cast<const char *>(all_frames.data()), std::streamsize(all_frames.size() * sizeof(float)));
            }
            std::ofstream raw(out + ".rgba.f32", std::ios::binary);
            raw.write(reinterpret_cast<const char *>(frame.data()), std::streamsize(frame.size() * sizeof(float)));
            write_ppm(out + ".ppm", frame, W, H);
            auto res = vr.getResolution();
            std::printf("{\"device\": \"%s\", \"volume\": [%u, %u, %u], \"width\": %zu, \"height\": %zu, \"frames\": %d, "
                        "\"renderers\": %zu, \"launch_sets\": %zu, \"frames_per_launch_set\": %zu, \"round_budget\": %u, "
                        "\"phase1_waves\": %u, \"phase2_waves\": %u, \"empty_skip\": %u, "
                        "\"ms_per_frame\": %.5f, \"ms_per_frame_host_clock\": %.5f, \"bench\": %s, "
                        "\"clock\": \"HIP events around the region on the first renderer's stream, the other renderers' "
                        "streams joined before the second%s\", \"out\": \"%s.rgba.f32\"}\n",
                        vr.getCurrentDeviceName().c_str(), res[0], res[1], res[2], W, H, frames, F, sets.size(),
                        sets[0].size(), li.round_budget, li.phase1_waves, li.phase2_waves, li.empty_skip,
                        double(ms) / frames, wall / frames * 1e3, bench ? "true" : "false",
                        bench ? "" : " (frames copied to the host inside the region)", out.c_str());
            for (float *b : blocks) (void)hipFree(b);
            (void)hipEventDestroy(ev0);
            (void)hipEventDestroy(ev1);
            for (auto &e : done) (void)hipEventDestroy(e);
            return 0;
        }
        const std::array<float, 16> the_view = have_view ? view : view_matrix(q, tr);
        for (int f = 0; f < frames; ++f) {
            if (independent) vr.updateView(the_view);   // (resets the running mean: iteration 0, volumerendercl.cpp:379-390)
            vr.runRaycastNoGL(W, H, frame);   // frames accumulate (running mean), like the reference
            kernel_s += vr.getLastExecTime();
            if (independent) all_frames.insert(all_frames.end(), frame.begin(), frame.end());
        }
        if (!all_frames.empty()) {
            std::ofstream af(out + ".frames.rgba.f32", std::ios::binary);
            af.write(reinterpret_cast<const char *>(all_frames.data()), std::streamsize(all_frames.size() * sizeof(float)));
        }
        std::ofstream raw(out + ".rgba.f32", std::ios::binary);
        raw.write(reinterpret_cast<const char *>(frame.data()), std::streamsize(frame.size() * sizeof(float)));
        write_ppm(out + ".ppm", frame, W, H);
        auto res = vr.getResolution();
        std::printf("{\"device\": \"%s\", \"volume\": [%u, %u, %u], \"width\": %zu, \"height\": %zu, "
                    "\"frames\": %d, \"kernel_ms_per_frame\": %.4f, \"out\": \"%s.rgba.f32\"}\n",
                    vr.getCurrentDeviceName().c_str(), res[0], res[1], res[2], W, H, frames,
                    kernel_s / frames * 1e3, out.c_str());
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
