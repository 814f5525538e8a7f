// volumerendercl.h -- host-side C++ class with the public surface of the reference's
// `VolumeRenderCL` (/root/reference/src/core/volumerendercl.h:39-482), implemented on the
// C ABI of libvrhip.so (include/vrhip.h).  A caller of the reference class
// (volumerenderwidget.cpp) compiles against this header unchanged for the ray-cast path:
// same method names, argument meaning, call order and exception types.
//
// Differences, all deliberate:
//  * OpenCL types that leaked into the interface have local stand-ins (cl_vendor, cl_GLuint);
//  * runRaycastNoGL fills width*height*4 fp32 RGBA, row 0 = top, as documented there (:173)
//    (the reference reads an UNORM8 image into that vector);
//  * frames accumulate in fp32 and `iteration` advances in both run paths (C9, C10);
//  * ARGB / BGRA volumes (uploaded by the reference, never handled by its kernel) throw
//    std::runtime_error;
//  * `buildScaledVol` (declared but never defined in the reference, :221) is dropped.
#pragma once

#include <array>
#include <memory>
#include <random>
#include <string>
#include <valarray>
#include <vector>

#include "vrhip.h"
#include "datrawreader.h"

typedef unsigned int uint;
typedef unsigned int cl_GLuint;
// openclutilities.h:56-62
enum cl_vendor { VENDOR_ANY, VENDOR_NVIDIA, VENDOR_AMD, VENDOR_INTEL };

class VolumeRenderCL
{
public:
    typedef vrhip_camera_params camera_params;
    typedef vrhip_rendering_params rendering_params;
    typedef vrhip_raycast_params raycast_params;
    typedef vrhip_pathtrace_params pathtrace_params;

    enum kernel_arg {
        VOLUME = 0, BRICKS = 1, TFF = 2, OUTPUT, TFF_PREFIX, IN_ACCUMULATE, OUT_ACCUMULATE,
        IN_HIT_IMG, OUT_HIT_IMG, ENVIRONMENT, CAMERA, RENDERING, RAYCAST, PATHTRACE
    };
    enum scaling_metric { MIN = 0, MAX, AVG, DENSITY };
    enum technique { TECH_RAYCAST = 0, TECH_PATHTRACE = 1 };

    VolumeRenderCL();
    ~VolumeRenderCL();
    VolumeRenderCL(const VolumeRenderCL &) = delete;
    VolumeRenderCL &operator=(const VolumeRenderCL &) = delete;

    // useGL / useCPU / vendor / platformId select OpenCL plumbing that does not exist here;
    // useCPU throws (there is no CPU path); deviceName "N" or platformId >= 0 pick HIP device N.
    void initialize(bool useGL = false, bool useCPU = false, cl_vendor vendor = VENDOR_ANY,
                    const std::string deviceName = "", const int platformId = -1);
    void updateView(const std::array<float, 16> viewMat);
    void updateSamplingRate(const double samplingRate);
    void updateOutputImg(const size_t width, const size_t height, cl_GLuint texId);
    void runRaycast(const size_t width, const size_t height);
    void runRaycastNoGL(const size_t width, const size_t height, std::vector<float> &output);
    size_t loadVolumeData(const DatRawReader::Properties volumeFileProps);
    bool hasData() const;
    const std::array<unsigned int, 4> getResolution() const;
    void setTransferFunction(std::vector<unsigned char> &tff);
    void setTffPrefixSum(std::vector<unsigned int> &tffPrefixSum);
    void scaleVolume(std::valarray<float> scale);

    void setCamOrtho(bool setCamOrtho);
    void setIllumination(unsigned int illum);
    void setShowESS(bool showESS);
    void setLinearInterpolation(bool linearSampling);
    void setContours(bool contours);
    void setAerial(bool aerial);
    void setImgEss(bool useEss);
    void setObjEss(bool useEss);
    void setBackground(std::array<float, 4> color);
    double getLastExecTime();
    const std::vector<std::string> getPlatformNames();
    const std::vector<std::string> getDeviceNames(size_t platformId, const std::string &type);
    const std::string getCurrentDeviceName();
    void setAmbientOcclusion(bool ao);
    const std::string volumeDownsampling(const size_t t, const int factor);
    const std::array<double, 256> &getHistogram(unsigned int timestep = 0);
    void createEnvironmentMap(const std::string &file_name);
    void setUseGradient(bool useGradient);
    void setTechnique(technique tech);
    void setExtinction(const double extinction);
    void setBBox(float bl_x, float bl_y, float bl_z, float tr_x, float tr_y, float tr_z);
    void setTimestep(const size_t t);

    // ---- additions for the headless / multi-GPU host (no reference counterpart)
    // Synthetic input of SURVEY 8(d) generated in HBM: kind "sphere" | "shells".
    void loadSyntheticVolume(const std::string &kind, unsigned int res, DatRawReader::data_format f);
    // Image tiles: compact device buffer [n][tile_h][tile_w][4].
    void renderTiles(size_t width, size_t height, size_t tile_w, size_t tile_h,
                     const std::vector<unsigned int> &tile_ids, float *out_tiles_dev,
                     bool advanceIteration = false);   // true: the frame counts for the running mean (:540)
    void setSeed(unsigned int seed);   // pin the per-frame jitter seed
    void clearSeed();                  // back to the std::mt19937 sequence
    vrhip_renderer *handle() { return _r; }
    const rendering_params &renderingParams() const { return _rendering_params; }

    // ---- the throughput path (no reference counterpart: runRaycast renders one frame per call and waits,
    // volumerendercl.cpp:506-558; a caller that wants independent frames -- a turntable, an image-tile share of a
    // multi-GPU split, a benchmark -- hands over several at once)
    // A second renderer on the same GPU that renders from THIS renderer's voxels, ESS bricks and footprint volume
    // (vrhip_share_volumes) with a stream, frame buffer and scratch of its own: two launch sets can be in flight over
    // one copy of the volume.  It starts with this renderer's transfer function and parameters; this renderer must
    // outlive it.
    std::unique_ptr<VolumeRenderCL> shareVolumes();
    // seeds.size() <= 256 INDEPENDENT frames (frame f jittered by seeds[f], iteration 0) in ONE set of launches
    // (vrhip_render_batch) into DEVICE memory dev_out[f][height][width][4]; returns without waiting.
    void renderFrames(size_t width, size_t height, const std::vector<unsigned int> &seeds, float *dev_out);
    // the same for a tile subset: dev_out[f][n_tiles][tile_h][tile_w][4]; frame_stride: pixels between the frames of
    // dev_out (0 = packed)
    void renderFramesTiles(size_t width, size_t height, size_t tile_w, size_t tile_h,
                           const std::vector<unsigned int> &tile_ids, const std::vector<unsigned int> &seeds,
                           float *dev_out, size_t frame_stride = 0);
    // the jitter seeds the next n runRaycast calls would use (the default-seeded std::mt19937 member, or the pinned seed)
    std::vector<unsigned int> drawSeeds(size_t n);
    // phase-1 sample rounds per ray (vrhip_set_round_budget): 10 for one frame at a time, 48 for launch sets of
    // several frames; no effect on any pixel
    void setRoundBudget(unsigned int rounds);
    // the events around every launch set (what getLastExecTime reads): off for back-to-back launch sets
    void setFrameTiming(bool on);
    void *stream();   // the hipStream_t the renderer launches on

private:
    void generateBricks();
    void calcScaling();
    void pushParams();
    void beginFrame();
    [[noreturn]] void fail(const char *what, int rc);
    void check(const char *what, int rc);

    vrhip_renderer *_r = nullptr;
    bool _volLoaded = false;
    size_t _timestep = 0;
    std::valarray<float> _modelScale;
    std::string _currentDevice;
    std::mt19937 _generator;   // default-seeded, like the reference's member
    bool _seedPinned = false;
    unsigned int _pinnedSeed = 0;
    camera_params _camera_params;
    rendering_params _rendering_params;
    raycast_params _raycast_params;
    pathtrace_params _pathtrace_params;
    DatRawReader _dr;
    bool _synthetic = false;
    int _channels = 1;          // 1 = R, 2 = RG, 4 = RGBA (volDataToCLmem, :697-705)
    std::vector<unsigned char> _tff;          // what setTransferFunction / setTffPrefixSum were last given
    std::vector<unsigned int> _tffPrefixSum;  // (shareVolumes hands them to the twin)
    bool _objEss = true;
    int _device = 0;
    unsigned int _roundBudget = 0;            // 0 = the library's default
    std::array<unsigned int, 4> _synthRes = {{0, 0, 0, 1}};
};
