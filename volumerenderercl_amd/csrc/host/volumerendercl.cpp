// volumerendercl.cpp -- VolumeRenderCL on libvrhip (see include/volumerendercl.h).  Each
// method cites the reference behaviour it reproduces
// (/root/reference/src/core/volumerendercl.cpp).
#include "hdrloader.h"
#include "volumerendercl.h"

#include <cstring>
#include <iostream>
#include <numeric>
#include <fstream>
#include <stdexcept>

VolumeRenderCL::VolumeRenderCL() : _modelScale{1.0f, 1.0f, 1.0f}
{
    // member defaults of volumerendercl.h:43-81
    std::memset(&_camera_params, 0, sizeof _camera_params);
    std::memset(&_rendering_params, 0, sizeof _rendering_params);
    std::memset(&_raycast_params, 0, sizeof _raycast_params);
    const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(_camera_params.viewMat, ident, sizeof ident);
    for (int i = 0; i < 3; ++i) { _camera_params.bbox_bl[i] = -1.f; _camera_params.bbox_tr[i] = 1.f; }
    for (int i = 0; i < 4; ++i) _rendering_params.backgroundColor[i] = 1.f;
    for (int i = 0; i < 3; ++i) _rendering_params.modelScale[i] = 1.f;
    _rendering_params.illumType = 1;
    _rendering_params.useLinear = 1;
    _rendering_params.seed = 42;
    _raycast_params.samplingRate = 1.5f;
    for (int i = 0; i < 3; ++i) _raycast_params.brickRes[i] = 1.f;
    _pathtrace_params.max_extinction = 100.f;
}

VolumeRenderCL::~VolumeRenderCL()
{
    if (_r) vrhip_destroy(_r);
}

// logCLerror (volumerendercl.cpp:83-89): failures surface as std::runtime_error
// "ERROR: <what> (<detail>)"; bad arguments as std::invalid_argument.
void VolumeRenderCL::fail(const char *what, int rc)
{
    const char *msg = vrhip_last_error(_r);
    std::string text = (msg && *msg) ? std::string(msg) : std::string("ERROR: ") + what;
    std::cerr << "Error in " << what << ": " << text << std::endl;
    if (rc == VRHIP_ERR_INVALID) throw std::invalid_argument(text);
    throw std::runtime_error(text);
}

void VolumeRenderCL::check(const char *what, int rc)
{
    if (rc != VRHIP_OK) fail(what, rc);
}

void VolumeRenderCL::initialize(bool useGL, bool useCPU, cl_vendor, const std::string deviceName,
                                const int platformId)
{
    if (useCPU)
        throw std::runtime_error("ERROR: no CPU device path; this renderer runs on MI355X only");
    if (useGL)
        std::cout << "OpenGL context sharing is not available in the headless build. "
                  << "Using buffer generation instead." << std::endl;
    int device = 0;
    if (!deviceName.empty() && deviceName.find_first_not_of("0123456789") == std::string::npos)
        device = std::stoi(deviceName);
    else if (platformId >= 0)
        device = platformId;
    if (_r) { vrhip_destroy(_r); _r = nullptr; }
    _device = device;
    int rc = vrhip_create(device, &_r);
    if (rc != VRHIP_OK) {
        const char *msg = vrhip_last_error(nullptr);
        throw std::runtime_error(msg && *msg ? msg : "ERROR: vrhip_create");
    }
    char name[256];
    vrhip_device_name(_r, name, sizeof name);
    _currentDevice = name;
    // upload volume data to the device if already loaded (:153-158)
    if (_dr.has_data()) {
        const auto &p = _dr.properties();
        for (size_t t = 0; t < _dr.data().size(); ++t)
            check("volDataToCLmem",
                  vrhip_upload_volume_channels(_r, _dr.data()[t].data(), p.volume_res.data(),
                                               int(p.format), _channels, uint32_t(t)));
    }
}

void VolumeRenderCL::updateView(const std::array<float, 16> viewMat)   // :379-390
{
    if (!_volLoaded || _modelScale.size() < 3) return;
    for (size_t i = 0; i < 16; ++i) _camera_params.viewMat[i] = viewMat[i];
    _rendering_params.iteration = 0;
}

void VolumeRenderCL::updateSamplingRate(const double samplingRate)   // :397-401
{
    _raycast_params.samplingRate = static_cast<float>(samplingRate);
}

void VolumeRenderCL::updateOutputImg(const size_t, const size_t, cl_GLuint)
{
    // output / accumulate buffers are sized by the render call itself (:465-499); the hit images of
    // image-order ESS start over with their initial contents (:482-488)
    if (_r) check("updateOutputImg", vrhip_reset_image_ess(_r));
}

void VolumeRenderCL::pushParams()
{
    for (int i = 0; i < 3; ++i) _rendering_params.modelScale[i] = _modelScale[i];   // :206-207
    check("setCameraArgs", vrhip_set_camera_params(_r, &_camera_params));
    check("setRenderingArgs", vrhip_set_rendering_params(_r, &_rendering_params));
    check("setRaycastArgs", vrhip_set_raycast_params(_r, &_raycast_params));
    check("setPathtraceArgs", vrhip_set_pathtrace_params(_r, &_pathtrace_params));
}

void VolumeRenderCL::beginFrame()   // setMemObjectsRaycast (:194-218): fresh seed per frame
{
    _rendering_params.seed = _seedPinned ? _pinnedSeed : static_cast<unsigned int>(_generator());
    pushParams();
}

void VolumeRenderCL::runRaycast(const size_t width, const size_t height)   // :506-558
{
    if (!_volLoaded) return;
    beginFrame();
    check("runRaycast", vrhip_render_frame(_r, uint32_t(width), uint32_t(height), nullptr, 0));
    _rendering_params.iteration++;   // :540
}

void VolumeRenderCL::runRaycastNoGL(const size_t width, const size_t height,
                                    std::vector<float> &output)   // :568-607
{
    if (!_volLoaded) return;
    beginFrame();
    output.resize(width * height * 4);   // :583
    check("runRaycastNoGL", vrhip_render_frame(_r, uint32_t(width), uint32_t(height),
                                               output.data(), 0));
    _rendering_params.iteration++;
}

void VolumeRenderCL::renderTiles(size_t width, size_t height, size_t tile_w, size_t tile_h,
                                 const std::vector<unsigned int> &tile_ids, float *out_tiles_dev,
                                 bool advanceIteration)
{
    if (!_volLoaded) return;
    beginFrame();
    check("renderTiles", vrhip_render_tiles(_r, uint32_t(width), uint32_t(height), uint32_t(tile_w),
                                            uint32_t(tile_h), tile_ids.data(),
                                            uint32_t(tile_ids.size()), out_tiles_dev));
    if (advanceIteration) _rendering_params.iteration++;
}

size_t VolumeRenderCL::loadVolumeData(const DatRawReader::Properties volumeFileProps)   // :765-805
{
    _volLoaded = false;
    _synthetic = false;
    if (volumeFileProps.dat_file_name.empty() && !volumeFileProps.raw_file_names.empty())
        std::cerr << "Loading raw volume data from " << volumeFileProps.raw_file_names.at(0)
                  << std::endl;
    else
        std::cout << "Loading volume data defined in " << volumeFileProps.dat_file_name << std::endl;
    try {
        _dr.read_files(volumeFileProps);
        const auto &p = _dr.properties();
        std::cout << _dr.data().front().size() * _dr.data().size() << " bytes have been read from "
                  << _dr.data().size() << " file(s)." << std::endl;
        std::cout << p.to_string() << std::endl;
        // volDataToCLmem (:690-759)
        const std::string &co = p.image_channel_order;
        int channels = 1;
        if (co == "RG") channels = 2;
        else if (co == "RGBA") channels = 4;
        else if (co == "ARGB" || co == "BGRA")
            // accepted by the reference's upload (:706-709), but its kernel has no branch for these
            // orders and composites an unset colour (volumeraycast.cl:800-855)
            throw std::invalid_argument("ARGB / BGRA volumes are not supported.");
        else if (!(co == "R" || co == "" || co == "I" || co == "LUMINANCE"))
            throw std::invalid_argument("Unknown or invalid volume color format.");
        _channels = channels;
        if (p.format != DatRawReader::UCHAR && p.format != DatRawReader::USHORT &&
            p.format != DatRawReader::FLOAT)
            throw std::invalid_argument("Unknown or invalid volume data format.");
        const size_t bpv = p.format == DatRawReader::UCHAR ? 1 : p.format == DatRawReader::USHORT ? 2 : 4;
        check("clearVolumes", vrhip_clear_volumes(_r));
        for (size_t t = 0; t < _dr.data().size(); ++t) {
            // (the reference checks one channel's worth, :740-742, and lets the image read beyond)
            if (size_t(p.volume_res[0]) * p.volume_res[1] * p.volume_res[2] * bpv * size_t(channels) >
                _dr.data()[t].size()) {
                _dr.clearData();
                throw std::runtime_error("Volume size does not match size specified in dat file.");
            }
            check("volDataToCLmem",
                  vrhip_upload_volume_channels(_r, _dr.data()[t].data(), p.volume_res.data(),
                                               int(p.format), _channels, uint32_t(t)));
        }
        calcScaling();
    } catch (std::invalid_argument &e) {
        throw std::runtime_error(e.what());   // :784-787
    }
    // default prefix sum of the linear ramp (:795-801)
    std::vector<unsigned int> prefixSum(1024, 0);
    for (size_t i = 0; i < prefixSum.size(); ++i) prefixSum[i] = static_cast<unsigned int>(i) * 4u;
    std::partial_sum(prefixSum.begin(), prefixSum.end(), prefixSum.begin());
    _volLoaded = true;
    setTffPrefixSum(prefixSum);
    return _dr.data().size();
}

void VolumeRenderCL::loadSyntheticVolume(const std::string &kind, unsigned int res,
                                         DatRawReader::data_format f)
{
    _volLoaded = false;
    const uint32_t r3[3] = {res, res, res};
    check("clearVolumes", vrhip_clear_volumes(_r));
    check("synthVolume", vrhip_synth_volume(_r, kind == "shells" ? 1 : 0, r3, int(f), 0));
    _synthetic = true;
    _synthRes = {{res, res, res, 1}};
    _modelScale = {1.f, 1.f, 1.f};
    std::vector<unsigned int> prefixSum(1024, 0);
    for (size_t i = 0; i < prefixSum.size(); ++i) prefixSum[i] = static_cast<unsigned int>(i) * 4u;
    std::partial_sum(prefixSum.begin(), prefixSum.end(), prefixSum.begin());
    _volLoaded = true;
    setTffPrefixSum(prefixSum);
}

void VolumeRenderCL::calcScaling()   // :347-362
{
    if (!_dr.has_data()) return;
    const auto &p = _dr.properties();
    _modelScale = {static_cast<float>(p.volume_res.at(0)), static_cast<float>(p.volume_res.at(1)),
                   static_cast<float>(p.volume_res.at(2))};
    std::valarray<float> thickness = {static_cast<float>(p.slice_thickness.at(0)),
                                      static_cast<float>(p.slice_thickness.at(1)),
                                      static_cast<float>(p.slice_thickness.at(2))};
    _modelScale *= thickness * (1.f / thickness[0]);
    _modelScale = _modelScale.max() / _modelScale;
}

void VolumeRenderCL::scaleVolume(std::valarray<float> scale) { _modelScale *= scale; }

bool VolumeRenderCL::hasData() const { return _volLoaded; }

const std::array<unsigned int, 4> VolumeRenderCL::getResolution() const   // :822-827
{
    if (_synthetic) return _synthRes;
    if (!_dr.has_data()) return std::array<unsigned int, 4>{{0, 0, 0, 1}};
    return _dr.properties().volume_res;
}

const std::array<double, 256> &VolumeRenderCL::getHistogram(unsigned int timestep)   // :853-858
{
    if (!_dr.has_data()) throw std::invalid_argument("Invalid timestep for histogram data.");
    return _dr.getHistogram(timestep);
}

void VolumeRenderCL::setTransferFunction(std::vector<unsigned char> &tff)   // :864-891
{
    if (!_volLoaded) return;
    _tff = tff;
    check("setTransferFunction",
          vrhip_set_transfer_function(_r, tff.data(), uint32_t(tff.size() / 4)));
    generateBricks();
    std::vector<unsigned int> prefixSum;
    for (size_t i = 3; i < tff.size(); i += 4) prefixSum.push_back(static_cast<unsigned int>(tff.at(i)));
    std::partial_sum(prefixSum.begin(), prefixSum.end(), prefixSum.begin());
    setTffPrefixSum(prefixSum);
    _rendering_params.iteration = 0;
}

void VolumeRenderCL::setTffPrefixSum(std::vector<unsigned int> &tffPrefixSum)   // :898-916
{
    if (!_volLoaded) return;
    _tffPrefixSum = tffPrefixSum;
    check("setTffPrefixSum",
          vrhip_set_tff_prefix_sum(_r, tffPrefixSum.data(), uint32_t(tffPrefixSum.size())));
}

void VolumeRenderCL::generateBricks()   // :614-684
{
    check("generateBricks", vrhip_build_bricks(_r));
    float brf[3];
    check("generateBricks", vrhip_get_brick_info(_r, nullptr, brf, nullptr));
    for (int i = 0; i < 3; ++i) _raycast_params.brickRes[i] = brf[i];   // :627-631
}

void VolumeRenderCL::setCamOrtho(bool v) { _camera_params.ortho = v ? 1u : 0u; }
void VolumeRenderCL::setIllumination(unsigned int illum) { _rendering_params.illumType = illum; }
void VolumeRenderCL::setAmbientOcclusion(bool ao) { _raycast_params.useAO = ao ? 1u : 0u; }
void VolumeRenderCL::setShowESS(bool v) { _rendering_params.showEss = v ? 1u : 0u; }
void VolumeRenderCL::setLinearInterpolation(bool v) { _rendering_params.useLinear = v ? 1u : 0u; }
void VolumeRenderCL::setContours(bool v) { _raycast_params.contours = v ? 1u : 0u; }
void VolumeRenderCL::setAerial(bool v) { _raycast_params.aerial = v ? 1u : 0u; }
void VolumeRenderCL::setImgEss(bool v) { _rendering_params.imgEss = v ? 1u : 0u; }
void VolumeRenderCL::setUseGradient(bool v) { _rendering_params.useGradient = v ? 1u : 0u; }

void VolumeRenderCL::setObjEss(bool useEss)   // :1006-1019: kernel-variant switch, no rebuild
{
    _objEss = useEss;
    check("setObjEss", vrhip_set_object_ess(_r, useEss ? 1 : 0));
}

void VolumeRenderCL::setBackground(std::array<float, 4> color)   // :1025-1030
{
    // the reference converts through cl_float3, which zeroes the 4th component
    _rendering_params.backgroundColor[0] = color[0];
    _rendering_params.backgroundColor[1] = color[1];
    _rendering_params.backgroundColor[2] = color[2];
    _rendering_params.backgroundColor[3] = 0.f;
}

void VolumeRenderCL::setTechnique(technique tech)   // :1042-1047
{
    _rendering_params.technique = static_cast<uint>(tech);
    _rendering_params.iteration = 0;
}

void VolumeRenderCL::setExtinction(const double extinction)
{
    _pathtrace_params.max_extinction = float(extinction);
}

void VolumeRenderCL::setBBox(float bl_x, float bl_y, float bl_z, float tr_x, float tr_y, float tr_z)
{
    _camera_params.bbox_bl[0] = bl_x; _camera_params.bbox_bl[1] = bl_y; _camera_params.bbox_bl[2] = bl_z;
    _camera_params.bbox_tr[0] = tr_x; _camera_params.bbox_tr[1] = tr_y; _camera_params.bbox_tr[2] = tr_z;
    _rendering_params.iteration = 0;
}

void VolumeRenderCL::setTimestep(const size_t t)   // :1167-1174
{
    if (_dr.has_data() && t >= _dr.properties().volume_res.at(3)) return;
    _timestep = t;
    check("setTimestep", vrhip_set_timestep(_r, uint32_t(t)));
    _rendering_params.iteration = 0;
}

double VolumeRenderCL::getLastExecTime() { return vrhip_last_kernel_seconds(_r); }

const std::vector<std::string> VolumeRenderCL::getPlatformNames() { return {"AMD HIP (ROCm)"}; }

const std::vector<std::string> VolumeRenderCL::getDeviceNames(size_t, const std::string &type)
{
    if (type == "CPU") return {};
    return {_currentDevice};
}

const std::string VolumeRenderCL::getCurrentDeviceName() { return _currentDevice; }

// volumerendercl.cpp:238-341: down-sample time step t on the GPU, write <dat name>_<N>.raw/.dat
// next to the loaded .dat, return the path without extension.  The .dat text is the reference's,
// including its quirks: `Format:` carries the enum's integer and `ObjectFileName:` is cut with
// substr(first + 1, lastindex) where lastindex is a position, not a length.
const std::string VolumeRenderCL::volumeDownsampling(const size_t t, const int factor)
{
    if (!_dr.has_data()) throw std::runtime_error("No volume data is loaded.");
    if (factor < 2) throw std::invalid_argument("Factor must be greater or equal 2.");
    uint32_t lo[3] = {0, 0, 0};
    int rc = vrhip_downsample_volume(_r, uint32_t(t), factor, nullptr, 0, lo);
    if (rc == VRHIP_ERR_INVALID) throw std::invalid_argument(vrhip_last_error(_r));
    check("vrhip_downsample_volume", rc);
    const DatRawReader::Properties &p = _dr.properties();
    const size_t mult = p.format == DatRawReader::UCHAR ? 1 : p.format == DatRawReader::USHORT ? 2 : 4;
    std::vector<unsigned char> outputData(size_t(lo[0]) * lo[1] * lo[2] * mult);
    check("vrhip_downsample_volume", vrhip_downsample_volume(_r, uint32_t(t), factor, outputData.data(),
                                                             outputData.size(), lo));
    size_t lastindex = p.dat_file_name.find_last_of(".");
    std::string rawname = p.dat_file_name.substr(0, lastindex);
    rawname += "_";
    rawname += std::to_string(lo[0]);
    std::ofstream file(rawname + ".raw", std::ios::out | std::ios::binary);
    file.write(reinterpret_cast<const char *>(outputData.data()), std::streamsize(outputData.size()));
    file.close();
    std::ofstream datFile(rawname + ".dat", std::ios::out);
    lastindex = rawname.find_last_of(".");
    size_t firstindex = rawname.find_last_of("/\\");
    std::string rawnameShort = rawname.substr(firstindex + 1, lastindex);
    datFile << "ObjectFileName: \t" << rawnameShort << ".raw\n";
    datFile << "Resolution: \t\t" << lo[0] << " " << lo[1] << " " << lo[2] << "\n";
    datFile << "SliceThickness: \t" << p.slice_thickness.at(0) << " " << p.slice_thickness.at(1) << " "
            << p.slice_thickness.at(2) << "\n";
    datFile << "Format: \t\t\t" << p.format << "\n";
    datFile.close();
    return rawname;
}

void VolumeRenderCL::createEnvironmentMap(const std::string &file_name)
{
    // an empty name installs the 1x1 white map, which the kernel never samples (:1130-1134, :655)
    if (file_name.empty()) {
        check("createEnvironmentMap", vrhip_set_environment_map(_r, nullptr, 0, 0));
        return;
    }
    vrhost::HdrImage img;
    if (!vrhost::load_hdr_float4(file_name, img))
        throw std::runtime_error("Error loading environment map file.");   // :1138
    check("createEnvironmentMap",
          vrhip_set_environment_map(_r, img.rgba.data(), img.width, img.height));
    std::cout << "Loaded environment map " << file_name << std::endl;
}

void VolumeRenderCL::setSeed(unsigned int seed) { _seedPinned = true; _pinnedSeed = seed; }
void VolumeRenderCL::clearSeed() { _seedPinned = false; }

// ---- the throughput path (include/volumerendercl.h "additions"): launch sets of independent frames on
// vrhip_share_volumes / vrhip_render_batch

std::unique_ptr<VolumeRenderCL> VolumeRenderCL::shareVolumes()
{
    if (!_volLoaded) throw std::runtime_error("No volume data is loaded.");
    std::unique_ptr<VolumeRenderCL> twin(new VolumeRenderCL());
    int rc = vrhip_create(_device, &twin->_r);
    if (rc != VRHIP_OK) {
        const char *msg = vrhip_last_error(nullptr);
        throw std::runtime_error(msg && *msg ? msg : "ERROR: vrhip_create");
    }
    twin->_device = _device;
    twin->_currentDevice = _currentDevice;
    twin->check("shareVolumes", vrhip_share_volumes(twin->_r, _r));
    twin->_synthetic = true;          // (resolution from _synthRes: the twin holds no DatRawReader data)
    twin->_synthRes = getResolution();
    twin->_channels = _channels;
    twin->_modelScale = _modelScale;
    twin->_timestep = _timestep;
    twin->_camera_params = _camera_params;
    twin->_rendering_params = _rendering_params;
    twin->_raycast_params = _raycast_params;
    twin->_pathtrace_params = _pathtrace_params;
    twin->_seedPinned = _seedPinned;
    twin->_pinnedSeed = _pinnedSeed;
    twin->_volLoaded = true;
    twin->check("setTimestep", vrhip_set_timestep(twin->_r, uint32_t(_timestep)));
    twin->setObjEss(_objEss);
    if (!_tff.empty()) {
        std::vector<unsigned char> tff = _tff;
        twin->setTransferFunction(tff);     // (takes the owner's bricks: vrhip_build_bricks on a sharer)
    }
    if (!_tffPrefixSum.empty()) {
        std::vector<unsigned int> prefix = _tffPrefixSum;
        twin->setTffPrefixSum(prefix);
    }
    twin->_rendering_params.iteration = _rendering_params.iteration;
    if (_roundBudget) twin->setRoundBudget(_roundBudget);
    return twin;
}

void VolumeRenderCL::renderFrames(size_t width, size_t height, const std::vector<unsigned int> &seeds, float *dev_out)
{
    renderFramesTiles(width, height, 0, 0, std::vector<unsigned int>(), seeds, dev_out, 0);
}

void VolumeRenderCL::renderFramesTiles(size_t width, size_t height, size_t tile_w, size_t tile_h,
                                       const std::vector<unsigned int> &tile_ids,
                                       const std::vector<unsigned int> &seeds, float *dev_out, size_t frame_stride)
{
    if (!_volLoaded) return;
    _rendering_params.iteration = 0;   // the frames of a launch set are independent
    pushParams();
    check("renderFrames", vrhip_render_batch(_r, uint32_t(width), uint32_t(height), uint32_t(tile_w), uint32_t(tile_h),
                                             tile_ids.empty() ? nullptr : tile_ids.data(), uint32_t(tile_ids.size()),
                                             seeds.data(), uint32_t(seeds.size()), dev_out, uint32_t(frame_stride)));
}

std::vector<unsigned int> VolumeRenderCL::drawSeeds(size_t n)
{
    std::vector<unsigned int> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = _seedPinned ? _pinnedSeed : static_cast<unsigned int>(_generator());
    return out;
}

void VolumeRenderCL::setRoundBudget(unsigned int rounds)
{
    _roundBudget = rounds;
    check("setRoundBudget", vrhip_set_round_budget(_r, rounds));
}

void VolumeRenderCL::setFrameTiming(bool on) { check("setFrameTiming", vrhip_set_frame_timing(_r, on ? 1 : 0)); }

void *VolumeRenderCL::stream()
{
    void *s = nullptr;
    check("stream", vrhip_get_stream(_r, &s));
    return s;
}
