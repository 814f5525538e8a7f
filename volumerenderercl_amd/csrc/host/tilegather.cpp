// tilegather.cpp -- see tilegather.h.
#include "tilegather.h"

#include <algorithm>
#include <chrono>
#include <utility>
#include <stdexcept>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "volumerendercl.h"

namespace {

void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string("ERROR: ") + what + " (" + hipGetErrorString(e) + ")");
}
void nccl_check(ncclResult_t e, const char *what)
{
    if (e != ncclSuccess) throw std::runtime_error(std::string("ERROR: ") + what + " (" + ncclGetErrorString(e) + ")");
}

} // namespace

TileGather::TileGather(const std::vector<VolumeRenderCL *> &ranks, const std::vector<int> &devices, size_t width,
                       size_t height, size_t tile, bool loopback, bool force_gather)
    : _ranks(ranks), _devices(devices), _W(width), _H(height), _tile(tile), _loopback(loopback),
      _self_exchange(force_gather && !loopback)
{
    const size_t n = _ranks.size();
    if (n == 0 || devices.size() != n) throw std::invalid_argument("TileGather: one device per rank");
    if (tile == 0 || tile % 16) throw std::invalid_argument("Tile size must be a positive multiple of 16.");
    _tiles_x = (_W + tile - 1) / tile;
    _tiles_y = (_H + tile - 1) / tile;
    _tiles.resize(n);
    const std::vector<unsigned int> owner = vr_deal_tiles(_W, _H, tile, n);
    for (size_t t = 0; t < owner.size(); ++t) _tiles[owner[t]].push_back(static_cast<unsigned int>(t));   // ids ascending
    _cap = 0;
    for (const auto &v : _tiles) _cap = std::max(_cap, v.size());
    const size_t slot_floats = tile * tile * 4, block = _cap * slot_floats;

    _streams.resize(n, nullptr);
    _local.resize(n, nullptr);
    for (size_t r = 0; r < n; ++r) {
        void *s = nullptr;
        if (vrhip_get_stream(_ranks[r]->handle(), &s) != VRHIP_OK) throw std::runtime_error("ERROR: vrhip_get_stream");
        _streams[r] = s;
    }
    hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
    hip_check(hipMalloc(reinterpret_cast<void **>(&_staging), n * block * sizeof(float)), "hipMalloc staging");
    hip_check(hipMemset(_staging, 0, n * block * sizeof(float)), "hipMemset staging");
    hip_check(hipMalloc(reinterpret_cast<void **>(&_frame), _W * _H * 4 * sizeof(float)), "hipMalloc frame");
    std::vector<unsigned int> slot(_tiles_x * _tiles_y, 0);
    for (size_t r = 0; r < n; ++r)
        for (size_t k = 0; k < _tiles[r].size(); ++k) slot[_tiles[r][k]] = static_cast<unsigned int>(r * _cap + k);
    hip_check(hipMalloc(reinterpret_cast<void **>(&_slot_of_tile), slot.size() * sizeof(unsigned int)), "hipMalloc slots");
    hip_check(hipMemcpy(_slot_of_tile, slot.data(), slot.size() * sizeof(unsigned int), hipMemcpyHostToDevice),
              "hipMemcpy slots");
    _local[0] = _staging;   // the root renders straight into its block ...
    if (_self_exchange) {   // ... unless it is to send its tiles to itself like a peer (see tilegather.h)
        hip_check(hipMalloc(reinterpret_cast<void **>(&_local[0]), block * sizeof(float)), "hipMalloc tiles");
        hip_check(hipMemset(_local[0], 0, block * sizeof(float)), "hipMemset tiles");
    }
    for (size_t r = 1; r < n; ++r) {
        hip_check(hipSetDevice(_devices[r]), "hipSetDevice");
        hip_check(hipMalloc(reinterpret_cast<void **>(&_local[r]), block * sizeof(float)), "hipMalloc tiles");
        hip_check(hipMemset(_local[r], 0, block * sizeof(float)), "hipMemset tiles");
    }
    if (!_loopback) {
        std::vector<ncclComm_t> comms(n);
        nccl_check(ncclCommInitAll(comms.data(), static_cast<int>(n), _devices.data()), "ncclCommInitAll");
        for (ncclComm_t c : comms) _comms.push_back(static_cast<void *>(c));
    }
}

TileGather::~TileGather()
{
    for (void *c : _comms) (void)ncclCommDestroy(static_cast<ncclComm_t>(c));
    for (size_t r = 1; r < _local.size(); ++r)
        if (_local[r]) {
            (void)hipSetDevice(_devices[r]);
            (void)hipFree(_local[r]);
        }
    (void)hipSetDevice(_devices[0]);
    if (_self_exchange && _local[0]) (void)hipFree(_local[0]);
    if (_staging) (void)hipFree(_staging);
    if (_frame) (void)hipFree(_frame);
    if (_slot_of_tile) (void)hipFree(_slot_of_tile);
}

double TileGather::renderFrame(std::vector<float> &out)
{
    const size_t n = _ranks.size();
    const size_t block = _cap * _tile * _tile * 4;
    const auto t0 = std::chrono::steady_clock::now();
    // every rank: its tiles of this frame into its compact buffer, on its own stream
    for (size_t r = 0; r < n; ++r)
        if (!_tiles[r].empty()) _ranks[r]->renderTiles(_W, _H, _tile, _tile, _tiles[r], _local[r], true);
    if (n > 1 || _self_exchange) {
        if (_loopback) {
            // rehearsal on one device: the peers' blocks reach the root's staging by copies ordered
            // behind the peers' streams
            for (size_t r = 1; r < n; ++r) {
                hip_check(hipMemcpyAsync(_staging + r * block, _local[r], block * sizeof(float), hipMemcpyDeviceToDevice,
                                         static_cast<hipStream_t>(_streams[r])), "hipMemcpyAsync");
                hip_check(hipStreamSynchronize(static_cast<hipStream_t>(_streams[r])), "hipStreamSynchronize");
            }
        } else {
            // one point-to-point exchange per peer, all in one group: the root's receives and the
            // peers' sends progress together, each on the stream its renderer launched on
            nccl_check(ncclGroupStart(), "ncclGroupStart");
            for (size_t r = _self_exchange ? 0 : 1; r < n; ++r) {   // (r = 0: the root's send to itself)
                nccl_check(ncclRecv(_staging + r * block, block, ncclFloat, static_cast<int>(r),
                                    static_cast<ncclComm_t>(_comms[0]), static_cast<hipStream_t>(_streams[0])), "ncclRecv");
                nccl_check(ncclSend(_local[r], block, ncclFloat, 0, static_cast<ncclComm_t>(_comms[r]),
                                    static_cast<hipStream_t>(_streams[r])), "ncclSend");
            }
            nccl_check(ncclGroupEnd(), "ncclGroupEnd");
        }
    }
    // root: de-interleave the tile slots into the frame, behind the receives on its stream
    if (vrhip_assemble_frame(_ranks[0]->handle(), _staging, _slot_of_tile, uint32_t(_W), uint32_t(_H), uint32_t(_tile),
                             uint32_t(_tile), _frame) != VRHIP_OK)
        throw std::runtime_error(std::string("ERROR: vrhip_assemble_frame (") + vrhip_last_error(_ranks[0]->handle()) + ")");
    hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(_streams[0])), "hipStreamSynchronize");
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out.resize(_W * _H * 4);
    hip_check(hipMemcpy(out.data(), _frame, out.size() * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy frame");
    for (size_t r = 1; r < n; ++r) {   // (the peers' buffers are free for the next frame)
        hip_check(hipSetDevice(_devices[r]), "hipSetDevice");
        hip_check(hipStreamSynchronize(static_cast<hipStream_t>(_streams[r])), "hipStreamSynchronize");
    }
    return secs;
}
