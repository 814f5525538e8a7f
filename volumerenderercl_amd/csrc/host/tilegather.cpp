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
                       size_t height, size_t tile, bool loopback, bool force_gather, double root_share,
                       size_t batch_frames)
    : _ranks(ranks), _devices(devices), _W(width), _H(height), _tile(tile), _loopback(loopback),
      _self_exchange(force_gather && !loopback), _B(batch_frames)
{
    const size_t n = _ranks.size();
    if (n == 0 || devices.size() != n) throw std::invalid_argument("TileGather: one device per rank");
    if (tile == 0 || tile % 16) throw std::invalid_argument("Tile size must be a positive multiple of 16.");
    _tiles_x = (_W + tile - 1) / tile;
    _tiles_y = (_H + tile - 1) / tile;
    _tiles.resize(n);
    const std::vector<unsigned int> owner = vr_deal_tiles(_W, _H, tile, n, root_share);
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
    if (_B == 0) return;
    // ---- the throughput path: two sets of batch buffers per rank, a stream per rank for its side of an exchange
    if (_B > 256) throw std::invalid_argument("TileGather: at most 256 frames per batch");
    if (n > 64 || _cap > 65536) throw std::invalid_argument("TileGather: at most 64 ranks of 65536 tiles each");
    const size_t S = _B * _cap, P = tile * tile, spad = (S + 3) / 4 * 4;
    const size_t msg_floats = spad + 4 * S + 4 * S * P;
    _comm_streams.resize(n, nullptr);
    for (int b = 0; b < 2; ++b) {
        _btiles[b].resize(n, nullptr); _bmsg[b].resize(n, nullptr); _brecv[b].resize(n, nullptr);
        _bscratch[b].resize(n, nullptr); _bcount[b].resize(n, nullptr);
        _ev_packed[b].resize(n, nullptr); _ev_sent[b].resize(n, nullptr);
    }
    for (size_t r = 0; r < n; ++r) {
        hip_check(hipSetDevice(_devices[r]), "hipSetDevice");
        hipStream_t cs = nullptr;
        hip_check(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking), "hipStreamCreate");
        _comm_streams[r] = cs;
        for (int b = 0; b < 2; ++b) {
            hip_check(hipMalloc(reinterpret_cast<void **>(&_btiles[b][r]), S * P * 4 * sizeof(float)), "hipMalloc batch tiles");
            hip_check(hipMemset(_btiles[b][r], 0, S * P * 4 * sizeof(float)), "hipMemset batch tiles");   // (slots beyond a rank's tiles stay one colour)
            hip_check(hipMalloc(reinterpret_cast<void **>(&_bmsg[b][r]), msg_floats * sizeof(float)), "hipMalloc message");
            hip_check(hipMalloc(reinterpret_cast<void **>(&_bscratch[b][r]), S * sizeof(int32_t)), "hipMalloc scratch");
            hip_check(hipMalloc(reinterpret_cast<void **>(&_bcount[b][r]), sizeof(uint32_t)), "hipMalloc count");
            hipEvent_t e0 = nullptr, e1 = nullptr;
            hip_check(hipEventCreateWithFlags(&e0, hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&e1, hipEventDisableTiming), "hipEventCreate");
            _ev_packed[b][r] = e0;
            _ev_sent[b][r] = e1;
        }
    }
    hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
    std::vector<uint32_t> rs(_tiles_x * _tiles_y, 0);
    for (size_t r = 0; r < n; ++r)
        for (size_t k = 0; k < _tiles[r].size(); ++k) rs[_tiles[r][k]] = static_cast<uint32_t>((r << 16) | k);
    hip_check(hipMalloc(reinterpret_cast<void **>(&_rank_slot), rs.size() * sizeof(uint32_t)), "hipMalloc rank slots");
    hip_check(hipMemcpy(_rank_slot, rs.data(), rs.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy rank slots");
    for (int b = 0; b < 2; ++b) {
        for (size_t r = 0; r < n; ++r)
            if (r > 0 || _self_exchange)   // (the root reads its own message where it was packed)
                hip_check(hipMalloc(reinterpret_cast<void **>(&_brecv[b][r]), msg_floats * sizeof(float)), "hipMalloc receive");
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&_hcount[b]), n * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&_bpos[b]), n * S * sizeof(int32_t)), "hipMalloc positions");
        hip_check(hipMalloc(reinterpret_cast<void **>(&_bframes[b]), _B * _W * _H * 4 * sizeof(float)), "hipMalloc frames");
        hipEvent_t e = nullptr;
        hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
        _ev_assembled[b] = e;
    }
}

TileGather::~TileGather()
{
    for (size_t r = 0; r < _comm_streams.size(); ++r) {
        (void)hipSetDevice(_devices[r]);
        if (_comm_streams[r]) (void)hipStreamSynchronize(static_cast<hipStream_t>(_comm_streams[r]));
        (void)hipStreamSynchronize(static_cast<hipStream_t>(_streams[r]));
        for (int b = 0; b < 2; ++b) {
            if (_btiles[b][r]) (void)hipFree(_btiles[b][r]);
            if (_bmsg[b][r]) (void)hipFree(_bmsg[b][r]);
            if (_bscratch[b][r]) (void)hipFree(_bscratch[b][r]);
            if (_bcount[b][r]) (void)hipFree(_bcount[b][r]);
            if (_ev_packed[b][r]) (void)hipEventDestroy(static_cast<hipEvent_t>(_ev_packed[b][r]));
            if (_ev_sent[b][r]) (void)hipEventDestroy(static_cast<hipEvent_t>(_ev_sent[b][r]));
        }
        if (_comm_streams[r]) (void)hipStreamDestroy(static_cast<hipStream_t>(_comm_streams[r]));
    }
    if (_B) {
        (void)hipSetDevice(_devices[0]);
        for (int b = 0; b < 2; ++b) {
            for (float *p : _brecv[b])
                if (p) (void)hipFree(p);
            if (_hcount[b]) (void)hipHostFree(_hcount[b]);
            if (_bpos[b]) (void)hipFree(_bpos[b]);
            if (_bframes[b]) (void)hipFree(_bframes[b]);
            if (_ev_assembled[b]) (void)hipEventDestroy(static_cast<hipEvent_t>(_ev_assembled[b]));
        }
        if (_rank_slot) (void)hipFree(_rank_slot);
    }
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

// ---- batches of independent frames, one exchange in flight (see tilegather.h)

void TileGather::submitFrames(const std::vector<unsigned int> &seeds)
{
    const size_t n = _ranks.size(), nf = seeds.size();
    if (_B == 0) throw std::runtime_error("TileGather: constructed without batch buffers");
    if (nf == 0 || nf > _B) throw std::invalid_argument("TileGather::submitFrames: 1 .. batch_frames frames per batch");
    if (_pending.size() >= 2) throw std::runtime_error("TileGather: two batches already pending: collect first");
    const int b = _next_buf;
    _next_buf ^= 1;
    const size_t P = _tile * _tile, S = nf * _cap;
    for (size_t r = 0; r < n; ++r) {
        hip_check(hipSetDevice(_devices[r]), "hipSetDevice");
        hipStream_t rs = static_cast<hipStream_t>(_streams[r]);
        // buffer set b is free again once the exchange that last read it has left this rank (and, on the root, once
        // the batch that last used it has been assembled: the root's message is read in place)
        if (_sent_valid[b]) {
            hip_check(hipStreamWaitEvent(rs, static_cast<hipEvent_t>(_ev_sent[b][r]), 0), "hipStreamWaitEvent");
            if (r == 0) hip_check(hipStreamWaitEvent(rs, static_cast<hipEvent_t>(_ev_assembled[b]), 0), "hipStreamWaitEvent");
        }
        if (!_tiles[r].empty())
            _ranks[r]->renderFramesTiles(_W, _H, _tile, _tile, _tiles[r], seeds, _btiles[b][r], _cap * P);
        if (vrhip_pack_tiles(_ranks[r]->handle(), rs, _btiles[b][r], uint32_t(S), uint32_t(P), _bscratch[b][r], _bmsg[b][r],
                             _bcount[b][r]) != VRHIP_OK)
            throw std::runtime_error(std::string("ERROR: vrhip_pack_tiles (") + vrhip_last_error(_ranks[r]->handle()) + ")");
        hip_check(hipMemcpyAsync(&_hcount[b][r], _bcount[b][r], sizeof(uint32_t), hipMemcpyDeviceToHost, rs), "hipMemcpyAsync count");
        hip_check(hipEventRecord(static_cast<hipEvent_t>(_ev_packed[b][r]), rs), "hipEventRecord");
    }
    // the batch before this one: its counts are on the host (or nearly), and the GPUs have this batch to render
    // while its messages travel
    for (Pending &p : _pending)
        if (!p.exchanged) startExchange(p);
    _pending.push_back(Pending{b, nf, false});
}

void TileGather::startExchange(Pending &p)
{
    const size_t n = _ranks.size(), P = _tile * _tile, S = p.n * _cap, spad = (S + 3) / 4 * 4;
    const int b = p.b;
    std::vector<size_t> floats(n);
    for (size_t r = 0; r < n; ++r) {
        hip_check(hipEventSynchronize(static_cast<hipEvent_t>(_ev_packed[b][r])), "hipEventSynchronize");   // (the count)
        floats[r] = spad + 4 * S + 4 * size_t(_hcount[b][r]) * P;
    }
    const size_t first_sender = _self_exchange ? 0 : 1;
    if (_loopback) {
        // every rank on one device: a peer's message reaches the root by a copy on the PEER's exchange stream (its
        // "sent" event must cover the read of its message), and the root's exchange stream waits for it below
        hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
        for (size_t r = 1; r < n; ++r)
            hip_check(hipMemcpyAsync(_brecv[b][r], _bmsg[b][r], floats[r] * sizeof(float), hipMemcpyDeviceToDevice,
                                     static_cast<hipStream_t>(_comm_streams[r])), "hipMemcpyAsync message");
    } else if (n > 1 || _self_exchange) {
        nccl_check(ncclGroupStart(), "ncclGroupStart");
        for (size_t r = first_sender; r < n; ++r) {
            nccl_check(ncclRecv(_brecv[b][r], floats[r], ncclFloat, static_cast<int>(r), static_cast<ncclComm_t>(_comms[0]),
                                static_cast<hipStream_t>(_comm_streams[0])), "ncclRecv");
            nccl_check(ncclSend(_bmsg[b][r], floats[r], ncclFloat, 0, static_cast<ncclComm_t>(_comms[r]),
                                static_cast<hipStream_t>(_comm_streams[r])), "ncclSend");
        }
        nccl_check(ncclGroupEnd(), "ncclGroupEnd");
    }
    for (size_t r = 0; r < n; ++r) {
        hip_check(hipSetDevice(_devices[r]), "hipSetDevice");
        hip_check(hipEventRecord(static_cast<hipEvent_t>(_ev_sent[b][r]), static_cast<hipStream_t>(_comm_streams[r])), "hipEventRecord");
        if (_loopback && r > 0)
            hip_check(hipStreamWaitEvent(static_cast<hipStream_t>(_comm_streams[0]), static_cast<hipEvent_t>(_ev_sent[b][r]), 0),
                      "hipStreamWaitEvent");
        if (r >= first_sender) {
            _sent_bytes += double(floats[r]) * sizeof(float);
            _dense_bytes += double(p.n) * double(_tiles[r].size()) * double(P) * 16.0;
        }
    }
    // root: the frames straight from the messages, behind the receives on its exchange stream
    hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
    std::vector<const float *> msgs(n);
    std::vector<uint32_t> counts(n);
    for (size_t r = 0; r < n; ++r) {
        msgs[r] = (r == 0 && !_self_exchange) ? _bmsg[b][0] : _brecv[b][r];
        counts[r] = _hcount[b][r];
    }
    vrhip_renderer *root = _ranks[0]->handle();
    if (vrhip_message_positions(root, _comm_streams[0], msgs.data(), counts.data(), uint32_t(n), uint32_t(S), _bpos[b]) != VRHIP_OK ||
        vrhip_assemble_batch(root, _comm_streams[0], msgs.data(), uint32_t(n), uint32_t(p.n), uint32_t(_cap), uint32_t(spad),
                             _bpos[b], _rank_slot, uint32_t(_W), uint32_t(_H), uint32_t(_tile), uint32_t(_tile),
                             _bframes[b]) != VRHIP_OK)
        throw std::runtime_error(std::string("ERROR: batch assembly (") + vrhip_last_error(root) + ")");
    hip_check(hipEventRecord(static_cast<hipEvent_t>(_ev_assembled[b]), static_cast<hipStream_t>(_comm_streams[0])), "hipEventRecord");
    _sent_valid[b] = true;
    p.exchanged = true;
}

const float *TileGather::collectFrames(std::vector<float> *host_out, size_t *n_frames)
{
    if (_pending.empty()) throw std::runtime_error("TileGather: nothing to collect");
    Pending &p = _pending.front();
    if (!p.exchanged) startExchange(p);
    const int b = p.b;
    const size_t nf = p.n;
    _pending.pop_front();
    hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
    hip_check(hipEventSynchronize(static_cast<hipEvent_t>(_ev_assembled[b])), "hipEventSynchronize");
    if (host_out) {
        host_out->resize(nf * _W * _H * 4);
        hip_check(hipMemcpy(host_out->data(), _bframes[b], host_out->size() * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy frames");
    }
    if (n_frames) *n_frames = nf;
    return _bframes[b];
}
