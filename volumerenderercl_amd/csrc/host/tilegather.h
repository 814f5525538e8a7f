// tilegather.h -- image-tile decomposition of a frame over the GPUs of a node for the headless
// C++ host (the reference is single-GPU, volumerendercl.cpp:140).  One process, one
// renderer per GPU (volume replicated), tiles dealt to the ranks by their distance from the frame's centre
// (every rank gets tiles of every distance: load balance); every rank renders its tiles into a compact device
// buffer and rank 0 receives them over RCCL point to point (ncclGroupStart; ncclRecv from every peer on the
// root | ncclSend on the peers; ncclGroupEnd -- xGMI is point to point, so the root's links carry one peer
// each) and assembles the frames with one kernel.
//
// Two ways to drive it:
//  * renderFrame(): one frame per exchange, dense, synchronous, the renderers' running mean advancing -- progressive
//    rendering, the path tracer;
//  * submitFrames() / collectFrames(): the throughput path -- BATCHES of independent frames (own jitter seeds) per
//    exchange, every rank rendering its share of a whole batch in one set of launches (vrhip_render_batch),
//    packing it on the GPU into the sparse message of tiles.py (a tile of one colour travels as one pixel:
//    vrhip_pack_tiles), ONE exchange in flight on streams of its own while the next batch renders into a second
//    set of buffers, and the root assembling a batch straight from the messages (vrhip_message_positions +
//    vrhip_assemble_batch).
#pragma once
#include <cstddef>
#include <cstdint>
#include <deque>
#include <string>
#include <vector>

class VolumeRenderCL;

// owner[t] of the tiles of a width x height frame (numbered row-major) over n ranks: tiles sorted by the
// distance of their centre from the frame's centre (integers, ties by id) and dealt like cards, back and
// forth (0 1 .. n-1 n-1 .. 1 0 0 1 ..), so that every rank gets tiles of every distance (load balance:
// a tile's cost follows the object in the middle of the view).  Same rule as tiles.py deal_tiles.
// root_share < 1: rank 0, which also assembles the frames, takes that fraction of a peer's tiles.
std::vector<unsigned int> vr_deal_tiles(size_t width, size_t height, size_t tile, size_t n, double root_share = 1.0);

class TileGather
{
public:
    // ranks[r] renders on HIP device devices[r].  loopback: no RCCL -- the peers' tiles reach the root
    // by device-to-device copies (every rank on ONE device: rehearsal of everything but the transport).
    // force_gather: the root does not render into its block of the staging buffer but into a buffer of
    // its own and SENDS it to itself inside the same RCCL group as the peers' sends (one ncclSend /
    // ncclRecv pair on the root's communicator and stream): with one rank this runs the transport --
    // ncclCommInitAll, a grouped point-to-point exchange, the assembly behind the receive -- on a
    // one-GPU box.  root_share: see vr_deal_tiles.  batch_frames > 0 prepares submitFrames() for batches of up
    // to that many frames (two sets of buffers).
    TileGather(const std::vector<VolumeRenderCL *> &ranks, const std::vector<int> &devices, size_t width,
               size_t height, size_t tile, bool loopback, bool force_gather = false, double root_share = 1.0,
               size_t batch_frames = 0);
    ~TileGather();
    TileGather(const TileGather &) = delete;
    TileGather &operator=(const TileGather &) = delete;

    // One frame: every rank renders its tiles (advancing its running mean), gather, assembly, copy
    // to `out` (width * height * 4 floats, row 0 = top).  Returns the seconds from the first launch to
    // the assembled frame on the root (host clock, streams synchronised).
    double renderFrame(std::vector<float> &out);

    // ---- batches of independent frames, one exchange in flight
    // Every rank renders its tiles of the seeds.size() <= batch_frames frames (frame f jittered by seeds[f]) with one
    // set of launches and packs its message; then the exchange of the batch submitted BEFORE this one is started
    // (its counts have reached the host by now, and this batch is already queued on the GPUs).  At most two
    // batches may be pending.
    void submitFrames(const std::vector<unsigned int> &seeds);
    // Finishes the oldest pending batch: exchange (if not started yet), assembly on the root; returns the root's
    // device buffer [n][height][width][4] holding its frames -- valid until the batch after the next is submitted --
    // and copies them to *host_out when that is not null.
    const float *collectFrames(std::vector<float> *host_out, size_t *n_frames = nullptr);
    size_t pending() const { return _pending.size(); }
    // bytes that travelled to the root in the batches collected so far / what a dense gather would have moved
    double sentBytes() const { return _sent_bytes; }
    double denseBytes() const { return _dense_bytes; }
    void *rootStream() const { return _streams[0]; }

    size_t tilesOf(size_t rank) const { return _tiles[rank].size(); }
    std::string transport() const
    {
        return _loopback ? "loopback (device copies)" : _self_exchange ? "RCCL send/recv (root included)" : "RCCL send/recv";
    }

private:
    struct Pending { int b; size_t n; bool exchanged; };
    void startExchange(Pending &p);

    std::vector<VolumeRenderCL *> _ranks;
    std::vector<int> _devices;
    size_t _W, _H, _tile, _tiles_x, _tiles_y, _cap;
    bool _loopback, _self_exchange;
    std::vector<std::vector<unsigned int>> _tiles;   // tile ids per rank
    std::vector<float *> _local;                     // per rank: its tiles (rank 0: block 0 of the staging)
    float *_staging = nullptr;                       // root: ranks x cap tile slots
    unsigned int *_slot_of_tile = nullptr;           // root: tile id -> slot
    float *_frame = nullptr;                         // root: assembled frame
    std::vector<void *> _streams;
    std::vector<void *> _comms;                      // ncclComm_t per rank

    // batches: two sets of buffers (b = 0, 1)
    size_t _B = 0;                                   // frames per batch at most
    std::vector<void *> _comm_streams;               // per rank: the stream its side of an exchange runs on
    std::vector<float *> _btiles[2], _bmsg[2], _brecv[2];   // per rank: tile slots, packed message, (root) received message
    std::vector<int32_t *> _bscratch[2];
    std::vector<uint32_t *> _bcount[2];              // per rank: device count of whole tiles
    uint32_t *_hcount[2] = {nullptr, nullptr};       // pinned host: counts of all ranks
    std::vector<void *> _ev_packed[2], _ev_sent[2];  // hipEvent_t per rank
    void *_ev_assembled[2] = {nullptr, nullptr};
    bool _sent_valid[2] = {false, false};
    int32_t *_bpos[2] = {nullptr, nullptr};          // root: world x S positions
    float *_bframes[2] = {nullptr, nullptr};         // root: the assembled frames of a batch
    uint32_t *_rank_slot = nullptr;                  // root: tile id -> rank << 16 | slot
    std::deque<Pending> _pending;
    int _next_buf = 0;
    double _sent_bytes = 0.0, _dense_bytes = 0.0;
};
