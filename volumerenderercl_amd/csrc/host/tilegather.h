// tilegather.h -- image-tile decomposition of one frame over the GPUs of a node for the headless
// C++ host (the reference is single-GPU, volumerendercl.cpp:140).  One process, one
// renderer per GPU (volume replicated), tiles dealt to the ranks by their distance from the frame's centre
// (every rank gets tiles of every distance: load balance); every rank
// renders its tiles into a compact device buffer and rank 0 receives them over RCCL point to point
// (ncclGroupStart; ncclRecv from every peer on the root | ncclSend on the peers; ncclGroupEnd -- xGMI
// is point to point, so the root's links carry one peer each) and assembles the frame with one kernel.
#pragma once
#include <cstddef>
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
    // one-GPU box.
    TileGather(const std::vector<VolumeRenderCL *> &ranks, const std::vector<int> &devices, size_t width,
               size_t height, size_t tile, bool loopback, bool force_gather = false);
    ~TileGather();
    TileGather(const TileGather &) = delete;
    TileGather &operator=(const TileGather &) = delete;

    // One frame: every rank renders its tiles (advancing its running mean), gather, assembly, copy
    // to `out` (width * height * 4 floats, row 0 = top).  Returns the seconds from the first launch to
    // the assembled frame on the root (host clock, streams synchronised).
    double renderFrame(std::vector<float> &out);

    size_t tilesOf(size_t rank) const { return _tiles[rank].size(); }
    std::string transport() const
    {
        return _loopback ? "loopback (device copies)" : _self_exchange ? "RCCL send/recv (root included)" : "RCCL send/recv";
    }

private:
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
};
