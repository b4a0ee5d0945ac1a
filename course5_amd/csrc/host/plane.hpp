// `plane` and `object2d` with the reference's public interface (plane.hpp:17-33, object2d.hpp:13-22),
// implemented on the C ABI of libcourse5_hip.so instead of the OpenMP pixel loops, on one GPU or on
// several GPUs of one node (one c5_context and one stream per device, this one process driving them all).
#pragma once

#include <array>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "course5_hip.h"
#include "row_blocks.hpp"
#include "scene.hpp"

// Pinned host images, recycled: a 2400x1800 frame is 34.6 MB, and pinning memory costs milliseconds.
// Portable pinned memory (hipHostMallocPortable): with several GPUs every device's copy engine writes its rows into
// the same image.  take() and the hand-back may run on different threads (a frame is handed back by whoever lets
// go of it last: the thread that writes the file).
class image_pool : public std::enable_shared_from_this<image_pool> {
public:
    explicit image_pool(std::size_t bytes) : _bytes(bytes) {}
    ~image_pool();
    std::shared_ptr<float> take();  // returned to the pool when the last owner lets go
    std::size_t bytes() const { return _bytes; }

private:
    std::size_t _bytes;
    std::mutex _lock;
    std::vector<float*> _free;
};

// Two-channel fp32 image, row-major [y][x][2] (the order export_to_vti writes, object2d.cpp:17-21).
class object2d {
public:
    object2d() = default;
    object2d(std::vector<float> pixels, std::size_t res_x, std::size_t res_y)
        : _owned(std::make_shared<std::vector<float>>(std::move(pixels))), _res_x(res_x), _res_y(res_y) {
        _pixels = std::shared_ptr<float>(_owned, _owned->data());
    }
    object2d(std::shared_ptr<float> pinned, std::size_t res_x, std::size_t res_y)
        : _pixels(std::move(pinned)), _res_x(res_x), _res_y(res_y) {}
    void export_to_vti(const std::string& filename) const;  // object2d.cpp:7-29
    // colour-mapped PNG of one channel over [lo, hi]; lo == hi == 0 and !fixed: the channel's own finite range
    void export_to_png(const std::string& filename) const;
    float at(std::size_t x, std::size_t y, std::size_t channel) const { return _pixels.get()[(y * _res_x + x) * 2 + channel]; }
    std::size_t res_x() const { return _res_x; }
    std::size_t res_y() const { return _res_y; }
    const float* data() const { return _pixels.get(); }

private:
    std::shared_ptr<std::vector<float>> _owned;
    std::shared_ptr<float> _pixels;
    std::size_t _res_x = 0, _res_y = 0;
};

// How the rows rendered by several GPUs come together:
//   host  every GPU copies its row tiles straight to their places in the pinned host image over its OWN
//         PCIe link (c5_render_frame_rows_async): no exchange between GPUs, no reassembly, frames pipelined.
//         The default: plane::trace_rays hands back host pixels (plane.cpp:144-172) anyway.
//   rccl  grouped ncclSend / ncclRecv over xGMI, every tile received at its final offset of the root GPU's
//         full image, then one copy to the host from there (north star: "RCCL gather of tile strips");
//   p2p   the same exchange as peer-to-peer 2-D copies (hipMemcpy2DAsync, the copy engines instead of a kernel).
enum class exchange_mode { host, rccl, p2p };

// Which rows a GPU renders when several share a frame:
//   blocks  ONE contiguous block of rows per GPU, sized by measured cost (segments per row of an earlier frame +
//           a base cost per pixel): a block is contiguous in the [y][x][2] image, so it travels as ONE message /
//           ONE copy per GPU and frame and lands at its final offset; every GPU builds only the records its rays
//           can reach.  The default.  Re-measured every kProbeEvery frames of a sweep, moved when it pays.
//   tiles   cyclic tiles of 16 rows (tile t -> GPU t mod N): balanced by construction, one message per tile.
enum class row_layout { blocks, tiles };

struct multi_gpu;  // RCCL communicators, peer access, per-device strips (plane.cpp)

class plane {
public:
    plane() = delete;
    // plane.cpp:260-315.  objects3d: volume grids (transparent) and solids, in the reference's order;
    // global_boundaries = {x_max, x_min, y_max, y_min}; empty: the bounding box of the transformed objects
    // (plane.cpp:278-288; the reference's own main always passes the domain, main.cpp:83,127).  devices: GPU ordinals, one context
    // each; with more than one the image rows are dealt to them in cyclic tiles of 16 rows.
    explicit plane(std::size_t res_x, std::size_t res_y, std::vector<object3d_base> objects3d,
                   std::vector<double> global_boundaries = {}, std::vector<int> devices = {0},
                   exchange_mode exchange = exchange_mode::host, row_layout layout = row_layout::blocks);
    ~plane();
    plane(const plane&) = delete;
    plane& operator=(const plane&) = delete;

    // plane.cpp:184-192: starts a frame on the GPU(s) (view transform, records, entries, solids, walk) and
    // its way to host memory; returns at once.  Up to C5_HOST_RING frames may be in flight.
    void find_intersections();
    // plane.cpp:144-172: waits for the OLDEST frame in flight and returns its image.  The two arguments
    // select the cell values for ch0/ch1 as in the reference; only (alpha, Q) is meaningful there and here.
    object2d trace_rays(tetra_value value_alpha, tetra_value value_Q);
    std::size_t count_all_intersections();  // plane.cpp:3-12 (segments of the last frame, all devices)

    std::size_t get_x() const { return _x; }
    std::size_t get_y() const { return _y; }
    std::size_t frames_in_flight() const { return _flight.size() + _parked.size(); }
    std::size_t retries() const { return _retries; }

    struct views_t {
        std::vector<c5_rotation> grid;
        std::vector<std::vector<c5_rotation>> solids;  // by solid slot
    };
    // Re-send the objects' rotation lists (a sweep changes only these; the grid stays on the GPU).
    void update_views(std::vector<object3d_base>& objects3d) { set_views(views_of(objects3d)); }
    // The same in two steps, for drivers that work the angles out on one thread and issue frames on another
    // (views_of reads the objects and nothing of the plane that changes after its construction).
    views_t views_of(std::vector<object3d_base>& objects3d) const;
    void set_views(const views_t& v);
    // of the last completed frame: counts summed over the devices, times of the slowest.  May be called with frames
    // in flight: a C5_RETRY it runs into (an internal buffer grew) is remembered and honoured by the next trace_rays.
    c5_stats stats();
    // the rows [begin, begin + count) every device renders (blocks layout) — for logs and tests
    std::vector<std::pair<int, int>> row_blocks() const { return _blocks; }
    std::size_t rebalances() const { return _rebalances; }
    // The library's per-stage GPU times (c5_stats::ms_*) come from six events per frame, 25 us of a 0.55-ms frame: a sweep
    // that does not print them switches them off (probe frames of the blocks layout keep theirs: they time the blocks).
    void stage_times(bool on) { _stage_times = on; }

private:
    struct frame_t {
        std::shared_ptr<float> image;
        views_t views;
        bool probe = false;  // the walk counted segments per row for this frame (blocks layout)
    };
    void check(int rc, const char* what, std::size_t dev = 0);
    void send_views(const views_t& v);
    void start(frame_t& f);                 // enqueue on every device
    void finish_exchange(frame_t& f);       // rccl / p2p: strips -> root image -> host (synchronous)
    void complete_front();                  // wait for the oldest frame in flight (renders again on C5_RETRY)
    void read_row_costs();                  // after a probe frame: costs of all rows, and whether other blocks would pay
    void apply_blocks(const std::vector<std::pair<int, int>>& blocks);
    bool retry_seen();                      // any device's stats() ran into C5_RETRY since the last look (cleared)
    std::vector<c5_context*> _ctx;
    std::vector<int> _devices;
    exchange_mode _exchange = exchange_mode::host;
    std::size_t _x = 0, _y = 0;
    std::vector<int> _slot_of_object;  // -1: part of the volume grid, >= 0: solid slot
    views_t _views;
    std::deque<frame_t> _flight;
    std::deque<frame_t> _parked;  // completed ahead of their trace_rays (a rebalance drained the frames in flight)
    std::shared_ptr<image_pool> _pool;
    std::unique_ptr<multi_gpu> _mg;
    std::size_t _retries = 0;
    row_layout _layout = row_layout::blocks;
    std::vector<std::pair<int, int>> _blocks, _wanted_blocks;  // per device: first row, rows
    std::vector<uint32_t> _row_cost;                           // segments per image row of the last probe frame
    std::vector<char> _retry_seen;
    std::size_t _issued = 0, _rebalances = 0;
    bool _rebalance_due = false;
    bool _stage_times = true;
    std::vector<int> _stage_on;  // per device: what "stage_timing" was last set to (-1: not yet)
};

// {x_max, x_min, y_max, y_min} of the objects' transformed vertices: what the reference's plane uses when no
// boundaries are given (plane.cpp:278-288)
std::array<double, 4> bounding_box(std::vector<object3d_base>& objects3d);

// What a one-GPU box can check of the RCCL exchange: librccl loads, every entry point the exchange uses resolves,
// a communicator comes up (ncclCommInitAll over the one device), and the exchange's own call patterns — (tiles) a
// group of ncclSend / ncclRecv pairs that land 16-row tiles at their final offsets of a frame, (blocks) ONE pair per
// peer landing a whole block of rows at its offset — move the right bytes, the device sending to itself.  Returns a
// line for the log; throws on any failure.
std::string rccl_selftest(int device);

// "0", "0-7", "0,2,4", "0,0" (the same GPU twice: rehearsal of the multi-GPU path on one GPU)
std::vector<int> parse_device_list(const std::string& text);
