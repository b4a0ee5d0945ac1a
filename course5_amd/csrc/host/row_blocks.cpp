// Contiguous row blocks of (nearly) equal cost for the GPUs that share a frame (plane.hpp: row_layout::blocks).
// Host-only arithmetic, no HIP: also built into the sanitizer harness (tests/cpp/host_san_main.cpp), where
// tests/test_sanitizers_cpu.py compares it with course5_amd/sharding.py: balanced_blocks on random cost vectors.
#include "row_blocks.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>

std::vector<std::pair<int, int>> balanced_row_blocks(const std::vector<uint32_t>& row_cost, int world, double base_cost_per_row) {
    const int res_y = static_cast<int>(row_cost.size());
    if (world < 1 || res_y < world) throw std::runtime_error("more devices than image rows");
    std::vector<double> cum(static_cast<std::size_t>(res_y) + 1, 0.0);
    for (int r = 0; r < res_y; ++r) cum[static_cast<std::size_t>(r) + 1] = cum[static_cast<std::size_t>(r)] + static_cast<double>(row_cost[static_cast<std::size_t>(r)]) + base_cost_per_row;
    const double total = cum.back();
    std::vector<int> edge{0};
    for (int r = 1; r < world; ++r) {
        const double target = total * r / world;
        int e = static_cast<int>(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());  // first cum[e] >= target
        // nearer of the two candidate cuts
        if (e > 0 && std::fabs(cum[static_cast<std::size_t>(e) - 1] - target) <= std::fabs(cum[static_cast<std::size_t>(std::min(e, res_y))] - target)) --e;
        e = std::max(e, edge.back() + 1);
        e = std::min(e, res_y - (world - r));
        edge.push_back(e);
    }
    edge.push_back(res_y);
    std::vector<std::pair<int, int>> blocks;
    for (int r = 0; r < world; ++r) blocks.emplace_back(edge[static_cast<std::size_t>(r)], edge[static_cast<std::size_t>(r) + 1] - edge[static_cast<std::size_t>(r)]);
    return blocks;
}

