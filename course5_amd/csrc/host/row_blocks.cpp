// Contiguous row blocks of (nearly) equal cost for the GPUs that share a frame (plane.hpp: row_layout::blocks).
// Host-only arithmetic, no HIP: also built into the sanitizer harness (tests/cpp/host_san_main.cpp), where
// tests/test_sanitizers_cpu.py compares it with course5_amd/sharding.py: balanced_blocks on random cost vectors.
#include "row_blocks.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>

std::vector<std::pair<int, int>> balanced_row_blocks(const std::vector<uint32_t>& row_cost, int world, double base_cost_per_row, int quantum) {
    std::vector<double> c(row_cost.size());
    for (std::size_t r = 0; r < row_cost.size(); ++r) c[r] = static_cast<double>(row_cost[r]) + base_cost_per_row;
    return balanced_row_blocks(c, world, quantum);
}

std::vector<double> time_weighted_row_costs(const std::vector<uint32_t>& row_cost, double base_cost_per_row,
                                            const std::vector<std::pair<int, int>>& blocks, const std::vector<double>& times) {
    std::vector<double> c(row_cost.size());
    for (std::size_t r = 0; r < row_cost.size(); ++r) c[r] = static_cast<double>(row_cost[r]) + base_cost_per_row;
    for (std::size_t k = 0; k < blocks.size() && k < times.size(); ++k) {
        const int b = blocks[k].first, n = blocks[k].second;
        double s = 0.0;
        for (int r = b; r < b + n; ++r) s += c[static_cast<std::size_t>(r)];
        if (n > 0 && s > 0.0 && times[k] > 0.0)
            for (int r = b; r < b + n; ++r) c[static_cast<std::size_t>(r)] *= times[k] / s;
    }
    return c;
}

std::vector<std::pair<int, int>> balanced_row_blocks(const std::vector<double>& row_cost, int world, int quantum) {
    const int res_y = static_cast<int>(row_cost.size());
    if (world < 1 || res_y < world) throw std::runtime_error("more devices than image rows");
    std::vector<double> cum(static_cast<std::size_t>(res_y) + 1, 0.0);
    for (int r = 0; r < res_y; ++r) cum[static_cast<std::size_t>(r) + 1] = cum[static_cast<std::size_t>(r)] + row_cost[static_cast<std::size_t>(r)];
    const double total = cum.back();
    std::vector<int> edge{0};
    for (int r = 1; r < world; ++r) {
        const double target = total * r / world;
        int e = static_cast<int>(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());  // first cum[e] >= target
        // nearer of the two candidate cuts
        if (e > 0 && std::fabs(cum[static_cast<std::size_t>(e) - 1] - target) <= std::fabs(cum[static_cast<std::size_t>(std::min(e, res_y))] - target)) --e;
        if (quantum > 1 && res_y >= 2 * quantum * world) e = (e + quantum / 2) / quantum * quantum;
        e = std::max(e, edge.back() + 1);
        e = std::min(e, res_y - (world - r));
        edge.push_back(e);
    }
    edge.push_back(res_y);
    std::vector<std::pair<int, int>> blocks;
    for (int r = 0; r < world; ++r) blocks.emplace_back(edge[static_cast<std::size_t>(r)], edge[static_cast<std::size_t>(r) + 1] - edge[static_cast<std::size_t>(r)]);
    return blocks;
}

