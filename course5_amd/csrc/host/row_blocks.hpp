#pragma once

#include <cstdint>
#include <utility>
#include <vector>

// Contiguous blocks of (nearly) equal cost: cost of a row = its segments + base_cost_per_row (the work every pixel
// costs regardless: entry lookup, store).  Every device gets at least one row.  The same partition
// course5_amd/sharding.py: balanced_blocks computes (tests/test_sanitizers_cpu.py compares the two).
std::vector<std::pair<int, int>> balanced_row_blocks(const std::vector<uint32_t>& row_cost, int world, double base_cost_per_row);
