#pragma once

#include <cstdint>
#include <utility>
#include <vector>

// Contiguous blocks of (nearly) equal cost: cost of a row = its segments + base_cost_per_row (the work every pixel
// costs regardless: entry lookup, store).  Every device gets at least one row.  The same partition
// course5_amd/sharding.py: balanced_blocks computes (tests/test_sanitizers_cpu.py compares the two).
// quantum > 1: cuts at multiples of `quantum` rows (the walk's tiles are 8 rows tall and counted from a block's first row:
// a block of 241 rows costs 31 rows of tiles); ignored for images too small to give every device two quanta.
std::vector<std::pair<int, int>> balanced_row_blocks(const std::vector<uint32_t>& row_cost, int world, double base_cost_per_row, int quantum = 1);
// ... of fractional costs (the same cuts; base cost already in them)
std::vector<std::pair<int, int>> balanced_row_blocks(const std::vector<double>& row_cost, int world, int quantum = 1);
// Row costs (segments + base) rescaled block by block so that every block's total is the TIME its device took for it
// (sharding.py: time_weighted_costs): the model misses what a segment costs where (the tiles of an oblique face step half
// as fast as those of the grid's body) and what a share costs whatever its rows.
std::vector<double> time_weighted_row_costs(const std::vector<uint32_t>& row_cost, double base_cost_per_row,
                                            const std::vector<std::pair<int, int>>& blocks, const std::vector<double>& times);
