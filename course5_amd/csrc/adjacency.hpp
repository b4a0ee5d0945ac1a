// Face adjacency of a conforming tetrahedral grid (host side, run once per upload).
// The reference discards connectivity (object3d_base.cpp:13-53 copies points per cell);
// the walk needs it, so it is rebuilt here from the cell -> point-id table.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace c5 {

// adj[4 * cell + f] = neighbour across face f (reference face numbering: 0 = (0,1,2),
// 1 = (0,1,3), 2 = (0,2,3), 3 = (1,2,3)) or -1; bfaces = (cell << 2 | f) of faces without one.
// Returns false (and a message) if a face is shared by more than two cells or an id is out of range.
bool build_face_adjacency(const int32_t* cell_vert, int64_t n_cells, int64_t n_pts,
                          std::vector<int32_t>& adj, std::vector<uint32_t>& bfaces, std::string& err);

// Point ids of coincident points made equal: rep[i] = smallest id whose coordinates equal those of
// point i (rep[i] == i for a grid without duplicates).  The reference never looks at connectivity — it
// copies four points per cell (object3d_base.cpp:37-42) — so a file whose cells carry private copies of
// their points, or whose writer duplicated points along a seam, renders there like any other; here two
// cells are neighbours only if they name the SAME ids, so the ids are welded by coordinate first.
// Returns the number of points that were merged into another one.
int64_t weld_points(const double* xyz, int64_t n_pts, std::vector<int32_t>& rep);

// Solid tet soup [n][4][3] -> unique points (bitwise equal coordinates merged) and unique faces
// (unordered point-id triples; 4 per tet before merging).  The centre-fan solids of the reference
// (object3d_base.cpp:152-193) share every sliver face between two cells and every point between ~24,
// and the mask raster only depends on a face's three points, so duplicates are pure overdraw.  Faces with a cell of
// non-zero volume on either side (interior: they cover nothing the others do not) come first, marked 1.
void unique_solid_faces(const double* tets, int64_t n_tets, std::vector<double>& points,
                        std::vector<int32_t>& faces /* 4 ints per face: a, b, c, interior ? 1 : 0 */);

// memcpy on the host threads this process may use (a single thread moves ~10 GB/s: 3.5 ms for a 2400x1800
// image, five times the PCIe transfer it follows).
void parallel_copy(void* dst, const void* src, size_t bytes);

}  // namespace c5
