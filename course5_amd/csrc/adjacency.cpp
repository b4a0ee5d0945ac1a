#include "adjacency.hpp"

#include <algorithm>

namespace c5 {

namespace {
struct FaceKey {
    uint32_t a, b, c;  // sorted point ids
    uint32_t ref;      // cell << 2 | face
};
inline bool key_less(const FaceKey& l, const FaceKey& r) {
    if (l.a != r.a) return l.a < r.a;
    if (l.b != r.b) return l.b < r.b;
    return l.c < r.c;
}
inline bool key_eq(const FaceKey& l, const FaceKey& r) { return l.a == r.a && l.b == r.b && l.c == r.c; }
}  // namespace

bool build_face_adjacency(const int32_t* cell_vert, int64_t n_cells, int64_t n_pts,
                          std::vector<int32_t>& adj, std::vector<uint32_t>& bfaces, std::string& err) {
    static const int FV[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
    std::vector<FaceKey> keys(static_cast<size_t>(4 * n_cells));
    bool bad_id = false;
#pragma omp parallel for schedule(static) reduction(|| : bad_id)
    for (int64_t c = 0; c < n_cells; ++c) {
        const int32_t* v = cell_vert + 4 * c;
        for (int k = 0; k < 4; ++k)
            if (v[k] < 0 || v[k] >= n_pts) bad_id = true;
        for (int f = 0; f < 4; ++f) {
            uint32_t t[3] = {static_cast<uint32_t>(v[FV[f][0]]), static_cast<uint32_t>(v[FV[f][1]]),
                             static_cast<uint32_t>(v[FV[f][2]])};
            if (t[0] > t[1]) std::swap(t[0], t[1]);
            if (t[1] > t[2]) std::swap(t[1], t[2]);
            if (t[0] > t[1]) std::swap(t[0], t[1]);
            keys[static_cast<size_t>(4 * c + f)] = FaceKey{t[0], t[1], t[2], static_cast<uint32_t>(4 * c + f)};
        }
    }
    if (bad_id) {
        err = "cell references a point id out of range";
        return false;
    }
    std::sort(keys.begin(), keys.end(), key_less);

    adj.assign(static_cast<size_t>(4 * n_cells), -1);
    bfaces.clear();
    const size_t n = keys.size();
    size_t i = 0;
    while (i < n) {
        size_t j = i + 1;
        while (j < n && key_eq(keys[i], keys[j])) ++j;
        const size_t run = j - i;
        if (run == 1) {
            bfaces.push_back(keys[i].ref);
        } else if (run == 2) {
            adj[keys[i].ref] = static_cast<int32_t>(keys[i + 1].ref >> 2);
            adj[keys[i + 1].ref] = static_cast<int32_t>(keys[i].ref >> 2);
        } else {
            err = "non-conforming grid: a face is shared by more than two cells";
            return false;
        }
        i = j;
    }
    std::sort(bfaces.begin(), bfaces.end());
    return true;
}

}  // namespace c5
