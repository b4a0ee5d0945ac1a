#include "adjacency.hpp"

#include <omp.h>
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>

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

// Threads for the host-side loops: what OpenMP offers, but no more than the CPUs this process may
// really use.  A container with a CPU quota still shows every core of the machine (256 here for a
// quota of 16): a team of 256 burns the quota in a few milliseconds and the whole process — the
// thread that enqueues the frames included — is then throttled for the rest of each period.
int host_threads() {
    static const int n = [] {
        long cpus = omp_get_max_threads();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min<long>(cpus, CPU_COUNT(&set));
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char quota[32] = {0};
            long period = 0;
            if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0)
                cpus = std::min(cpus, std::max(1L, std::atol(quota) / period));
            std::fclose(f);
        } else {
            long quota = -1, period = 0;  // cgroup v1
            if (FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(q, "%ld", &quota) != 1) quota = -1;
                std::fclose(q);
            }
            if (FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(q, "%ld", &period) != 1) period = 0;
                std::fclose(q);
            }
            if (quota > 0 && period > 0) cpus = std::min(cpus, std::max(1L, quota / period));
        }
        return static_cast<int>(std::max(1L, std::min(cpus, 64L)));
    }();
    return n;
}

// Sort by (a, b, c) on all OpenMP threads: the keys are dealt into buckets by ranges of `a` (their
// smallest point id, spread evenly over [0, n_ids)), the buckets are sorted independently, and their
// concatenation is sorted.  (std::sort alone took 250 ms of the 290 ms of c5_upload_grid at 1M cells.)
template <class Less>
void sort_keys(std::vector<FaceKey>& keys, uint64_t n_ids, Less less) {
    const int threads = host_threads();
    if (threads < 2 || keys.size() < (size_t{1} << 16) || n_ids == 0) {
        std::sort(keys.begin(), keys.end(), less);
        return;
    }
    const int n_buckets = threads * 8;
    auto bucket_of = [&](const FaceKey& k) {
        const uint64_t b = static_cast<uint64_t>(k.a) * static_cast<uint64_t>(n_buckets) / n_ids;
        return static_cast<int>(b < static_cast<uint64_t>(n_buckets) ? b : n_buckets - 1);
    };
    const int64_t n = static_cast<int64_t>(keys.size());
    std::vector<size_t> hist(static_cast<size_t>(threads) * n_buckets, 0);
    std::vector<FaceKey> tmp(keys.size());
    std::vector<size_t> start(static_cast<size_t>(n_buckets) + 1, 0);
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num();
        const int64_t lo = n * t / threads, hi = n * (t + 1) / threads;
        size_t* mine = hist.data() + static_cast<size_t>(t) * n_buckets;
        for (int64_t i = lo; i < hi; ++i) ++mine[bucket_of(keys[static_cast<size_t>(i)])];
#pragma omp barrier
#pragma omp single
        {   // bucket b of thread t starts after all of bucket b - 1 and after bucket b of the threads before t
            size_t run = 0;
            for (int b = 0; b < n_buckets; ++b) {
                start[static_cast<size_t>(b)] = run;
                for (int u = 0; u < threads; ++u) {
                    const size_t c = hist[static_cast<size_t>(u) * n_buckets + b];
                    hist[static_cast<size_t>(u) * n_buckets + b] = run;
                    run += c;
                }
            }
            start[static_cast<size_t>(n_buckets)] = run;
        }
        for (int64_t i = lo; i < hi; ++i) {
            const FaceKey& k = keys[static_cast<size_t>(i)];
            tmp[mine[bucket_of(k)]++] = k;
        }
#pragma omp barrier
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < n_buckets; ++b)
            std::sort(tmp.begin() + static_cast<std::ptrdiff_t>(start[static_cast<size_t>(b)]),
                      tmp.begin() + static_cast<std::ptrdiff_t>(start[static_cast<size_t>(b) + 1]), less);
    }
    keys.swap(tmp);
}

// Comparison sort on all OpenMP threads, for any strict weak order: equal chunks sorted independently,
// then merged pairwise (the last merges run on few threads, but a merge is a single linear pass).
template <class T, class Less>
void parallel_sort(std::vector<T>& v, Less less) {
    const size_t n = v.size();
    int parts = 1;
    while (parts * 2 <= host_threads() && parts < 64) parts *= 2;
    if (parts < 2 || n < (size_t{1} << 16)) {
        std::sort(v.begin(), v.end(), less);
        return;
    }
    std::vector<size_t> edge(static_cast<size_t>(parts) + 1);
    for (int p = 0; p <= parts; ++p) edge[static_cast<size_t>(p)] = n * static_cast<size_t>(p) / static_cast<size_t>(parts);
#pragma omp parallel for schedule(static) num_threads(parts)
    for (int p = 0; p < parts; ++p)
        std::sort(v.begin() + static_cast<std::ptrdiff_t>(edge[static_cast<size_t>(p)]),
                  v.begin() + static_cast<std::ptrdiff_t>(edge[static_cast<size_t>(p) + 1]), less);
    std::vector<T> other(n);
    T* src = v.data();
    T* dst = other.data();
    for (int width = 1; width < parts; width *= 2) {
        const int pairs = parts / (2 * width);
#pragma omp parallel for schedule(static) num_threads(pairs)
        for (int k = 0; k < pairs; ++k) {
            const size_t lo = edge[static_cast<size_t>(2 * width * k)], mid = edge[static_cast<size_t>(2 * width * k + width)],
                         hi = edge[static_cast<size_t>(2 * width * (k + 1))];
            std::merge(src + lo, src + mid, src + mid, src + hi, dst + lo, less);
        }
        std::swap(src, dst);
    }
    if (src != v.data()) v.swap(other);
}
}  // namespace

void parallel_copy(void* dst, const void* src, size_t bytes) {
    const int threads = std::min(host_threads(), 8);
    if (threads < 2 || bytes < (size_t{1} << 20)) {
        std::memcpy(dst, src, bytes);
        return;
    }
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int t = 0; t < threads; ++t) {
        const size_t lo = (bytes * static_cast<size_t>(t) / static_cast<size_t>(threads)) & ~size_t{63};
        const size_t hi = t + 1 == threads ? bytes : (bytes * static_cast<size_t>(t + 1) / static_cast<size_t>(threads)) & ~size_t{63};
        std::memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, hi - lo);
    }
}

int64_t weld_points(const double* xyz, int64_t n_pts, std::vector<int32_t>& rep) {
    std::vector<uint32_t> order(static_cast<size_t>(n_pts));
    for (int64_t i = 0; i < n_pts; ++i) order[static_cast<size_t>(i)] = static_cast<uint32_t>(i);
    // by value (the coordinates are finite: c5_upload_grid checks), so -0.0 and 0.0 are one point; ties by id
    auto pt_less = [&](uint32_t l, uint32_t r) {
        for (int k = 0; k < 3; ++k) {
            const double a = xyz[3 * static_cast<size_t>(l) + k], b = xyz[3 * static_cast<size_t>(r) + k];
            if (a != b) return a < b;
        }
        return l < r;
    };
    auto pt_eq = [&](uint32_t l, uint32_t r) {
        for (int k = 0; k < 3; ++k)
            if (xyz[3 * static_cast<size_t>(l) + k] != xyz[3 * static_cast<size_t>(r) + k]) return false;
        return true;
    };
    parallel_sort(order, pt_less);
    rep.resize(static_cast<size_t>(n_pts));
    int64_t merged = 0;
    uint32_t head = 0;
    for (size_t i = 0; i < order.size(); ++i) {
        if (i == 0 || !pt_eq(order[i - 1], order[i])) head = order[i];  // smallest id of the run (ties sort by id)
        else ++merged;
        rep[order[i]] = static_cast<int32_t>(head);
    }
    return merged;
}

bool build_face_adjacency(const int32_t* cell_vert, int64_t n_cells, int64_t n_pts,
                          std::vector<int32_t>& adj, std::vector<uint32_t>& bfaces, std::string& err) {
    static const int FV[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
    std::vector<FaceKey> keys(static_cast<size_t>(4 * n_cells));
    bool bad_id = false;
#pragma omp parallel for schedule(static) reduction(|| : bad_id) num_threads(host_threads())
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
    // a cell that names a point twice (possible after welding coincident points) has no volume and two
    // identical faces: it would become its own neighbour.  The reference bins and sorts such a cell like any
    // other (its chords are zero); so does bin_sort_resolve, which takes grids a walk cannot.
    bool repeated = false;
#pragma omp parallel for schedule(static) reduction(|| : repeated) num_threads(host_threads())
    for (int64_t c = 0; c < n_cells; ++c) {
        const int32_t* v = cell_vert + 4 * c;
        if (v[0] == v[1] || v[0] == v[2] || v[0] == v[3] || v[1] == v[2] || v[1] == v[3] || v[2] == v[3]) repeated = true;
    }
    if (repeated) {
        err = "non-conforming grid: a cell names the same point twice";
        return false;
    }
    sort_keys(keys, static_cast<uint64_t>(n_pts), key_less);

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

void unique_solid_faces(const double* tets, int64_t n_tets, std::vector<double>& points,
                        std::vector<int32_t>& faces) {
    static const int FV[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
    const int64_t n_raw = 4 * n_tets;
    // 1. unique points: sort raw point indices by the bit patterns of (x, y, z)
    auto bits = [](double v) {
        uint64_t u;
        std::memcpy(&u, &v, sizeof u);
        return u;
    };
    std::vector<uint32_t> order(static_cast<size_t>(n_raw));
    for (int64_t i = 0; i < n_raw; ++i) order[static_cast<size_t>(i)] = static_cast<uint32_t>(i);
    auto pt_less = [&](uint32_t l, uint32_t r) {
        for (int k = 0; k < 3; ++k) {
            const uint64_t a = bits(tets[3 * static_cast<size_t>(l) + k]), b = bits(tets[3 * static_cast<size_t>(r) + k]);
            if (a != b) return a < b;
        }
        return false;
    };
    auto pt_eq = [&](uint32_t l, uint32_t r) {
        for (int k = 0; k < 3; ++k)
            if (bits(tets[3 * static_cast<size_t>(l) + k]) != bits(tets[3 * static_cast<size_t>(r) + k])) return false;
        return true;
    };
    parallel_sort(order, pt_less);
    std::vector<int32_t> id_of(static_cast<size_t>(n_raw));
    points.clear();
    for (size_t i = 0; i < order.size(); ++i) {
        if (i == 0 || !pt_eq(order[i - 1], order[i])) {
            for (int k = 0; k < 3; ++k) points.push_back(tets[3 * static_cast<size_t>(order[i]) + k]);
        }
        id_of[order[i]] = static_cast<int32_t>(points.size() / 3 - 1);
    }
    // 2. unique faces
    std::vector<FaceKey> keys(static_cast<size_t>(n_raw));
    for (int64_t t = 0; t < n_tets; ++t) {
        for (int f = 0; f < 4; ++f) {
            uint32_t v[3] = {static_cast<uint32_t>(id_of[static_cast<size_t>(4 * t + FV[f][0])]),
                             static_cast<uint32_t>(id_of[static_cast<size_t>(4 * t + FV[f][1])]),
                             static_cast<uint32_t>(id_of[static_cast<size_t>(4 * t + FV[f][2])])};
            if (v[0] > v[1]) std::swap(v[0], v[1]);
            if (v[1] > v[2]) std::swap(v[1], v[2]);
            if (v[0] > v[1]) std::swap(v[0], v[1]);
            keys[static_cast<size_t>(4 * t + f)] = FaceKey{v[0], v[1], v[2], static_cast<uint32_t>(4 * t + f)};
        }
    }
    parallel_sort(keys, [](const FaceKey& l, const FaceKey& r) {
        if (l.a != r.a) return l.a < r.a;
        if (l.b != r.b) return l.b < r.b;
        if (l.c != r.c) return l.c < r.c;
        return l.ref < r.ref;
    });
    // keep the first occurrence of every face, then restore the generator's order: consecutive cells
    // of init_polar are angular neighbours, so consecutive faces cover neighbouring pixels of the same
    // image rows (coalesced mask accesses); faces through vertex 0 (the fan centre: long slivers) go
    // first and the short surface faces (slot 3) last, so a wavefront holds faces of similar height.
    //
    // INTERIOR faces.  A face that has a cell of non-zero volume on EITHER side covers no pixel its solid's other faces
    // do not cover: a ray through it runs on through both cells and leaves each through another face, and so on until
    // it meets a face with nothing behind it (finitely many cells, the depth grows with every cell) - the mask is the
    // union over all faces (plane.cpp:130-131 -> line.cpp:246-249), so such a face is pure overdraw: on the centre-fan
    // solids of the reference the three fan faces of every cell (long slivers from the centre to the surface: 460 times
    // the pixels of the surface triangles together).  Marked in the face's fourth word and sorted to the front, so that
    // the raster can start behind them (c_api.hip, option "solid_interior_faces").  Sides by the sign of the cell's fourth
    // vertex against the face through its three (sorted, hence shared) point ids; a cell too flat to have a side
    // (|det| <= 1e-9 of the product of its edge lengths: the duplicated points at the poles of init_polar) counts for
    // neither, and a face with cells on one side only is a boundary face whatever its multiplicity.
    auto side_of = [&](const FaceKey& k) {
        const uint32_t t = k.ref >> 2, f = k.ref & 3u;
        const double* pa = &points[3 * static_cast<size_t>(k.a)];
        const double* pb = &points[3 * static_cast<size_t>(k.b)];
        const double* pc = &points[3 * static_cast<size_t>(k.c)];
        const double* pd = &tets[3 * (4 * static_cast<size_t>(t) + (3u - f))];  // FV[f] leaves out vertex 3 - f
        const double u[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        const double v[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
        const double w[3] = {pd[0] - pa[0], pd[1] - pa[1], pd[2] - pa[2]};
        const double det = u[0] * (v[1] * w[2] - v[2] * w[1]) - u[1] * (v[0] * w[2] - v[2] * w[0]) + u[2] * (v[0] * w[1] - v[1] * w[0]);
        auto len = [](const double* q) { return std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]); };
        const double scale = len(u) * len(v) * len(w);
        if (!(std::fabs(det) > 1e-9 * scale)) return 0;
        return det > 0.0 ? 1 : -1;
    };
    std::vector<FaceKey> uniq;
    std::vector<uint8_t> interior_of_ref(static_cast<size_t>(n_raw), 0);
    uniq.reserve(keys.size());
    for (size_t i = 0; i < keys.size();) {
        size_t j = i;
        bool above = false, below = false;
        for (; j < keys.size() && key_eq(keys[i], keys[j]); ++j) {
            const int sd = side_of(keys[j]);
            above = above || sd > 0;
            below = below || sd < 0;
        }
        uniq.push_back(keys[i]);
        interior_of_ref[keys[i].ref] = (above && below) ? 1 : 0;
        i = j;
    }
    parallel_sort(uniq, [&](const FaceKey& l, const FaceKey& r) {
        const bool li = interior_of_ref[l.ref] != 0, ri = interior_of_ref[r.ref] != 0;
        if (li != ri) return li;  // interior faces first
        const bool ls = (l.ref & 3u) == 3u, rs = (r.ref & 3u) == 3u;
        if (ls != rs) return !ls;
        return l.ref < r.ref;
    });
    faces.clear();
    faces.reserve(4 * uniq.size());
    for (const FaceKey& k : uniq) {
        faces.push_back(static_cast<int32_t>(k.a));
        faces.push_back(static_cast<int32_t>(k.b));
        faces.push_back(static_cast<int32_t>(k.c));
        faces.push_back(interior_of_ref[k.ref]);
    }
}

}  // namespace c5
