#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#define __host__
#define __device__
#include "device_types.hpp"
int main() {
    using namespace c5;
    long checked = 0;
    for (int world = 1; world <= 5; ++world)
        for (int rank = 0; rank < world; ++rank)
            for (int tile_rows : {1, 3, 16})
                for (int row_begin : {0, 7})
                    for (int row_count : {1, 5, 16, 47, 100}) {
                        ImageParams im{};
                        im.res_x = 10; im.res_y = 200; im.tile_rows = tile_rows; im.rank = rank; im.world = world;
                        im.row_begin = row_begin; im.row_count = row_count;
                        int n = 0;
                        for (int r = 0; r < im.res_y; ++r) if (local_row_of(im, r) >= 0) ++n;
                        im.n_local_rows = n;
                        for (int g0 = row_begin; g0 < row_begin + row_count; ++g0)
                            for (int g1 = g0; g1 < row_begin + row_count; ++g1) {
                                int first = 1 << 30, last = -1;
                                for (int r = g0; r <= g1; ++r) {
                                    const int l = local_row_of(im, r);
                                    if (l >= 0) { if (l < first) first = l; if (l > last) last = l; }
                                }
                                int f, l;
                                local_row_span(im, g0, g1, f, l);
                                const bool empty_want = last < 0, empty_got = l < f;
                                if (empty_want != empty_got || (!empty_want && (f != first || l != last))) {
                                    printf("MISMATCH world %d rank %d tile %d begin %d count %d g [%d,%d]: want [%d,%d] got [%d,%d]\n",
                                           world, rank, tile_rows, row_begin, row_count, g0, g1, first, last, f, l);
                                    return 1;
                                }
                                if (!empty_want)
                                    for (int k = f; k <= l; ++k) {
                                        const int g = global_row_of(im, k);
                                        if (g < g0 || g > g1 || local_row_of(im, g) != k) { printf("BAD ROW\n"); return 1; }
                                    }
                                ++checked;
                            }
                    }
    printf("ok %ld spans\n", checked);
}
