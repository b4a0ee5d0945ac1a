// "precision" 1: the walk with single-precision geometry and double-precision accumulators.
//
// The fp64 walk (walk_kernels.hip) reproduces the reference to the last bit of its fp32 output; it reads
// 160 bytes per step and spends most of its vector instructions at the fp64 rate.  This variant keeps what the
// 1e-5 bar of the north star needs and no more:
//   * per cell 64 bytes of geometry (GeoRecord: four face planes in fp32 about an origin on the pixel lattice
//     close to the cell, so fp32 resolves 1e-7 of the CELL, not of the domain) + 16 bytes of optics: half the
//     L2 / LDS traffic per step, one staging load instruction for sixteen cells instead of two for eight each;
//   * face depths, the z_top / z_bot pairing (line.cpp:99-131) and the exit face in fp32 (v_min3 / v_max3);
//   * exp(-alpha dz) - 1 by a short fp32 series while |alpha dz| < 1/8 (always, on grids that resolve the
//     image), the general fp64 exp otherwise (one wave-uniform branch);
//   * tau and I accumulate in fp64:  tau += dz alpha (line.cpp:189);  I += (I - Q/alpha)(e^{-alpha dz} - 1),
//     which is line.cpp:220-224's  I = (Q - (Q - alpha I) e^{-alpha dz}) / alpha  rearranged.
// Per-chord error ~1e-7 relative, uncorrelated between cells; images agree with the reference to ~1e-6
// (tests: every golden vector, the fuzz sweep and the full C3 frame against the oracle, all at the 1e-5 bar).
// What it gives up: bit-equality with the fp64 walk, and with it the exact segment COUNT — a ray within
// ~1e-9 of a projected edge may count a sliver the reference does not (or the other way round).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "device_types.hpp"
#include "kernels.hpp"
#include "walk_common.hpp"
#include "walk_mixed_common.hpp"

namespace c5 {

bool mixed_precision_fits(int64_t n_cells, const ImageParams& im) {
    // 32-bit byte offsets into the 64-byte records; the lattice origin holds 16 bits per axis
    // (ids shifted by 6 and the optics behind the geometry, both inside 32 bits: 2^25 cells)
    return n_cells < (int64_t{1} << 25) && im.res_x <= 65535 && im.res_y <= 65535;
}

// ------------------------------------------------------------------------------------------
// build_records_mixed: one thread per cell.  The planes are the fp64 ones of build_records (same
// classification, same walk order, same neighbour words), re-expressed about the lattice origin and narrowed.
// ------------------------------------------------------------------------------------------

struct alignas(16) F4 {
    float a, b, c, d;
};
struct alignas(16) U4 {
    uint32_t a, b, c, d;
};

// kOptics: also (re)write the cells' OptRecords — of every cell, whatever the row band (they depend on the scalars
// and the alpha limit only; the host asks for them when one of those changed, c_api.hip)
template <bool kOptics>
__global__ __launch_bounds__(256) void build_records_mixed(GridView g, ImageParams im, const double* __restrict__ Xtab,
                                                           const double* __restrict__ Ytab, double alpha_limit, int order,
                                                           double steep_ratio) {
    const int64_t cell = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (cell >= g.n_cells) return;
    if (kOptics) {
        const CellOptics o = cell_optics(g, alpha_limit, order, cell);
        OptRecord q;
        q.alpha_raw = static_cast<float>(o.alpha_raw);
        q.alpha_c = static_cast<float>(o.alpha_c);
        q.source = (o.alpha_c != 0.0) ? static_cast<float>(o.q / o.alpha_c) : 0.0f;
        q.pad = 0.0f;
        *reinterpret_cast<F4*>(g.opt32 + cell) = *reinterpret_cast<const F4*>(&q);
    }
    CellRecord r;
    CellOptics o_unused;
    double v[4][3];
    if (!build_cell_impl<false>(g, alpha_limit, order, cell, r, o_unused, v)) return;  // outside this context's row band

    // origin: the pixel nearest the centre of the cell's footprint, clamped to the image (a cell outside the
    // domain is never walked)
    const double cx = 0.25 * (v[0][0] + v[1][0] + v[2][0] + v[3][0]), cy = 0.25 * (v[0][1] + v[1][1] + v[2][1] + v[3][1]);
    const double fc = rint((cx - im.x_min) / im.step_x), fr = rint((cy - im.y_min) / im.step_y);
    const int col0 = static_cast<int>(fmin(fmax(fc, 0.0), im.res_x - 1.0));
    const int row0 = static_cast<int>(fmin(fmax(fr, 0.0), im.res_y - 1.0));
    const double Xc = Xtab[col0], Yr = Ytab[row0];
    const double ox = Xc - r.x0, oy = Yr - r.y0;
    double c0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c0[k] = fma(r.plane[k][1], ox, fma(r.plane[k][2], oy, r.plane[k][0]));  // depth of plane k at the origin
    // depth origin: vertex 0 (NOT a plane's depth at the lattice origin: a steep face is far from the cell there)
    const float z0f = static_cast<float>(v[0][2]);
    GeoRecord out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out.plane[k][0] = static_cast<float>(c0[k] - static_cast<double>(z0f));  // +-inf stays +-inf
        out.plane[k][1] = static_cast<float>(r.plane[k][1] * im.step_x);
        out.plane[k][2] = static_cast<float>(r.plane[k][2] * im.step_y);
    }
    // Is single precision enough for this cell?  fp32 evaluates plane k at a pixel with an absolute error of
    // about 1e-7 (|c'| + |gx' dcol| + |gy' drow|): harmless while those terms are of the size of the cell, but a
    // face that is steep against the rays makes them large and cancelling.  Bound them over the cell's own
    // footprint (they are largest at one of its vertices) and compare with the cell's extent along the rays:
    // beyond steep_ratio (option "steep_ratio") the cell is marked and the walk evaluates it from its fp64 record instead.
    const double z_lo = fmin(fmin(v[0][2], v[1][2]), fmin(v[2][2], v[3][2]));
    const double z_hi = fmax(fmax(v[0][2], v[1][2]), fmax(v[2][2], v[3][2]));
    uint32_t steep_slots = 0;  // bit k: face slot k is evaluated in fp64 by the walk
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!(fabs(c0[k]) <= DBL_MAX)) continue;  // edge-on / flat slot: +-inf, exact
        const double cabs = fabs(c0[k] - static_cast<double>(z0f));
        double worst = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // only the face's own vertices bound its footprint: the fourth vertex is where the plane is
            // extrapolated furthest, and no ray inside the cell sees the plane there.  (A vertex counts as the
            // face's own if the plane passes through it; a sliver's fourth vertex may pass too: merely cautious.)
            const double at_v = fma(r.plane[k][1], v[j][0] - r.x0, fma(r.plane[k][2], v[j][1] - r.y0, r.plane[k][0]));
            if (fabs(at_v - v[j][2]) > 1e-6 * (z_hi - z_lo)) continue;
            worst = fmax(worst, cabs + fabs(r.plane[k][1] * (v[j][0] - Xc)) + fabs(r.plane[k][2] * (v[j][1] - Yr)));
        }
        if (steep_ratio > 0.0 && !(worst <= steep_ratio * (z_hi - z_lo))) steep_slots |= 1u << k;  // (NaN -> steep)
    }
    const bool steep = steep_slots != 0u;
    if (steep) {
        // the same four planes in double precision, about the same origin and depth origin as the fp32 ones:
        // SteepPlanes in the cell's (otherwise unused) CellRecord slot
        SteepPlanes sp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sp.p[k][0] = c0[k] - static_cast<double>(z0f);
            sp.p[k][1] = r.plane[k][1] * im.step_x;
            sp.p[k][2] = r.plane[k][2] * im.step_y;
        }
        sp.pad[0] = sp.pad[1] = sp.pad[2] = sp.pad[3] = 0.0;
        const U4* rs = reinterpret_cast<const U4*>(&sp);
        U4* rd = reinterpret_cast<U4*>(g.rec + cell);
#pragma unroll
        for (int k = 0; k < 8; ++k) rd[k] = rs[k];
    }
    const uint32_t n_up = r.nbr[0] >> kUpperCountShift;
    const int first = (order == 0) ? 0 : 1;  // exit candidates: slots 0..2 walking up, 1..3 walking down
    out.w[0] = (r.nbr[first] & kIdMask) | (n_up << kUpperCountShift) | (steep ? kExactBit : 0u);
    out.w[1] = (r.nbr[first + 1] & kIdMask) | (steep_slots << kSteepSlotShift);
    out.w[2] = r.nbr[first + 2] & kIdMask;
    out.w[3] = static_cast<uint32_t>(col0) | (static_cast<uint32_t>(row0) << 16);

    const U4* src = reinterpret_cast<const U4*>(&out);
    U4* dst = reinterpret_cast<U4*>(g.geo + cell);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = src[k];
    g.z0[cell] = z0f;
}

void launch_build_records_mixed(hipStream_t s, const GridView& g, const ImageParams& im, const double* Xtab,
                                const double* Ytab, double alpha_limit, int order, double steep_ratio, bool with_optics) {
    if (g.n_cells <= 0) return;
    const unsigned blocks = static_cast<unsigned>((g.n_cells + 255) / 256);
    if (with_optics)
        hipLaunchKernelGGL(build_records_mixed<true>, dim3(blocks), dim3(256), 0, s, g, im, Xtab, Ytab, alpha_limit, order, steep_ratio);
    else
        hipLaunchKernelGGL(build_records_mixed<false>, dim3(blocks), dim3(256), 0, s, g, im, Xtab, Ytab, alpha_limit, order, steep_ratio);
}

// ------------------------------------------------------------------------------------------
// walk_composite_mixed
// ------------------------------------------------------------------------------------------
template <int TILE, int ORDER>
__global__ __launch_bounds__(256, 8) __attribute__((amdgpu_num_sgpr(80))) void walk_composite_mixed(WalkParams P) {
    using TS = TileShape<TILE>;
    constexpr int TW = TS::WW * TS::GX, TH = TS::WH * TS::GY;
    constexpr bool kUp = (ORDER == 0);
    constexpr int kWaves = TS::GX * TS::GY;  // wavefronts per workgroup
    __shared__ V4F s_stage[kWaves][kMixSlots * kMixStride];
    __shared__ int s_elect[kWaves][kMixBuckets + 128];  // leader tables of 256 and 64 buckets + the cell id of every slot
    __shared__ double s_scur[kWaves][64];

    const ImageParams& im = P.im;
    const int tiles_x = (im.res_x + TW - 1) / TW;
    const int tiles_y = (im.n_local_rows + TH - 1) / TH;
    int tx, ty;
    if (P.xcd_mode == 0) {
        ty = blockIdx.x / tiles_x;
        tx = blockIdx.x - ty * tiles_x;
    } else {
        // square super-blocks of S x S workgroups dealt round-robin to the 8 XCDs (walk_composite_lds)
        const int S = P.band_tiles;
        const int sbx_n = (tiles_x + S - 1) / S, sby_n = (tiles_y + S - 1) / S;
        const int xcd = blockIdx.x & 7;
        const int seq = blockIdx.x >> 3;
        const int sb = (seq / (S * S)) * 8 + xcd;
        const int within = seq - (seq / (S * S)) * (S * S);
        if (sb >= sbx_n * sby_n) return;
        int sby = sb / sbx_n;
        const int sbx = sb - sby * sbx_n;
        if (P.n_sb_rows == sby_n) sby = static_cast<int>(P.sb_order[sby]);  // dearest rows first (walk_kernels.hip)
        ty = sby * S + within / S;
        tx = sbx * S + (within - (within / S) * S);
        if (tx >= tiles_x || ty >= tiles_y) return;
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = tx * TW + (wave % TS::GX) * TS::WW + (lane % TS::WW);
    const int lrow = ty * TH + (wave / TS::GX) * TS::WH + (lane / TS::WW);
    auto pixel_index = [&]() { return static_cast<size_t>(lrow) * im.res_x + col; };
    const bool in_image = (col < im.res_x) && (lrow < im.n_local_rows);
    V4F* const my_stage = s_stage[wave];
    int* const my_elect = s_elect[wave];
    double* const my_scur = s_scur[wave];
    const char* const geo_bytes = reinterpret_cast<const char*>(P.geo);
    const char* const opt_bytes = reinterpret_cast<const char*>(P.opt32);

    constexpr unsigned kOverflowBit = 0x80000000u;
    unsigned n_seg = 0;
    unsigned n_step_wave = 0;
    double tau = 0.0, I = 0.0, T = 1.0;
    int grow = 0;  // global image row of this lane's pixel
    int nb = -1;

    {
        size_t lp = 0;
        uint32_t mv = 0;
        EntryHead ent{0, 0};
        if (in_image) {
            lp = pixel_index();
            mv = P.mask ? P.mask[lp] : 0u;
            ent = load_entry_head(P.entry_head + lp);
        }
        if (__builtin_amdgcn_ballot_w64(mv != 0u || ent.count != 0) == 0ull) {  // neither grid nor solid: zeros, done
            if (in_image) {
                __builtin_nontemporal_store(0.f, &P.out[lp].x);
                __builtin_nontemporal_store(0.f, &P.out[lp].y);
            }
            return;
        }
        if (in_image && !mv) {
            grow = global_row_of(im, lrow);
            double w_cur = -DBL_MAX, w_entry = 0.0;  // (this walk evaluates both faces of a cell itself: the entry's depth is not used)
            if (ent.count > 0) nb = next_entry<kUp>(P, lp, ent, w_cur, w_entry);
            my_scur[lane] = w_cur;
        }
    }
    my_elect[kMixBuckets + 64 + lane] = 0;  // slot ids: always a valid cell id

    // the contribution of the step just taken waits here while the next records are in flight
    bool pend = false;
    float pend_dz = 0.0f;
    V4F pend_opt = {0.0f, 0.0f, 0.0f, 0.0f};  // {alpha_raw, alpha_c, source, -}

    // Staging by LDS-DMA (global_load_lds_dwordx4: destination = wave-uniform base + 16 * lane; no vector register in
    // between, no ds_write_b128): a pass of 60 lanes fills twelve whole slots of five 16-byte units, so a lane
    // fetches the same piece of the same slot-within-the-pass in both passes.  Three registers: where that slot's
    // cell id is posted (beyond every limit for the four idle lanes), log2 of the bytes per cell of the array the
    // piece comes from (pieces 0-3: GeoRecord, 4: OptRecord) and the piece's offset from P.geo.
    constexpr int kDmaSlots = 64 / kMixStride;
    constexpr int kDmaPasses = (kMixSlots + kDmaSlots - 1) / kDmaSlots;
    using LdsInts = const __attribute__((address_space(3))) int*;
    uint32_t dma_id_at, dma_pitch, dma_off;
    {
        const uint32_t ids_at = (uint32_t)(uintptr_t)(LdsInts)(my_elect + kMixBuckets + 64);  // LDS byte address
        const int s_ = lane / kMixStride, pc = lane - s_ * kMixStride;
        dma_id_at = s_ < kDmaSlots ? ids_at + 4u * static_cast<uint32_t>(s_) : 0xFFFFF000u;
        dma_pitch = pc < 4 ? 6u : 4u;
        dma_off = pc < 4 ? 16u * static_cast<uint32_t>(pc) : static_cast<uint32_t>(opt_bytes - geo_bytes);
    }

    for (unsigned iter = 0;; ++iter) {
        const bool need = nb >= 0;
        const unsigned long long needs = __builtin_amdgcn_ballot_w64(need);
        if (needs == 0ull) break;
        if (iter >= P.max_steps) {
            if (need) n_seg |= kOverflowBit;
            break;
        }
        n_step_wave += static_cast<unsigned>(__popcll(needs));

        // 1. one slot per DISTINCT cell: leader election through a hashed table in LDS (walk_composite_lds)
        const unsigned unb = static_cast<unsigned>(nb);
        const unsigned h1 = (unb ^ (unb >> 8)) & (kMixBuckets - 1u);
        const int ticket = static_cast<int>((unb << 6) | static_cast<unsigned>(lane));  // ids of this kernel have 26 bits
        if (need) my_elect[h1] = ticket;
        __builtin_amdgcn_wave_barrier();
        const int won = my_elect[h1];
        __builtin_amdgcn_wave_barrier();
        int w = won & 63;
        const bool other = (static_cast<unsigned>(won) >> 6) != unb;
        const bool open = need && other;
        if ((__builtin_amdgcn_ballot_w64(other) & needs) != 0ull) {
            const unsigned t = unb >> 6;
            const unsigned h2 = (unb + t + (t << 2) + (unb >> 12)) & 63u;
            if (open) my_elect[kMixBuckets + h2] = ticket;
            __builtin_amdgcn_wave_barrier();
            const int won2 = my_elect[kMixBuckets + h2];
            __builtin_amdgcn_wave_barrier();
            if (open) w = ((static_cast<unsigned>(won2) >> 6) == unb) ? (won2 & 63) : lane;
        }
        const unsigned long long heads =
            __builtin_amdgcn_uicmp(static_cast<unsigned>(w), static_cast<unsigned>(lane), 32 /* eq */) & needs;
        const int n_runs = __builtin_popcountll(heads);
        const int rank = static_cast<int>(__builtin_amdgcn_mbcnt_hi(
            static_cast<uint32_t>(heads >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(heads), 0u)));
        if (need && w == lane) my_elect[kMixBuckets + 64 + rank] = nb;
        const int slot = __builtin_amdgcn_ds_bpermute(w << 2, rank);
        __builtin_amdgcn_wave_barrier();
        const int n_staged = __builtin_amdgcn_readfirstlane(n_runs < kMixSlots ? n_runs : kMixSlots);

        // 2. cooperative loads, straight into the slots: pass j stages slots 12 j ... 12 j + 11 (LDS units from 60 j)
        {
            const uint32_t ids_end = (uint32_t)(uintptr_t)(LdsInts)(my_elect + kMixBuckets + 64) + 4u * static_cast<uint32_t>(n_staged);
#pragma unroll
            for (int j = 0; j < kDmaPasses; ++j) {
                if (j == 0 || kDmaSlots * j < n_staged) {  // wave-uniform
                    if (dma_id_at < ids_end - 4u * kDmaSlots * j) {  // this lane's slot 12 j + s is staged
                        const uint32_t id_ = static_cast<uint32_t>(((LdsInts)(uintptr_t)dma_id_at)[kDmaSlots * j]);
                        const uint32_t off = (id_ << dma_pitch) + dma_off;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(geo_bytes + off),
                                                         (__attribute__((address_space(3))) void*)(my_stage + kDmaSlots * kMixStride * j), 16, 0, 0);
                    }
                }
            }
        }

        // ... while they are in flight: emission / absorption of the step just taken.
        //     I' = I + (I - S)(e^{-alpha dz} - 1)  ==  (Q - (Q - alpha I) e^{-alpha dz}) / alpha   (line.cpp:220-224)
        {
            const float xarg = -(pend_opt.y * pend_dz);
            const bool big_arg = pend && !(xarg > -0.125f);
            if (__builtin_amdgcn_ballot_w64(big_arg) == 0ull) {
                if (pend) {
                    const double em1 = static_cast<double>(expm1_small(xarg));
                    const double S = static_cast<double>(pend_opt.z);
                    if (ORDER == 0) {
                        I = fma(I - S, em1, I);
                    } else if (T >= P.t_cutoff) {
                        I = fma(-(T * S), em1, I);  // + T S (1 - e)
                        T = fma(T, em1, T);         // T e
                    }
                }
            } else if (pend) {
                const double e = exp_nonpositive_local(static_cast<double>(xarg));
                const double S = static_cast<double>(pend_opt.z);
                if (ORDER == 0) {
                    if (pend_opt.y != 0.0f) I = fma(I - S, e - 1.0, I);
                } else if (T >= P.t_cutoff) {
                    I = fma(T * S, 1.0 - e, I);
                    T *= e;
                }
            }
            pend = false;
        }

        // 3. the pieces have landed (nothing orders a ds_read behind this wavefront's own LDS-DMA but its vmcnt)
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();

        // 4. every ray fetches its cell
        if (need) {
            // Five LDS reads, issued together and kept LDS reads: left to itself hipcc merges the staged and
            // the direct path into flat loads through a generic pointer and defers the optics read into the
            // "contributes" branch, one more round trip on the critical path.
            const V4F* r = reinterpret_cast<const V4F*>(reinterpret_cast<const char*>(my_stage) +
                                                        __umul24(static_cast<unsigned>(slot) & (kMixSlots - 1u), kMixStride * 16u));
            // (the optics go straight into the pending registers: free here, the previous step's emission is done,
            // and only looked at again if this step contributes)
            V4F g0 = r[0], g1 = r[1], g2 = r[2], gwf = r[3];
            pend_opt = r[4];
            asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(gwf), "+v"(pend_opt));
            if (slot >= kMixSlots) {  // more distinct cells than slots: rare in 8x8 tiles
                const V4F* gr = reinterpret_cast<const V4F*>(P.geo + nb);
                g0 = gr[0];
                g1 = gr[1];
                g2 = gr[2];
                gwf = gr[3];
                pend_opt = *reinterpret_cast<const V4F*>(P.opt32 + nb);
            }
            V4U gw;
            gw.x = __float_as_uint(gwf.x);
            gw.y = __float_as_uint(gwf.y);
            gw.z = __float_as_uint(gwf.z);
            gw.w = __float_as_uint(gwf.w);
            // planes (c, gx, gy): g0.xyz | g0.w g1.xy | g1.zw g2.x | g2.yzw
            const float dcol = static_cast<float>(col - static_cast<int>(gw.w & 0xFFFFu));
            const float drow = static_cast<float>(grow - static_cast<int>(gw.w >> 16));
            const float z0f_ = fmaf(g0.y, dcol, fmaf(g0.z, drow, g0.x));
            const float z1f_ = fmaf(g1.x, dcol, fmaf(g1.y, drow, g0.w));
            const float z2f_ = fmaf(g1.w, dcol, fmaf(g2.x, drow, g1.z));
            const float z3f_ = fmaf(g2.z, dcol, fmaf(g2.w, drow, g2.y));
            // Faces steep against the rays (build_records_mixed; ~2 % of the cells have one): fp32 loses such a face's
            // depth in the cancellation of its large terms, so the face is evaluated from double-precision
            // coefficients — about the same lattice origin and depth origin, so the result simply replaces the fp32
            // depth of that slot and everything below runs as for any other cell.  Per distinct such cell of the
            // wavefront (usually one) and steep slot, the three coefficients arrive by ONE scalar load (eight SGPRs,
            // no vector registers) and the lanes inside that cell do two fp64 FMAs.
            float z0 = z0f_, z1 = z1f_, z2 = z2f_, z3 = z3f_;
            {
                const uint32_t steep_bit = gw.x & kExactBit;
                unsigned long long todo = __builtin_amdgcn_uicmp(steep_bit, 0u, 33 /* ne */);
                if (todo != 0ull) {
                    const double dcol_d = static_cast<double>(dcol), drow_d = static_cast<double>(drow);
                    while (todo != 0ull) {
                        const int first_lane = __builtin_ctzll(todo);
                        const int id = __builtin_amdgcn_readlane(nb, first_lane);
                        const uint32_t slots = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(gw.y), first_lane)) >> kSteepSlotShift;
                        const bool mine = steep_bit != 0u && nb == id;
                        todo &= ~__builtin_amdgcn_ballot_w64(mine);
                        const char* planes = reinterpret_cast<const char*>(P.rec + id);  // uniform address: SteepPlanes
                        auto dbl = [](int lo, int hi) { return __hiloint2double(hi, lo); };
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if ((slots >> k) & 1u) {  // wave-uniform
                                SRec8 h;
                                asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(h) : "s"(planes), "n"(24 * k) : "memory");
                                const float zk = static_cast<float>(fma(dbl(h[2], h[3]), dcol_d, fma(dbl(h[4], h[5]), drow_d, dbl(h[0], h[1]))));
                                if (mine) {
                                    if (k == 0) z0 = zk;
                                    if (k == 1) z1 = zk;
                                    if (k == 2) z2 = zk;
                                    if (k == 3) z3 = zk;
                                }
                            }
                        }
                    }
                }
            }
            const uint32_t n_up = gw.x >> kUpperCountShift;  // 1..3: slot 0 always upper, slot 3 always lower
            const bool up1 = n_up > 1u, up2 = n_up > 2u;
            const float u1 = up1 ? z1 : INFINITY, l1 = up1 ? -INFINITY : z1;
            const float u2 = up2 ? z2 : INFINITY, l2 = up2 ? -INFINITY : z2;
            const float z_top = min3_f32(z0, u1, u2);
            const float z_bot = max3_f32(z3, l1, l2);
            const float dz = z_top - z_bot;  // chord through the cell (line.cpp:124-131)
            uint32_t w_out;                  // neighbour word of the exit face
            float z_exit;
            if (kUp) {  // leaves through the lowest upper face: ids of slots 0, 1, 2
                w_out = (z0 == z_top) ? gw.x : (u1 == z_top) ? gw.y : gw.z;
                z_exit = z_top;
            } else {    // through the highest lower face: ids of slots 1, 2, 3
                w_out = (z3 == z_bot) ? gw.z : (l2 == z_bot) ? gw.y : gw.x;
                z_exit = z_bot;
            }
            const bool has_exit = fabsf(z_exit) < INFINITY;
            const double dz_tau = static_cast<double>(dz);
            if (dz > 0.0f && dz < INFINITY) {
                ++n_seg;
                tau = fma(dz_tau, static_cast<double>(pend_opt.x), tau);  // line.cpp:189 (unclamped alpha)
                pend = true;
                pend_dz = dz;
            }
            const uint32_t id = w_out & kIdMask;
            int nxt = static_cast<int>(id);
            if (id == kNoCell) {  // left the grid: re-entry of a non-convex grid?
                const size_t lp = pixel_index();
                double w_cur = my_scur[lane], w_entry = 0.0;
                if (has_exit) {
                    const double z_abs = static_cast<double>(P.z0[nb]) + static_cast<double>(z_exit);
                    w_cur = fmax(w_cur, kUp ? z_abs : -z_abs);  // walk coordinate: grows along the walk
                }
                nxt = next_entry<kUp>(P, lp, load_entry_head(P.entry_head + lp), w_cur, w_entry);
                my_scur[lane] = w_cur;
            }
            nb = nxt;
        }
    }
    if (pend) {  // the last step's contribution
        const float xarg = -(pend_opt.y * pend_dz);
        const double em1 = (xarg > -0.125f) ? static_cast<double>(expm1_small(xarg)) : exp_nonpositive_local(static_cast<double>(xarg)) - 1.0;
        const double S = static_cast<double>(pend_opt.z);
        if (ORDER == 0) {
            I = fma(I - S, em1, I);
        } else if (T >= P.t_cutoff) {
            I = fma(-(T * S), em1, I);
            T = fma(T, em1, T);
        }
    }

    const unsigned overflow = n_seg >> 31;
    n_seg &= ~kOverflowBit;
    unsigned is_solid = 0, n_entries = 0;
    if (in_image) {
        const size_t lp = pixel_index();
        float2 result = make_float2(static_cast<float>(tau), static_cast<float>(I));  // plane.cpp:165-166
        const uint32_t mv = P.mask ? P.mask[lp] : 0u;
        if (mv) {
            double colour = 0.0;
            for (int s = 0; s < P.solids.n_slots; ++s)
                if (mv == static_cast<uint32_t>(s) + 1u) colour = P.solids.colour[s];
            result.x = static_cast<float>(colour);
            result.y = result.x;
            is_solid = 1;
        }
        n_entries = static_cast<unsigned>(load_entry_head(P.entry_head + lp).count);
        if (n_entries && !P.keep_entries) __builtin_nontemporal_store(0ll, reinterpret_cast<long long*>(P.entry_head + lp));
        __builtin_nontemporal_store(result.x, &P.out[lp].x);
        __builtin_nontemporal_store(result.y, &P.out[lp].y);
    }

    if (P.row_cost) {
        unsigned rs = n_seg;
#pragma unroll
        for (int d = TS::WW / 2; d >= 1; d >>= 1) rs += __shfl_xor(rs, d);
        if ((lane % TS::WW) == 0 && rs && lrow < im.n_local_rows) atomicAdd(P.row_cost + lrow, rs);
    }
    const unsigned s_seg = wave_sum_u32(n_seg);
    const unsigned s_cov = wave_sum_u32(n_seg > 0 ? 1u : 0u);
    const unsigned s_sol = wave_sum_u32(is_solid);
    const unsigned s_ovf = wave_sum_u32(overflow);
    const unsigned s_ent = wave_sum_u32(n_entries);
    if (lane == 0) {
        FrameCounters* const fc = P.counters + ((blockIdx.x * 4u + static_cast<unsigned>(wave)) % kCounterShards);
        if (s_seg && P.sb_cost && P.xcd_mode != 0) {  // what this wavefront cost, to its row of super-blocks
            const int sb_row = ty / P.band_tiles;
            if (sb_row < kMaxSbRows) atomicAdd(P.sb_cost + sb_row, s_seg);
        }
        if (s_ent) atomicAdd(&fc->entries, static_cast<unsigned long long>(s_ent));
        if (s_seg) atomicAdd(&fc->segments, static_cast<unsigned long long>(s_seg));
        if (n_step_wave) atomicAdd(&fc->steps, static_cast<unsigned long long>(n_step_wave));
        if (s_cov) atomicAdd(&fc->covered, static_cast<unsigned long long>(s_cov));
        if (s_sol) atomicAdd(&fc->solid_pixels, static_cast<unsigned long long>(s_sol));
        if (s_ovf) {
            atomicAdd(&fc->walk_overflow, s_ovf);
            atomicAdd(P.sticky + 1, s_ovf);
        }
    }
}

template <int TILE, int ORDER>
static void launch_mixed_t(hipStream_t s, const WalkParams& p) {
    using TS = TileShape<TILE>;
    constexpr int TW = TS::WW * TS::GX, TH = TS::WH * TS::GY;
    const int tiles_x = (p.im.res_x + TW - 1) / TW;
    const int tiles_y = (p.im.n_local_rows + TH - 1) / TH;
    if (tiles_x <= 0 || tiles_y <= 0) return;
    WalkParams q = p;
    long long blocks;
    if (p.xcd_mode == 0) {
        blocks = static_cast<long long>(tiles_x) * tiles_y;
    } else {
        const int sb_rows = p.band_rows > 0 ? p.band_rows : 32;
        const int S = sb_rows / TH > 0 ? sb_rows / TH : 1;
        const long long n_sb = static_cast<long long>((tiles_x + S - 1) / S) * ((tiles_y + S - 1) / S);
        blocks = 8ll * ((n_sb + 7) / 8) * S * S;
        q.band_tiles = S;
    }
    hipLaunchKernelGGL((walk_composite_mixed<TILE, ORDER>), dim3(static_cast<unsigned>(blocks)), dim3(64u * TS::GX * TS::GY),
                       static_cast<size_t>(p.lds_pad), s, q);
}

void launch_walk_mixed(hipStream_t s, const WalkParams& p, int tile_shape) {
    if (p.order == 0) {
        switch (tile_shape) {
            case 1: launch_mixed_t<1, 0>(s, p); break;
            case 2: launch_mixed_t<2, 0>(s, p); break;
            case 3: launch_mixed_t<3, 0>(s, p); break;
            default: launch_mixed_t<0, 0>(s, p); break;
        }
    } else {
        switch (tile_shape) {
            case 1: launch_mixed_t<1, 1>(s, p); break;
            case 2: launch_mixed_t<2, 1>(s, p); break;
            case 3: launch_mixed_t<3, 1>(s, p); break;
            default: launch_mixed_t<0, 1>(s, p); break;
        }
    }
}

}  // namespace c5
