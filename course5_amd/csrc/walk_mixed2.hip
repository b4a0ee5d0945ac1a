// walk_composite_mixed2: the mixed-precision walk (walk_mixed.hip) with TWO rays per lane.
//
// A step of the walk is a dependent chain (election -> staging load -> LDS -> geometry -> exit) of a few
// thousand cycles, and a SIMD holds at most eight wavefronts: the one-ray kernel is bound by that chain.  Here a
// wavefront owns an 8 x 16 pixel tile and every lane walks two rays, eight rows apart, in the same iteration:
// two independent chains per lane interleave, and the per-step overhead — leader election, staging loads, loop
// control — is paid once for 128 rays.  The price is registers (per-ray state twice): fewer wavefronts per SIMD,
// each carrying twice the rays.  Same records, same arithmetic per ray, same results as walk_composite_mixed.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "device_types.hpp"
#include "kernels.hpp"
#include "walk_common.hpp"
#include "walk_mixed_common.hpp"

namespace c5 {

constexpr int kSlots2 = 32;  // staged cells per wavefront and step: two load instructions of sixteen
// per-wavefront LDS words: leader tables of 256 and 64 buckets, cell id of every slot, slot of every leader code
constexpr int kElect2 = 256 + 64 + kSlots2 + 128;

struct Ray2 {
    int nb;            // next cell, -1 when the ray is finished
    int grow;          // global image row
    unsigned n_seg;
    double tau, I, T;
    float pend_x;      // -alpha_c dz of the step whose emission is still to be applied
    float pend_S;      // its source function Q / alpha_c
    bool pend;
};

template <int ORDER>
__device__ __forceinline__ void emit_pending(Ray2& R, bool short_series, double t_cutoff) {
    if (!R.pend) return;
    const double em1 = short_series ? static_cast<double>(expm1_small(R.pend_x))
                                    : exp_nonpositive_local(static_cast<double>(R.pend_x)) - 1.0;
    const double S = static_cast<double>(R.pend_S);
    if (ORDER == 0) {
        R.I = fma(R.I - S, em1, R.I);  // (Q - (Q - alpha I) e^{-alpha dz}) / alpha, line.cpp:220-224
    } else if (R.T >= t_cutoff) {
        R.I = fma(-(R.T * S), em1, R.I);
        R.T = fma(R.T, em1, R.T);
    }
    R.pend = false;
}

template <int ORDER>
__global__ __launch_bounds__(256, 5) __attribute__((amdgpu_num_sgpr(96))) void walk_composite_mixed2(WalkParams P) {
    constexpr bool kUp = (ORDER == 0);
    constexpr int TW = 16, TH = 32;  // workgroup: 2 x 2 wavefronts of 8 x 16 pixels
    __shared__ V4F s_stage[4][kSlots2 * kMixStride];
    __shared__ int s_elect[4][kElect2];
    __shared__ double s_scur[4][128];

    const ImageParams& im = P.im;
    const int tiles_x = (im.res_x + TW - 1) / TW;
    const int tiles_y = (im.n_local_rows + TH - 1) / TH;
    int tx, ty;
    if (P.xcd_mode == 0) {
        ty = blockIdx.x / tiles_x;
        tx = blockIdx.x - ty * tiles_x;
    } else {
        const int S = P.band_tiles;
        const int sbx_n = (tiles_x + S - 1) / S, sby_n = (tiles_y + S - 1) / S;
        const int xcd = blockIdx.x & 7;
        const int seq = blockIdx.x >> 3;
        const int sb = (seq / (S * S)) * 8 + xcd;
        const int within = seq - (seq / (S * S)) * (S * S);
        if (sb >= sbx_n * sby_n) return;
        const int sby = sb / sbx_n, sbx = sb - sby * sbx_n;
        ty = sby * S + within / S;
        tx = sbx * S + (within - (within / S) * S);
        if (tx >= tiles_x || ty >= tiles_y) return;
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = tx * TW + (wave & 1) * 8 + (lane & 7);
    const int lrow0 = ty * TH + (wave >> 1) * 16 + (lane >> 3);  // ray 0; ray 1 is eight rows further
    V4F* const my_stage = s_stage[wave];
    int* const my_elect = s_elect[wave];
    int* const slot_id = my_elect + 320;
    int* const rank_tab = my_elect + 320 + kSlots2;
    double* const my_scur = s_scur[wave];
    const char* const geo_bytes = reinterpret_cast<const char*>(P.geo);
    const char* const opt_bytes = reinterpret_cast<const char*>(P.opt32);

    constexpr unsigned kOverflowBit = 0x80000000u;
    Ray2 ray[2];
    unsigned n_step_wave = 0;
    bool any = false;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        Ray2& R = ray[r];
        R.nb = -1;
        R.grow = 0;
        R.n_seg = 0;
        R.tau = 0.0;
        R.I = 0.0;
        R.T = 1.0;
        R.pend_x = 0.0f;
        R.pend_S = 0.0f;
        R.pend = false;
        const int lrow = lrow0 + 8 * r;
        if (col < im.res_x && lrow < im.n_local_rows) {
            const size_t lp = static_cast<size_t>(lrow) * im.res_x + col;
            const uint32_t mv = P.mask ? P.mask[lp] : 0u;
            const EntryHead ent = load_entry_head(P.entry_head + lp);
            any |= mv != 0u || ent.count != 0;
            if (!mv) {
                R.grow = global_row_of(im, lrow);
                double s_cur = DBL_MAX;
                if (ent.count > 0) R.nb = next_entry<kUp>(P, lp, ent, s_cur);
                my_scur[64 * r + lane] = s_cur;
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0ull) {  // neither grid nor solid on the whole tile: zeros, done
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int lrow = lrow0 + 8 * r;
            if (col < im.res_x && lrow < im.n_local_rows) {
                const size_t lp = static_cast<size_t>(lrow) * im.res_x + col;
                __builtin_nontemporal_store(0.f, &P.out[lp].x);
                __builtin_nontemporal_store(0.f, &P.out[lp].y);
            }
        }
        return;
    }
    if (lane < kSlots2) slot_id[lane] = 0;  // slot ids: always a valid cell id

    const uint32_t geo_piece_off = static_cast<uint32_t>(lane & 3) * 16u;
    const int gslot = lane >> 2;

    for (unsigned iter = 0;; ++iter) {
        const bool need0 = ray[0].nb >= 0, need1 = ray[1].nb >= 0;
        const unsigned long long needs0 = __builtin_amdgcn_ballot_w64(need0), needs1 = __builtin_amdgcn_ballot_w64(need1);
        if ((needs0 | needs1) == 0ull) break;
        if (iter >= P.max_steps) {
            if (need0) ray[0].n_seg |= kOverflowBit;
            if (need1) ray[1].n_seg |= kOverflowBit;
            break;
        }
        n_step_wave += static_cast<unsigned>(__popcll(needs0) + __popcll(needs1));

        // 1. one slot per DISTINCT cell among the 128 rays: leader election through a hashed LDS table.  A ticket is
        //    (cell id << 7 | ray << 6 | lane); ids of this kernel have at most 25 bits.
        const unsigned u0 = static_cast<unsigned>(ray[0].nb), u1 = static_cast<unsigned>(ray[1].nb);
        const int code0 = lane, code1 = 64 | lane;
        const unsigned h0 = (u0 ^ (u0 >> 8)) & 255u, h1 = (u1 ^ (u1 >> 8)) & 255u;
        const int t0 = static_cast<int>((u0 << 7) | static_cast<unsigned>(code0));
        const int t1 = static_cast<int>((u1 << 7) | static_cast<unsigned>(code1));
        if (need0) my_elect[h0] = t0;
        if (need1) my_elect[h1] = t1;
        __builtin_amdgcn_wave_barrier();
        const int won0 = my_elect[h0], won1 = my_elect[h1];
        __builtin_amdgcn_wave_barrier();
        int w0 = won0 & 127, w1 = won1 & 127;
        const bool open0 = need0 && (static_cast<unsigned>(won0) >> 7) != u0;
        const bool open1 = need1 && (static_cast<unsigned>(won1) >> 7) != u1;
        if (__builtin_amdgcn_ballot_w64(open0 || open1) != 0ull) {  // a bucket shared by two cells: second table
            const unsigned a0 = u0 >> 6, a1 = u1 >> 6;
            const unsigned g0 = (u0 + a0 + (a0 << 2) + (u0 >> 12)) & 63u, g1 = (u1 + a1 + (a1 << 2) + (u1 >> 12)) & 63u;
            if (open0) my_elect[256 + g0] = t0;
            if (open1) my_elect[256 + g1] = t1;
            __builtin_amdgcn_wave_barrier();
            const int x0 = my_elect[256 + g0], x1 = my_elect[256 + g1];
            __builtin_amdgcn_wave_barrier();
            if (open0) w0 = ((static_cast<unsigned>(x0) >> 7) == u0) ? (x0 & 127) : code0;
            if (open1) w1 = ((static_cast<unsigned>(x1) >> 7) == u1) ? (x1 & 127) : code1;
        }
        const bool head0 = need0 && w0 == code0, head1 = need1 && w1 == code1;
        const unsigned long long heads0 = __builtin_amdgcn_ballot_w64(head0), heads1 = __builtin_amdgcn_ballot_w64(head1);
        const int n_a = __builtin_popcountll(heads0);
        const int n_runs = n_a + __builtin_popcountll(heads1);
        const int rank0 = static_cast<int>(__builtin_amdgcn_mbcnt_hi(
            static_cast<uint32_t>(heads0 >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(heads0), 0u)));
        const int rank1 = n_a + static_cast<int>(__builtin_amdgcn_mbcnt_hi(
            static_cast<uint32_t>(heads1 >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(heads1), 0u)));
        if (head0) {
            rank_tab[code0] = rank0;
            if (rank0 < kSlots2) slot_id[rank0] = ray[0].nb;
        }
        if (head1) {
            rank_tab[code1] = rank1;
            if (rank1 < kSlots2) slot_id[rank1] = ray[1].nb;
        }
        __builtin_amdgcn_wave_barrier();
        const int slot0 = rank_tab[w0], slot1 = rank_tab[w1];  // (garbage for a finished ray: never used)
        const int n_staged = __builtin_amdgcn_readfirstlane(n_runs < kSlots2 ? n_runs : kSlots2);

        // 2. cooperative loads: sixteen GeoRecords per instruction (4 lanes x 16 B each), lanes 0-31 one OptRecord each
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
        V4F stage_g0, stage_g1, stage_o;
#pragma clang diagnostic pop
        {
            const uint32_t idg = static_cast<uint32_t>(slot_id[gslot]);
            stage_g0 = *reinterpret_cast<const V4F*>(geo_bytes + ((idg << 6) | geo_piece_off));
        }
        if (n_staged > 16) {
            const uint32_t idg = static_cast<uint32_t>(slot_id[16 + gslot]);
            stage_g1 = *reinterpret_cast<const V4F*>(geo_bytes + ((idg << 6) | geo_piece_off));
        }
        if (lane < kSlots2) {
            const uint32_t ido = static_cast<uint32_t>(slot_id[lane]);
            stage_o = *reinterpret_cast<const V4F*>(opt_bytes + (ido << 4));
        }

        // ... while they are in flight: emission / absorption of the steps just taken
        {
            const bool big = (ray[0].pend && !(ray[0].pend_x > -0.125f)) || (ray[1].pend && !(ray[1].pend_x > -0.125f));
            const bool short_series = __builtin_amdgcn_ballot_w64(big) == 0ull;
            if (short_series) {
                emit_pending<ORDER>(ray[0], true, P.t_cutoff);
                emit_pending<ORDER>(ray[1], true, P.t_cutoff);
            } else {
                emit_pending<ORDER>(ray[0], false, P.t_cutoff);
                emit_pending<ORDER>(ray[1], false, P.t_cutoff);
            }
        }

        // 3. park the pieces in LDS
        __builtin_amdgcn_wave_barrier();
        if (gslot < n_staged) my_stage[gslot * kMixStride + (lane & 3)] = stage_g0;
        if (16 + gslot < n_staged) my_stage[(16 + gslot) * kMixStride + (lane & 3)] = stage_g1;
        if (lane < n_staged) my_stage[lane * kMixStride + 4] = stage_o;  // (n_staged <= 32)
        __builtin_amdgcn_wave_barrier();

        // 4. every ray fetches its cell and takes its step
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            Ray2& R = ray[r];
            const int slot = r == 0 ? slot0 : slot1;
            if (R.nb < 0) continue;
            const V4F* rd = reinterpret_cast<const V4F*>(reinterpret_cast<const char*>(my_stage) +
                                                         __umul24(static_cast<unsigned>(slot) & (kSlots2 - 1u), kMixStride * 16u));
            V4F g0 = rd[0], g1 = rd[1], g2 = rd[2], gwf = rd[3], o = rd[4];
            asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(gwf), "+v"(o));  // five ds_read_b128, issued together
            if (slot >= kSlots2) {  // more distinct cells than slots
                const V4F* gr = reinterpret_cast<const V4F*>(P.geo + R.nb);
                g0 = gr[0];
                g1 = gr[1];
                g2 = gr[2];
                gwf = gr[3];
                o = *reinterpret_cast<const V4F*>(P.opt32 + R.nb);
            }
            V4U gw;
            gw.x = __float_as_uint(gwf.x);
            gw.y = __float_as_uint(gwf.y);
            gw.z = __float_as_uint(gwf.z);
            gw.w = __float_as_uint(gwf.w);
            const float dcol = static_cast<float>(col - static_cast<int>(gw.w & 0xFFFFu));
            const float drow = static_cast<float>(R.grow - static_cast<int>(gw.w >> 16));
            const float z0 = fmaf(g0.y, dcol, fmaf(g0.z, drow, g0.x));
            const float z1 = fmaf(g1.x, dcol, fmaf(g1.y, drow, g0.w));
            const float z2 = fmaf(g1.w, dcol, fmaf(g2.x, drow, g1.z));
            const float z3 = fmaf(g2.z, dcol, fmaf(g2.w, drow, g2.y));
            const uint32_t n_up = gw.x >> kUpperCountShift;
            const bool up1 = n_up > 1u, up2 = n_up > 2u;
            const float u1 = up1 ? z1 : INFINITY, l1 = up1 ? -INFINITY : z1;
            const float u2 = up2 ? z2 : INFINITY, l2 = up2 ? -INFINITY : z2;
            const float z_top = min3_f32(z0, u1, u2);
            const float z_bot = max3_f32(z3, l1, l2);
            float dz = z_top - z_bot;  // line.cpp:124-131
            uint32_t w_out;
            float z_exit;
            if (kUp) {
                w_out = (z0 == z_top) ? gw.x : (u1 == z_top) ? gw.y : gw.z;
                z_exit = z_top;
            } else {
                w_out = (z3 == z_bot) ? gw.z : (l2 == z_bot) ? gw.y : gw.x;
                z_exit = z_bot;
            }
            bool has_exit = fabsf(z_exit) < INFINITY;
            double dz_tau = static_cast<double>(dz);

            // cells with a face steep against the rays: fp64 record through scalar loads (walk_mixed.hip)
            const bool steep = (gw.x & kExactBit) != 0u;
            unsigned long long todo = __builtin_amdgcn_ballot_w64(steep);
            while (todo != 0ull) {
                const int first_lane = __builtin_ctzll(todo);
                const int id = __builtin_amdgcn_readlane(R.nb, first_lane);
                const bool mine = steep && R.nb == id;
                todo &= ~__builtin_amdgcn_ballot_w64(mine);
                const CellRecord* rec = P.rec + id;
                auto dbl = [](int a, int b) { return __hiloint2double(b, a); };
                double dx = 0.0, dy = 0.0, e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
                uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                if (mine) {
                    dx = P.Xtab[col];
                    dy = P.Ytab[R.grow];
                }
                {
                    SRec16 h;
                    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(h) : "s"(rec) : "memory");
                    if (mine) {
                        dx -= dbl(h[0], h[1]);
                        dy -= dbl(h[2], h[3]);
                        e0 = fma(dbl(h[6], h[7]), dx, fma(dbl(h[8], h[9]), dy, dbl(h[4], h[5])));
                        e1 = fma(dbl(h[12], h[13]), dx, fma(dbl(h[14], h[15]), dy, dbl(h[10], h[11])));
                    }
                }
                {
                    SRec16 h;
                    asm volatile("s_load_dwordx16 %0, %1, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(h) : "s"(rec) : "memory");
                    if (mine) {
                        e2 = fma(dbl(h[2], h[3]), dx, fma(dbl(h[4], h[5]), dy, dbl(h[0], h[1])));
                        e3 = fma(dbl(h[8], h[9]), dx, fma(dbl(h[10], h[11]), dy, dbl(h[6], h[7])));
                    }
                    q0 = static_cast<uint32_t>(h[12]);
                    q1 = static_cast<uint32_t>(h[13]);
                    q2 = static_cast<uint32_t>(h[14]);
                    q3 = static_cast<uint32_t>(h[15]);
                }
                if (mine) {
                    const uint32_t nu = q0 >> kUpperCountShift;
                    const double a1 = nu > 1u ? e1 : INFINITY, b1 = nu > 1u ? -INFINITY : e1;
                    const double a2 = nu > 2u ? e2 : INFINITY, b2 = nu > 2u ? -INFINITY : e2;
                    const double zt = fmin(e0, fmin(a1, a2)), zb = fmax(e3, fmax(b1, b2));
                    dz_tau = zt - zb;
                    dz = static_cast<float>(dz_tau);
                    double ze;
                    if (kUp) {
                        w_out = (e0 == zt) ? q0 : (a1 == zt) ? q1 : q2;
                        ze = zt;
                    } else {
                        w_out = (e3 == zb) ? q3 : (b2 == zb) ? q2 : q1;
                        ze = zb;
                    }
                    if ((w_out & kIdMask) == kNoCell && fabs(ze) < INFINITY)
                        my_scur[64 * r + lane] = fmin(my_scur[64 * r + lane], kUp ? -ze : ze);
                    has_exit = false;
                }
            }

            if (dz > 0.0f && dz < INFINITY) {
                ++R.n_seg;
                R.tau = fma(dz_tau, static_cast<double>(o.x), R.tau);  // line.cpp:189 (unclamped alpha)
                R.pend = true;
                R.pend_x = -(o.y * dz);
                R.pend_S = o.z;
            }
            const uint32_t id = w_out & kIdMask;
            int nxt = static_cast<int>(id);
            if (id == kNoCell) {  // left the grid: re-entry of a non-convex grid?
                const size_t lp = static_cast<size_t>(lrow0 + 8 * r) * im.res_x + col;
                double s_cur = my_scur[64 * r + lane];
                if (has_exit) {
                    const double z_abs = static_cast<double>(P.z0[R.nb]) + static_cast<double>(z_exit);
                    s_cur = fmin(s_cur, kUp ? -z_abs : z_abs);
                }
                nxt = next_entry<kUp>(P, lp, load_entry_head(P.entry_head + lp), s_cur);
                my_scur[64 * r + lane] = s_cur;
            }
            R.nb = nxt;
        }
    }
    {   // the last steps' contributions
        const bool big = (ray[0].pend && !(ray[0].pend_x > -0.125f)) || (ray[1].pend && !(ray[1].pend_x > -0.125f));
        if (__builtin_amdgcn_ballot_w64(big) == 0ull) {
            emit_pending<ORDER>(ray[0], true, P.t_cutoff);
            emit_pending<ORDER>(ray[1], true, P.t_cutoff);
        } else {
            emit_pending<ORDER>(ray[0], false, P.t_cutoff);
            emit_pending<ORDER>(ray[1], false, P.t_cutoff);
        }
    }

    unsigned t_seg = 0, t_cov = 0, t_sol = 0, t_ovf = 0, t_ent = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        Ray2& R = ray[r];
        const unsigned overflow = R.n_seg >> 31;
        R.n_seg &= ~kOverflowBit;
        unsigned is_solid = 0, n_entries = 0;
        const int lrow = lrow0 + 8 * r;
        if (col < im.res_x && lrow < im.n_local_rows) {
            const size_t lp = static_cast<size_t>(lrow) * im.res_x + col;
            float2 result = make_float2(static_cast<float>(R.tau), static_cast<float>(R.I));  // plane.cpp:165-166
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
            if (n_entries) __builtin_nontemporal_store(0ll, reinterpret_cast<long long*>(P.entry_head + lp));
            __builtin_nontemporal_store(result.x, &P.out[lp].x);
            __builtin_nontemporal_store(result.y, &P.out[lp].y);
        }
        if (P.row_cost) {
            unsigned rs = R.n_seg;
#pragma unroll
            for (int d = 4; d >= 1; d >>= 1) rs += __shfl_xor(rs, d);
            if ((lane & 7) == 0 && rs && lrow < im.n_local_rows) atomicAdd(P.row_cost + lrow, rs);
        }
        t_seg += R.n_seg;
        t_cov += R.n_seg > 0 ? 1u : 0u;
        t_sol += is_solid;
        t_ovf += overflow;
        t_ent += n_entries;
    }
    const unsigned s_seg = wave_sum_u32(t_seg);
    const unsigned s_cov = wave_sum_u32(t_cov);
    const unsigned s_sol = wave_sum_u32(t_sol);
    const unsigned s_ovf = wave_sum_u32(t_ovf);
    const unsigned s_ent = wave_sum_u32(t_ent);
    if (lane == 0) {
        FrameCounters* const fc = P.counters + ((blockIdx.x * 4u + static_cast<unsigned>(wave)) % kCounterShards);
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

bool mixed2_fits(int64_t n_cells) { return n_cells < (int64_t{1} << 25); }  // ticket: id << 7 in 32 bits

void launch_walk_mixed2(hipStream_t s, const WalkParams& p) {
    constexpr int TW = 16, TH = 32;
    const int tiles_x = (p.im.res_x + TW - 1) / TW;
    const int tiles_y = (p.im.n_local_rows + TH - 1) / TH;
    if (tiles_x <= 0 || tiles_y <= 0) return;
    WalkParams q = p;
    long long blocks;
    if (p.xcd_mode == 0) {
        blocks = static_cast<long long>(tiles_x) * tiles_y;
    } else {
        const int sb_rows = p.band_rows > 0 ? p.band_rows : 64;
        const int S = sb_rows / TH > 0 ? sb_rows / TH : 1;
        const long long n_sb = static_cast<long long>((tiles_x + S - 1) / S) * ((tiles_y + S - 1) / S);
        blocks = 8ll * ((n_sb + 7) / 8) * S * S;
        q.band_tiles = S;
    }
    if (p.order == 0)
        hipLaunchKernelGGL((walk_composite_mixed2<0>), dim3(static_cast<unsigned>(blocks)), dim3(256), static_cast<size_t>(p.lds_pad), s, q);
    else
        hipLaunchKernelGGL((walk_composite_mixed2<1>), dim3(static_cast<unsigned>(blocks)), dim3(256), static_cast<size_t>(p.lds_pad), s, q);
}

}  // namespace c5
