// Helpers of the mixed-precision walk kernel (walk_mixed.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>

#include "device_types.hpp"
#include "kernels.hpp"
#include "walk_common.hpp"

namespace c5 {

constexpr uint32_t kExactBit = 1u << 29;  // GeoRecord::w[0]: the cell has a face evaluated in fp64 (SteepPlanes)
constexpr int kSteepSlotShift = 28;       // GeoRecord::w[1] bits 28-31: which face slots those are
// What a steep cell keeps in its CellRecord slot (128 bytes): the four face planes (c, gx, gy) in double precision
// about the GeoRecord's lattice origin and depth origin.
struct alignas(16) SteepPlanes {
    double p[4][3];
    double pad[4];
};
static_assert(sizeof(SteepPlanes) == sizeof(CellRecord), "SteepPlanes lives in the cell's CellRecord slot");
using SRec8 = int __attribute__((ext_vector_type(8)));

#ifndef C5_MIX_SLOTS
#define C5_MIX_SLOTS 16  // (32 measured slower on the C3 frame, equal at 1200 x 900)
#endif
constexpr int kMixSlots = C5_MIX_SLOTS;  // distinct cells staged per wavefront and step (a power of two)
constexpr int kMixStride = 5;        // 16-byte units per slot: 4 of GeoRecord + 1 of OptRecord.  80 bytes = 20
                                     // banks: sixteen slots start on sixteen different 16-byte bank columns
constexpr unsigned kMixBuckets = 256;

using V4F = float __attribute__((ext_vector_type(4)));
using SRec16 = int __attribute__((ext_vector_type(16)));  // half a CellRecord in scalar registers
using V4U = uint32_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float min3_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// e^x - 1 for -1/8 < x <= 0 in fp32: x (1 + x/2 (1 + x/3 (...))) to x^7 (truncation 5e-10 relative)
__device__ __forceinline__ float expm1_small(float x) {
    float p = fmaf(x, 1.0f / 5040.0f, 1.0f / 720.0f);
    p = fmaf(p, x, 1.0f / 120.0f);
    p = fmaf(p, x, 1.0f / 24.0f);
    p = fmaf(p, x, 1.0f / 6.0f);
    p = fmaf(p, x, 0.5f);
    p = fmaf(p, x, 1.0f);
    return p * x;
}

// exp(x) for x <= 0, the general case (|alpha dz| >= 1/8: rare in this kernel).  exp_nonpositive keeps its
// coefficients in scalar registers for the whole loop — twenty SGPRs, which here would cost a wavefront per SIMD —
// so this copy materialises each constant where it is used (the empty asm keeps it from being hoisted).
__device__ __forceinline__ double exp_nonpositive_local(double x) {
#pragma clang fp contract(fast)
    auto local = [](double c) {
        asm volatile("" : "+s"(c));
        return c;
    };
    x = fmax(x, local(-746.0));
    const double k = rint(x * local(1.4426950408889634074));
    double r = fma(k, local(-6.93147180369123816490e-01), x);
    r = fma(k, local(-1.90821492927058770002e-10), r);
    double p = fma(local(1.0 / 6227020800.0), r, local(1.0 / 479001600.0));
    p = fma(p, r, local(1.0 / 39916800.0));
    p = fma(p, r, local(1.0 / 3628800.0));
    p = fma(p, r, local(1.0 / 362880.0));
    p = fma(p, r, local(1.0 / 40320.0));
    p = fma(p, r, local(1.0 / 5040.0));
    p = fma(p, r, local(1.0 / 720.0));
    p = fma(p, r, local(1.0 / 120.0));
    p = fma(p, r, local(1.0 / 24.0));
    p = fma(p, r, local(1.0 / 6.0));
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, static_cast<int>(k));
}


}  // namespace c5
