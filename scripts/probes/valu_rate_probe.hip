// Issue rate of a few vector instructions on gfx950: N independent chains per lane, 8 wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const float m = 0.999f, c = 0.001f;
    const v2f pm = {m, m}, pc = {c, c};
    _Float16 h = static_cast<_Float16>(seed);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {  // 8 x v_fma_f32
                a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
                a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
            } else if (KIND == 1) {  // 4 x v_pk_fma_f32 (the same 8 FMAs)
                p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
                p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
            } else if (KIND == 2) {  // 4 x v_fma_f64
                d0 = fma(d0, 0.999, 0.001); d1 = fma(d1, 0.999, 0.001); d2 = fma(d2, 0.999, 0.001); d3 = fma(d3, 0.999, 0.001);
            } else {  // 8 x v_fma_mix_f32
                a0 = fmaf(static_cast<float>(h), m, a0); a1 = fmaf(static_cast<float>(h), m, a1); a2 = fmaf(static_cast<float>(h), m, a2); a3 = fmaf(static_cast<float>(h), m, a3);
                a4 = fmaf(static_cast<float>(h), m, a4); a5 = fmaf(static_cast<float>(h), m, a5); a6 = fmaf(static_cast<float>(h), m, a6); a7 = fmaf(static_cast<float>(h), m, a7);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + static_cast<float>(d0 + d1 + d2 + d3);
}
template <int KIND>
static void run(const char* name, int instr_per_iter, float* out) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: blocks * 4 waves / (256 CUs * 4 SIMDs) * iters * instr_per_iter
    const double per_simd = double(blocks) * 4 / 1024 * iters * instr_per_iter;
    std::printf("%-14s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n", name, ms, ms * 1e6 / per_simd);
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", 64, out);
    run<1>("v_pk_fma_f32", 32, out);
    run<2>("v_fma_f64", 32, out);
    run<3>("v_fma_mix_f32", 64, out);
    return 0;
}
