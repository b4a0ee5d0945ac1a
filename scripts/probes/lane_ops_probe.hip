// Probe of three gfx950 lane-crossing behaviours the LDS walk kernel relies on:
//   1. ds_permute_b32 leaves 0 in lanes no active lane pushed to,
//   2. ds_bpermute_b32 adds its immediate offset field to the per-lane byte address,
//   3. v_mov_b32 DPP wave_shr:1 shifts across all 64 lanes (lane 0 gets 0 with bound_ctrl).
// Build: hipcc -O2 --offload-arch=gfx950 lane_ops_probe.hip -o lane_ops_probe ; prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    // 1. lanes 0..9 push (1000 + lane) to lane 3 * lane; everything else pushes 7 to lane 63
    const int dst = (lane < 10) ? 3 * lane : 63;
    const int val = (lane < 10) ? 1000 + lane : 7;
    out[lane] = __builtin_amdgcn_ds_permute(dst << 2, val);
    // 2. bpermute with an immediate offset of 32 bytes (= 8 lanes)
    int r;
    const int addr = (lane >> 3) << 2;
    const int data = 2000 + lane;
    asm volatile("ds_bpermute_b32 %0, %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(data));
    out[64 + lane] = r;
    // 3. DPP wave_shr:1
    out[128 + lane] = __builtin_amdgcn_mov_dpp(3000 + lane, 0x138, 0xf, 0xf, true);
}

int main() {
    int* d;
    int h[192];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    int bad1 = 0, bad2 = 0, bad3 = 0;
    for (int l = 0; l < 64; ++l) {
        int want = 0;
        if (l == 63) want = 7;
        if (l % 3 == 0 && l / 3 < 10) want = 1000 + l / 3;
        if (h[l] != want) { ++bad1; printf("permute lane %d: got %d want %d\n", l, h[l], want); }
        const int w2 = 2000 + (((l >> 3) + 8) & 63);
        if (h[64 + l] != w2) { ++bad2; printf("bpermute lane %d: got %d want %d\n", l, h[64 + l], w2); }
        const int w3 = l == 0 ? 0 : 3000 + l - 1;
        if (h[128 + l] != w3) { ++bad3; printf("dpp lane %d: got %d want %d\n", l, h[128 + l], w3); }
    }
    printf("%s ds_permute zero fill\n%s ds_bpermute offset field\n%s dpp wave_shr:1\n", bad1 ? "FAIL" : "PASS",
           bad2 ? "FAIL" : "PASS", bad3 ? "FAIL" : "PASS");
    return (bad1 || bad2 || bad3) ? 1 : 0;
}
