// Probe: which lane's E8M0 scale byte multiplies which operand elements of v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3)
// operands.  B holds 1.0 only in lanes of group g_d (= lane >> 4), elements j of half h (j < 16 or j >= 16); A is all ones with
// scale 1; lane (col, g) of B carries the scale 2^g.  D = 16 * 2^(g of the lane whose scale was applied).  Same for the A side.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(4))) float f4;

__device__ i8v ones_fp6(bool lo, bool hi) {
    // 32 codes of 0x08 (1.0): every 6-bit field 001000; elements 0..15 = bits 0..95 (dwords 0-2), 16..31 = dwords 3-5
    i8v v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lo) { v[0] = 0x08208208; v[1] = (int)0x82082082; v[2] = 0x20820820; }
    if (hi) { v[3] = 0x08208208; v[4] = (int)0x82082082; v[5] = 0x20820820; }
    return v;
}

__global__ void k(float* D, int g_d, int h, int side, int opsel) {
    const int l = threadIdx.x, g = l >> 4;
    const bool mine = g == g_d;
    i8v full = ones_fp6(true, true);
    i8v part = ones_fp6(mine && h == 0, mine && h == 1);
    f4 c = {0, 0, 0, 0};
    const int sc = (127 + g) << (8 * opsel);
    if (opsel == 0) {
        if (side == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(full, part, c, 2, 2, 0, 127, 0, sc);
        else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(part, full, c, 2, 2, 0, sc, 0, 127);
    } else {
        if (side == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(full, part, c, 2, 2, 0, 127, 1, sc);
        else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(part, full, c, 2, 2, 1, sc, 0, 127);
    }
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

int main() {
    float* dD; hipMalloc(&dD, 1024);
    float hD[256];
    for (int opsel = 0; opsel < 2; ++opsel)
    for (int side = 0; side < 2; ++side) {
        printf("%s-side scales (opsel %d): ", side ? "A" : "B", opsel);
        for (int g_d = 0; g_d < 4; ++g_d) for (int h = 0; h < 2; ++h) {
            k<<<1, 64>>>(dD, g_d, h, side, opsel);
            hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
            bool uniform = true;
            for (int i = 1; i < 256; ++i) uniform &= hD[i] == hD[0];
            printf(" data(g=%d,half=%d)->%s%g", g_d, h, uniform ? "" : "~", log2(hD[0] / 16.0));
        }
        printf("\n");
    }
    return 0;
}
