// Probe: how to make the fp8 conversions saturate (instead of producing NaN) on gfx950.  The assembler rejects a clamp modifier on
// them; MODE.FP16_OVFL (hwreg MODE bit 23) is the documented switch -- it also turns fp16 overflow into 65504.
// hipcc --offload-arch=gfx950 -O3 tools/micro/fp8_clamp_probe.hip -o /tmp/fp8_clamp && /tmp/fp8_clamp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float a = in[i], b = in[i] * 0.5f;
    unsigned plain = 0, clamped = 0, sc_plain = 0, sc_clamped = 0;
    asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(plain) : "v"(a), "v"(b));
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);  // MODE.FP16_OVFL = 1
    asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(clamped) : "v"(a), "v"(b));
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 0);
    const float scale = 0.000244140625f;  // 2^-12: the conversion divides by it
    const float c = a * scale, d = b * scale;
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(sc_plain) : "v"(c), "v"(d), "v"(scale));
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(sc_clamped) : "v"(c), "v"(d), "v"(scale));
    _Float16 hf;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hf) : "v"(a * 100.f));
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 0);
    if ((float)hf == 12345.f) sc_clamped = 0;
    out[4 * i + 0] = plain; out[4 * i + 1] = clamped; out[4 * i + 2] = sc_plain; out[4 * i + 3] = sc_clamped;
}
int main() {
    const float h[8] = {1.0f, 447.0f, 448.0f, 470.0f, 1000.0f, -1000.0f, 1e9f, -3.0f};
    float* d; unsigned* o; unsigned ho[32];
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(ho));
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 8);
    (void)hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i)
        printf("x=%12g (and x/2): cvt_pk %04x  cvt_pk OVFL %04x | scalef32 %04x  scalef32 OVFL %04x\n", h[i], ho[4 * i] & 0xffff,
               ho[4 * i + 1] & 0xffff, ho[4 * i + 2] & 0xffff, ho[4 * i + 3] & 0xffff);
    return 0;
}
