// Shared device helpers for the tqdne gfx950 kernels.
// Written for CDNA4 (MI355X) only: wave64, MFMA 16x16x32 bf16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

namespace tq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int WAVE = 64;

// Activations live in HBM as (B, T, C) fp32 with C contiguous ("channels-last").
// GroupNorm always has 32 groups, eps 1e-5 (reference tqdne/nn.py:11-13, 90-105).
constexpr int GN_GROUPS = 32;
constexpr float GN_EPS = 1e-5f;
// Per-channel partial statistics are produced per 128-position slot of T.
constexpr int STAT_SLOT = 128;

__device__ __forceinline__ float silu_f(float u) {
    // u * sigmoid(u); v_exp + v_rcp, a few ulp (reference: nn.SiLU in fp32)
    return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));
}
__device__ __forceinline__ float dsilu_f(float u) {
    // d/du [u*sigmoid(u)] = s*(1 + u*(1-s))
    float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    return s * (1.0f + u * (1.0f - s));
}

// fp32 -> (hi, lo) bf16 pair with hi + lo == x to ~2^-17 relative.  Three bf16 MFMA
// products (hi*hi + hi*lo + lo*hi) then reproduce an fp32 product to ~2^-16: the
// "bf16x3" scheme.  gfx950 has no xf32/TF32 MFMA; exact-f32 MFMA runs at 1/16 of the
// bf16 rate, so three bf16 products are >5x faster at better-than-TF32 accuracy.
__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

union Frag {
    bf16x8 v;
    uint4 u;
    uint2 h[2];
};

__device__ __forceinline__ f32x4 mfma_bf16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 3-product accumulate: c += (ah+al)*(bh+bl) minus the al*bl term.
__device__ __forceinline__ f32x4 mfma_x3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
    c = mfma_bf16(al, bh, c);
    c = mfma_bf16(ah, bl, c);
    c = mfma_bf16(ah, bh, c);
    return c;
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): ids i and i + 8 share an L2.  Kernels whose `inner`
// consecutive workgroups re-read the same operand (the query tiles of one (batch, head) all stream its K / V) remap the linear id
// so that those workgroups land on ONE XCD: on XCD x the s-th workgroup gets outer = x + 8 * (s / inner), inner index s % inner.
// Falls back to the identity when the outer count is not a multiple of 8.
__device__ __forceinline__ int xcd_group_id(int bid, int inner, int n_outer) {
    if (n_outer & 7) return bid;
    const int x = bid & 7, s = bid >> 3;
    return (x + 8 * (s / inner)) * inner + (s % inner);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Counter-based dropout RNG: the keep decision of element `e` (= row * C + channel inside one sample) of sample `b` at dropout
// site `site` is  drop_hash(drop_key(seed, site, b), e) >= threshold, re-evaluated wherever the mask is needed (forward staging,
// weight-gradient staging, data-gradient epilogue).  Round 4: a 32-bit finaliser (two 32-bit multiplies per element) on a per-sample
// key that the scalar unit forms once per workgroup / unit; rounds 1-3 ran a 64-bit splitmix per element (three 64-bit multiplies =
// twelve quarter-rate 32-bit ones), which cost the dropout convs +14 % forward, +14...17 % in their weight gradient and +6...8 % in
// their data gradient (profiles/r03_f_layers_train_b64.txt).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // "lowbias32" integer finaliser (bijective, full avalanche)
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// Round 5: the key has TWO words.  With one word every mask of every (sample, site, step) was an XOR-translated window of ONE fixed
// 2^32-entry sequence (mix32(e ^ key)); the second word enters between the finaliser's two multiplies, so different keys select
// different functions of e, not translates of one -- for one more 32-bit add per element (the key words are scalar operands).
struct DropKey { uint32_t a, b; };
__device__ __forceinline__ DropKey drop_key(uint64_t seed, uint32_t site, uint32_t b) {
    uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9u);
    k = mix32(k ^ (uint32_t)(seed >> 32));
    k = mix32(k + 0x85EBCA6Bu * (site + 1u));
    DropKey r;
    r.a = mix32(k + 0xC2B2AE35u * (b + 1u));
    r.b = mix32((k ^ 0x27D4EB2Fu) + 0x165667B1u * (b + 1u));
    return r;
}
__device__ __forceinline__ uint32_t drop_hash(DropKey key, uint32_t e) {
    uint32_t x = e ^ key.a;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x += key.b;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

}  // namespace tq

#define TQ_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)
