// Shared helpers for the gfx950 LiteralKG kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/literalkg_hip.h"

#define LKG_WAVE 64

void lkg_set_error(const char *fmt, ...);

#define LKG_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            lkg_set_error(__VA_ARGS__);             \
            return LKG_ERR_INVALID_ARG;             \
        }                                           \
    } while (0)

#define LKG_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            lkg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return LKG_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

static inline bool lkg_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#ifdef __HIPCC__
// ---- wave64 cross-lane reductions -------------------------------------------------
// Sum over aligned groups of WIDTH lanes (WIDTH a power of two <= 64); every lane of a
// group ends with the group's total.
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = WIDTH / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int m = WIDTH / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_fma(float4 &a, float s, const float4 &x) {
    a.x = fmaf(s, x.x, a.x);
    a.y = fmaf(s, x.y, a.y);
    a.z = fmaf(s, x.z, a.z);
    a.w = fmaf(s, x.w, a.w);
}

// numerically safe -logsigmoid(x) = softplus(-x), the form ATen uses:
// -(min(x,0) - log1p(exp(-|x|)))
__device__ __forceinline__ float neg_logsigmoid(float x) {
    return -(fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// tanh on the hot elementwise paths (attention logits: 4 per lane per edge; gate blend).  The ocml tanhf costs
// ~40 VALU instructions; this form is ~15: an odd series below 0.25 (truncation < 2e-9 relative) and
// 1 - 2 / (exp(2|x|) + 1) above, on v_exp_f32 / v_rcp_f32.  Absolute error <= 2e-7 over the real line.
__device__ __forceinline__ float tanh_series(float x) {   // |x| < 0.25 only
    const float x2 = x * x;
    return x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
}
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fminf(fabsf(x), 10.f);
    const float poly = tanh_series(x);
    const float e = __expf(2.f * ax);
    const float big = copysignf(1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f), x);
    return ax < 0.25f ? poly : big;
}
// sigmoid on v_exp_f32 / v_rcp_f32 (relative error ~1e-6); the exponent is clamped so exp never overflows
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.f + __expf(-fmaxf(fminf(x, 80.f), -80.f)));
}
#endif
