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
#endif
