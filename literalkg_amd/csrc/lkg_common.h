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
// Reduction over aligned groups of WIDTH lanes (WIDTH a power of two <= 64); every lane of a group ends with the
// group's total.  The butterfly runs on the VALU: DPP lane permutes inside a row of 16 (they fuse into the add /
// max itself), v_permlane16_swap / v_permlane32_swap (gfx950) across rows -- HIP's __shfl_xor is a ds_bpermute,
// an LDS-pipeline round trip per step.  Bit-identical to the xor butterfly (after steps 1 and 2 the lanes of a quad
// hold the same bits, so the mirror permutes deliver what lane ^ 4 / lane ^ 8 would).  Call with all lanes active.
template <int CTRL>
__device__ __forceinline__ float dpp_read(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
struct lkg_sum_op {
    static __device__ __forceinline__ float apply(float a, float b) { return a + b; }
};
struct lkg_max_op {
    static __device__ __forceinline__ float apply(float a, float b) { return fmaxf(a, b); }
};
template <int WIDTH, typename OP>
__device__ __forceinline__ float group_reduce(float v) {
    static_assert(WIDTH >= 1 && WIDTH <= 64 && (WIDTH & (WIDTH - 1)) == 0, "group width");
    if constexpr (WIDTH >= 2) v = OP::apply(v, dpp_read<0xB1>(v));    // quad_perm [1,0,3,2]: lane ^ 1
    if constexpr (WIDTH >= 4) v = OP::apply(v, dpp_read<0x4E>(v));    // quad_perm [2,3,0,1]: lane ^ 2
    if constexpr (WIDTH >= 8) v = OP::apply(v, dpp_read<0x141>(v));   // row_half_mirror: the other quad of 8
    if constexpr (WIDTH >= 16) v = OP::apply(v, dpp_read<0x140>(v));  // row_mirror: the other half of 16
    if constexpr (WIDTH >= 32) {   // rows 1,3 of the first copy <-> rows 0,2 of the second
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
        v = OP::apply(__int_as_float(r[0]), __int_as_float(r[1]));
    }
    if constexpr (WIDTH >= 64) {   // lanes 32-63 of the first copy <-> lanes 0-31 of the second
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
        v = OP::apply(__int_as_float(r[0]), __int_as_float(r[1]));
    }
    return v;
}
// v + (v of lane ^ M) for one butterfly step, M in {8, 16, 32} (the steps ABOVE a sub-group of 8+ lanes)
template <int M>
__device__ __forceinline__ float lane_xor_add(float v) {
    static_assert(M == 8 || M == 16 || M == 32, "lane_xor_add: step");
    if constexpr (M == 8) {
        return v + dpp_read<0x128>(v);    // row_ror:8 inside a row of 16 == lane ^ 8
    } else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
        return __int_as_float(r[0]) + __int_as_float(r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
        return __int_as_float(r[0]) + __int_as_float(r[1]);
    }
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) { return group_reduce<WIDTH, lkg_sum_op>(v); }
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) { return group_reduce<WIDTH, lkg_max_op>(v); }
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_fma(float4 &a, float s, const float4 &x) {
    a.x = fmaf(s, x.x, a.x);
    a.y = fmaf(s, x.y, a.y);
    a.z = fmaf(s, x.z, a.z);
    a.w = fmaf(s, x.w, a.w);
}

// Counter-based dropout mask: keep(seed, row, col) is a pure function (two rounds of the murmur3 32-bit
// finaliser), so the backward regenerates the mask instead of storing it.  torch's Philox stream cannot be
// reproduced from outside ATen, so training-mode parity with the reference is statistical.
__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}
__device__ __forceinline__ unsigned drop_row_key(unsigned long long seed, unsigned long long row) {
    return fmix32((unsigned)seed ^ fmix32((unsigned)row * 0x9E3779B1u + (unsigned)(row >> 32) + (unsigned)(seed >> 32)));
}
__device__ __forceinline__ float drop_scale(unsigned row_key, unsigned col, float p, float inv_keep) {
    const unsigned h = fmix32(row_key ^ (col * 0x27D4EB2Fu + 0x165667B1u));
    return (float)(h >> 8) * (1.0f / 16777216.0f) >= p ? inv_keep : 0.f;
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
