// Batch-side primitives of the TransR scoring path (model.py:371-395): grouping the batch by relation
// (so that W_r is applied as one GEMM per relation instead of a B x C x D gather), row gather of the
// batch's entity rows, and the scatter-add of their gradients (autograd of the three `index` ops).
#include <algorithm>

#include "lkg_common.h"

namespace {

constexpr int GB_WAVES = 16;

// Deterministic (stable) counting sort of `keys` by value, one workgroup.
//   perm[p] = position in the input of the p-th element in key order, seg[k] = first p of key k.
__global__ __launch_bounds__(1024) void group_by_key_kernel(long n, int n_keys, const long *__restrict__ keys,
                                                             int *__restrict__ perm, int *__restrict__ seg) {
    extern __shared__ int sm[];   // [n_keys][GB_WAVES] running offsets
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < n_keys * GB_WAVES; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    const long chunk = ((n + GB_WAVES - 1) / GB_WAVES + 63) / 64 * 64;
    const long lo = min(n, (long)w * chunk), hi = min(n, lo + chunk);
    for (long i = lo + lane; i < hi; i += 64) {
        const int k = (int)min(max(keys[i], 0L), (long)n_keys - 1);
        atomicAdd(&sm[k * GB_WAVES + w], 1);
    }
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int k = 0; k < n_keys; ++k) {
            seg[k] = run;
            for (int j = 0; j < GB_WAVES; ++j) {
                const int c = sm[k * GB_WAVES + j];
                sm[k * GB_WAVES + j] = run;
                run += c;
            }
        }
        seg[n_keys] = run;
    }
    __syncthreads();
    for (long i0 = lo; i0 < hi; i0 += 64) {
        const long i = i0 + lane;
        const bool valid = i < hi;
        const int k = valid ? (int)min(max(keys[i], 0L), (long)n_keys - 1) : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int k0 = __shfl(k, leader, 64);
            const unsigned long long same = __ballot(k == k0);
            const int base = sm[k0 * GB_WAVES + w];
            if (k == k0) {
                const int rank = __popcll(same & ((1ull << lane) - 1ull));
                perm[base + rank] = (int)i;
            }
            // all lanes of the wave have read `base` before lane `leader` bumps it (same wave, in order)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane == leader) sm[k0 * GB_WAVES + w] = base + __popcll(same);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            todo &= ~same;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(long n, int d, const float *__restrict__ src, long lds,
                                                           const long *__restrict__ idx, const int *__restrict__ perm,
                                                           float *__restrict__ dst, long ldd) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const long p = perm ? perm[i] : i;
    const long r = idx ? idx[p] : p;
    if constexpr (VEC) {
        const float4 *s = reinterpret_cast<const float4 *>(src + r * lds);
        float4 *o = reinterpret_cast<float4 *>(dst + i * ldd);
        for (int c = lane; c < d / 4; c += 64) o[c] = s[c];
    } else {
        for (int c = lane; c < d; c += 64) dst[i * ldd + c] = src[r * lds + c];
    }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(long n, int d, const float *__restrict__ src, long lds,
                                                                const long *__restrict__ idx,
                                                                const int *__restrict__ perm, float *__restrict__ dst,
                                                                long ldd) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const long p = perm ? perm[i] : i;
    const long r = idx ? idx[p] : p;
    for (int c = lane; c < d; c += 64) atomicAdd(dst + r * ldd + c, src[i * lds + c]);
}

__global__ void gather_i64_kernel(long n, const long *__restrict__ src, const int *__restrict__ perm,
                                  long *__restrict__ dst) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = src[perm[i]];
}

}  // namespace

extern "C" int lkg_group_by_key_i64(int64_t n, int32_t n_keys, const int64_t *keys, int32_t *perm, int32_t *seg,
                                    void *stream) {
    LKG_REQUIRE(n >= 0 && n < INT32_MAX && n_keys >= 1, "lkg_group_by_key_i64: bad sizes");
    LKG_REQUIRE(n_keys <= 1024, "lkg_group_by_key_i64: at most 1024 distinct keys supported (got %d)", n_keys);
    LKG_REQUIRE(seg && (n == 0 || (keys && perm)), "lkg_group_by_key_i64: null pointer");
    hipLaunchKernelGGL(group_by_key_kernel, dim3(1), dim3(1024), sizeof(int) * n_keys * GB_WAVES, (hipStream_t)stream,
                       (long)n, n_keys, (const long *)keys, perm, seg);
    LKG_CHECK_LAUNCH("lkg_group_by_key_i64");
    return LKG_OK;
}

extern "C" int lkg_gather_rows_f32(int64_t n, int32_t d, const float *src, int64_t lds, const int64_t *idx,
                                   const int32_t *perm, float *dst, int64_t ldd, void *stream) {
    LKG_REQUIRE(n >= 0 && d > 0 && lds >= d && ldd >= d, "lkg_gather_rows_f32: bad sizes");
    if (n == 0) return LKG_OK;
    LKG_REQUIRE(src && dst, "lkg_gather_rows_f32: null pointer");
    const dim3 grid((unsigned)((n + 3) / 4));
    const bool vec = d % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && lkg_aligned16(src) && lkg_aligned16(dst);
    if (vec)
        hipLaunchKernelGGL((gather_rows_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, (long)n, d, src,
                           (long)lds, (const long *)idx, perm, dst, (long)ldd);
    else
        hipLaunchKernelGGL((gather_rows_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, (long)n, d, src,
                           (long)lds, (const long *)idx, perm, dst, (long)ldd);
    LKG_CHECK_LAUNCH("lkg_gather_rows_f32");
    return LKG_OK;
}

extern "C" int lkg_scatter_add_rows_f32(int64_t n, int32_t d, const float *src, int64_t lds, const int64_t *idx,
                                        const int32_t *perm, float *dst, int64_t ldd, void *stream) {
    LKG_REQUIRE(n >= 0 && d > 0 && lds >= d && ldd >= d, "lkg_scatter_add_rows_f32: bad sizes");
    if (n == 0) return LKG_OK;
    LKG_REQUIRE(src && dst, "lkg_scatter_add_rows_f32: null pointer");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (long)n, d, src, (long)lds, (const long *)idx, perm, dst, (long)ldd);
    LKG_CHECK_LAUNCH("lkg_scatter_add_rows_f32");
    return LKG_OK;
}

extern "C" int lkg_gather_i64(int64_t n, const int64_t *src, const int32_t *perm, int64_t *dst, void *stream) {
    LKG_REQUIRE(n >= 0, "lkg_gather_i64: negative n");
    if (n == 0) return LKG_OK;
    LKG_REQUIRE(src && perm && dst, "lkg_gather_i64: null pointer");
    const int64_t blocks = std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(gather_i64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long)n,
                       (const long *)src, perm, (long *)dst);
    LKG_CHECK_LAUNCH("lkg_gather_i64");
    return LKG_OK;
}
