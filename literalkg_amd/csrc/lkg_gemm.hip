// Dense fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact f32, bitwise a
// k-ordered fmaf chain), plus the two grouped forms the TransR projection needs.
//
//   C[m,n] = alpha * sum_k opA(A)[m,k] * opB(B)[k,n] + beta * C[m,n] (+ bias[n])
//
// Tiling: 128 x 128 x 16 block tile, 256 threads = 4 waves in 2 x 2, each wave a 64 x 64 tile as 2 x 2
// MFMA 32x32 accumulators (64 accumulator registers).  Both operands are staged K-MAJOR in LDS
// (As[k][m], Bs[k][n], row pitch 132 floats) so that the MFMA operand read -- lane l needs
// A[m = l & 31][k = l >> 5] -- is one conflict-free ds_read_b32 per operand: 32 consecutive floats per
// half-wave, the two halves one k-row apart.  Global loads of tile t+1 are issued before the MFMAs of
// tile t and written to the other LDS buffer afterwards (one barrier per k-tile).
//
// Grouped forms (relation-grouped W_r GEMM, model.py:372/390-395 without materialising W_r[r]):
//   rows mode : rows [seg[g], seg[g+1]) of A and C use B + g * stride_b     (projection, data gradient)
//   k mode    : the reduction runs over rows [seg[g], seg[g+1]) of A^T and B, output C + g * stride_c
//               (weight gradient  g_W[r] = X_r^T G_r)
// Plain long-K / small-output products (nn.Linear weight gradients, K = n_entities) are split over K
// with f32 atomic accumulation into a zeroed C.
#include <algorithm>
#include <type_traits>

#include "lkg_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, PITCH = 132;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    long m, n, k;
    float alpha, beta;
    const float *a;
    long lda;
    const float *b;
    long ldb;
    float *c;
    long ldc;
    const float *bias;
    const int *seg;      // grouped forms: device int32[n_groups + 1]
    long stride_b, stride_c;
    int mode;            // 0 plain, 1 rows grouped, 2 k grouped
    int k_splits;        // plain mode only (atomic accumulation when > 1)
    int tiles_m, tiles_n;
};

// Stage a (rows x BK) tile whose global layout has K contiguous (A not transposed / B transposed):
// element (r, k) at src[r * ld + k].  Thread t owns row t/2 and 8 consecutive k.
struct KContigLoader {
    float v[8];
    __device__ __forceinline__ void load(const float *src, long ld, long r0, long r_end, long k0, long k_end, int t,
                                         bool fast) {
        const long r = r0 + (t >> 1);
        const long k = k0 + (t & 1) * 8;
        if (fast) {
            const float4 *p = reinterpret_cast<const float4 *>(src + r * ld + k);
            const float4 x = p[0], y = p[1];
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (r < r_end && k + j < k_end) ? src[r * ld + k + j] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float *lds, int t) const {
        const int r = t >> 1, k = (t & 1) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) lds[(k + j) * PITCH + r] = v[j];
    }
};

// Stage a (BK x cols) tile whose global layout has the m/n index contiguous (A transposed / B not
// transposed): element (k, c) at src[k * ld + c].  Thread t owns k = t/16 and 8 consecutive columns.
struct MContigLoader {
    float v[8];
    __device__ __forceinline__ void load(const float *src, long ld, long c0, long c_end, long k0, long k_end, int t,
                                         bool fast) {
        const long k = k0 + (t >> 4);
        const long c = c0 + (t & 15) * 8;
        if (fast) {
            const float4 *p = reinterpret_cast<const float4 *>(src + k * ld + c);
            const float4 x = p[0], y = p[1];
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (k < k_end && c + j < c_end) ? src[k * ld + c + j] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float *lds, int t) const {
        float *p = lds + (t >> 4) * PITCH + (t & 15) * 8;
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * PITCH];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int tile = blockIdx.x;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int z = blockIdx.y;

    long m_lo = 0, m_hi = g.m, k_lo = 0, k_hi = g.k;
    const float *A = g.a, *B = g.b;
    float *C = g.c;
    bool atomic_out = false;
    if (g.mode == 1) {
        m_lo = g.seg[z];
        m_hi = g.seg[z + 1];
        B += (long)z * g.stride_b;
    } else if (g.mode == 2) {
        k_lo = g.seg[z];
        k_hi = g.seg[z + 1];
        C += (long)z * g.stride_c;
    } else if (g.k_splits > 1) {
        const long per = ((g.k + g.k_splits - 1) / g.k_splits + BK - 1) / BK * BK;
        k_lo = (long)z * per;
        k_hi = min(g.k, k_lo + per);
        atomic_out = true;
    }
    const long m0 = m_lo + (long)tm * BM;
    const long n0 = (long)tn * BN;
    if (m0 >= m_hi || n0 >= g.n) return;
    if (k_lo >= k_hi && (g.mode != 2)) {
        if (atomic_out) return;
    }

    // can this block use unguarded 16-byte loads?
    const bool a_al = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_al = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    const bool m_full = m0 + BM <= m_hi, n_full = n0 + BN <= g.n;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typename std::conditional<TA, MContigLoader, KContigLoader>::type la;
    typename std::conditional<TB, KContigLoader, MContigLoader>::type lb;

    auto fetch = [&](long k0) {
        const bool k_full = k0 + BK <= k_hi;
        // k offsets are multiples of 8 from k_lo; 16-byte alignment along k needs k_lo % 4 == 0
        if constexpr (TA)
            la.load(A, g.lda, m0, m_hi, k0, k_hi, t, a_al && m_full && k_full && (m0 % 4 == 0));
        else
            la.load(A, g.lda, m0, m_hi, k0, k_hi, t, a_al && m_full && k_full && (k0 % 4 == 0));
        if constexpr (TB)
            lb.load(B, g.ldb, n0, g.n, k0, k_hi, t, b_al && n_full && k_full && (k0 % 4 == 0));
        else
            lb.load(B, g.ldb, n0, g.n, k0, k_hi, t, b_al && n_full && k_full);
    };

    int buf = 0;
    if (k_lo < k_hi) {
        fetch(k_lo);
        la.store(As[0], t);
        lb.store(Bs[0], t);
    }
    __syncthreads();
    for (long k0 = k_lo; k0 < k_hi; k0 += BK) {
        const bool more = k0 + BK < k_hi;
        if (more) fetch(k0 + BK);
        const float *as = As[buf] + wm * 64 + (lane & 31) + (lane >> 5) * PITCH;
        const float *bs = Bs[buf] + wn * 64 + (lane & 31) + (lane >> 5) * PITCH;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = as[kk * PITCH], a1 = as[kk * PITCH + 32];
            const float b0 = bs[kk * PITCH], b1 = bs[kk * PITCH + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            la.store(As[buf ^ 1], t);
            lb.store(Bs[buf ^ 1], t);
        }
        __syncthreads();
        buf ^= 1;
    }

    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.n) continue;
            const float bv = (g.bias && (!atomic_out || z == 0)) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= m_hi) continue;
                float *dst = C + row * g.ldc + col;
                const float val = g.alpha * acc[i][j][r] + bv;
                if (atomic_out)
                    atomicAdd(dst, val);
                else
                    *dst = (g.beta != 0.f) ? fmaf(g.beta, *dst, val) : val;
            }
        }
}

int run(bool ta, bool tb, const GemmArgs &g, dim3 grid, hipStream_t s) {
    if (!ta && !tb)
        hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, s, g);
    else if (!ta && tb)
        hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, s, g);
    else if (ta && !tb)
        hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, s, g);
    LKG_CHECK_LAUNCH("lkg_gemm_f32");
    return LKG_OK;
}

}  // namespace

extern "C" int lkg_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, float alpha,
                            const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                            int64_t ldc, const float *bias, void *stream) {
    LKG_REQUIRE(m >= 0 && n >= 0 && k >= 0, "lkg_gemm_f32: negative size");
    if (m == 0 || n == 0) return LKG_OK;
    LKG_REQUIRE(c && ldc >= n, "lkg_gemm_f32: bad C (ldc=%lld, n=%lld)", (long long)ldc, (long long)n);
    LKG_REQUIRE(k == 0 || (a && b), "lkg_gemm_f32: null operand");
    LKG_REQUIRE(k == 0 || (lda >= (trans_a ? m : k) && ldb >= (trans_b ? k : n)), "lkg_gemm_f32: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.m = m; g.n = n; g.k = k; g.alpha = alpha; g.beta = beta;
    g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.c = c; g.ldc = ldc; g.bias = bias;
    g.mode = 0;
    g.tiles_m = (int)((m + BM - 1) / BM);
    g.tiles_n = (int)((n + BN - 1) / BN);
    const long tiles = (long)g.tiles_m * g.tiles_n;
    LKG_REQUIRE(tiles < INT32_MAX, "lkg_gemm_f32: too many tiles");
    // long-K, small-output products: split K across the chip, accumulate with f32 atomics
    int splits = 1;
    if (beta == 0.f && tiles < 256 && k >= 8192) {
        splits = (int)std::min<long>(std::min<long>(1024 / tiles, k / 2048), 65535);
        if (splits < 2) splits = 1;
    }
    g.k_splits = splits;
    if (splits > 1) {
        if (ldc == n) {
            if (hipMemsetAsync(c, 0, sizeof(float) * m * n, s) != hipSuccess) {
                lkg_set_error("lkg_gemm_f32: hipMemsetAsync failed");
                return LKG_ERR_HIP;
            }
        } else if (hipMemset2DAsync(c, sizeof(float) * ldc, 0, sizeof(float) * n, m, s) != hipSuccess) {
            lkg_set_error("lkg_gemm_f32: hipMemset2DAsync failed");
            return LKG_ERR_HIP;
        }
    }
    return run(trans_a != 0, trans_b != 0, g, dim3((unsigned)tiles, (unsigned)splits), s);
}

extern "C" int lkg_grouped_gemm_f32(int32_t mode, int32_t n_groups, const int32_t *seg, int64_t max_seg_len,
                                    int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, float alpha,
                                    const float *a, int64_t lda, const float *b, int64_t ldb, int64_t stride_b,
                                    float beta, float *c, int64_t ldc, int64_t stride_c, void *stream) {
    LKG_REQUIRE(mode == 1 || mode == 2, "lkg_grouped_gemm_f32: mode must be 1 (rows) or 2 (k)");
    LKG_REQUIRE(n_groups >= 0 && n_groups <= 65535 && max_seg_len >= 0, "lkg_grouped_gemm_f32: bad group count");
    if (n_groups == 0 || n == 0) return LKG_OK;
    LKG_REQUIRE(seg && a && b && c && ldc >= n, "lkg_grouped_gemm_f32: null pointer / bad ldc");
    LKG_REQUIRE(mode == 1 ? !trans_a : (trans_a && !trans_b),
                "lkg_grouped_gemm_f32: rows mode needs A row-major, k mode needs A^T and B stored k-major");
    GemmArgs g{};
    g.alpha = alpha; g.beta = beta; g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.c = c; g.ldc = ldc;
    g.bias = nullptr; g.seg = seg; g.stride_b = stride_b; g.stride_c = stride_c; g.mode = mode; g.k_splits = 1;
    g.n = n;
    if (mode == 1) {
        g.m = max_seg_len;   // upper bound; the kernel reads the true range from seg
        g.k = k;
        if (max_seg_len == 0) return LKG_OK;
    } else {
        g.m = m;
        g.k = max_seg_len;
        if (m == 0) return LKG_OK;
    }
    g.tiles_m = (int)((g.m + BM - 1) / BM);
    g.tiles_n = (int)((n + BN - 1) / BN);
    const long tiles = (long)g.tiles_m * g.tiles_n;
    LKG_REQUIRE(tiles < INT32_MAX, "lkg_grouped_gemm_f32: too many tiles");
    return run(trans_a != 0, trans_b != 0, g, dim3((unsigned)tiles, (unsigned)n_groups), (hipStream_t)stream);
}
