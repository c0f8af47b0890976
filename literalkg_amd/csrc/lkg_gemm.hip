// Dense fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact f32, bitwise a
// k-ordered fmaf chain), plus the two grouped forms the TransR projection needs.
//
//   C[m,n] = alpha * sum_k opA(A)[m,k] * opB(B)[k,n] + beta * C[m,n] (+ bias[n])
//
// Tiling: 128 x 128 x 16 block tile, 256 threads = 4 waves in 2 x 2, each wave a 64 x 64 tile as 2 x 2
// MFMA 32x32 accumulators (64 accumulator registers).
//
// LDS images follow the GLOBAL layout of each operand, so staging is always a straight 16-byte copy
// (no transposing scatter):
//   K-contiguous operand (A row-major, or B given transposed = nn.Linear weight): image [row][k], pitch 20
//     floats; a lane fetches FOUR consecutive k of its row with one ds_read_b128 (conflict-free: 20 l mod 64
//     hits 16 distinct 4-bank slots for any 16 lanes of a half-wave);
//   M/N-contiguous operand (A given transposed, B row-major): image [k][col], pitch 132 floats; a lane
//     fetches its four k with four ds_read_b32 (32 consecutive floats per half-wave, halves one row apart).
// The 32x32x2 MFMA wants k = 0 on lanes 0-31 and k = 1 on lanes 32-63.  Inside a group of 8 k the four
// MFMAs take k = j on the lower half and k = 4 + j on the upper half (j = 0..3): any pairing is legal as
// long as A and B agree, and this one lets a lane read 4 consecutive k.
// Global loads of tile t+1 are issued before the MFMAs of tile t and written to the other LDS buffer
// afterwards (one barrier per k-tile).  Workgroup ids are remapped so that each XCD walks a contiguous
// range of tiles (n fastest): the tiles that share an A panel run on the same L2 (measured +1 %).
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

constexpr int BM = 128, BN = 128, BK = 16;   // BK = 32 measured slower on the tall-skinny shapes (occupancy)
constexpr int EPT = BK / 2;      // floats per thread per operand tile (128 * BK / 256)
constexpr int PK = BK + 4;      // pitch of a [row][k] image
constexpr int PM = BM + 4;      // pitch of a [k][col] image
constexpr int IMG = (BM * PK > BK * PM) ? BM * PK : BK * PM;   // floats per operand image
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

// K-contiguous operand: element (r, k) at src[r * ld + k]; LDS image [r][k].  Thread t owns row t/2 and
// EPT = 8 consecutive k.
struct KContig {
    float v[EPT];
    __device__ __forceinline__ void load(const float *src, long ld, long r0, long r_end, long k0, long k_end, int t,
                                         bool fast) {
        const long r = r0 + (t >> 1);
        const long k = k0 + (t & 1) * EPT;
        if (fast) {
            const float4 *p = reinterpret_cast<const float4 *>(src + r * ld + k);
#pragma unroll
            for (int q = 0; q < EPT / 4; ++q) {
                const float4 x = p[q];
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            }
        } else {
            // edge tile / unaligned operand: branch-free (clamped address + select) so that the loads stay
            // in flight across the MFMAs instead of being fenced by a wait at every guard's join
            const float *row = src + min(r, r_end - 1) * ld;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const float x = row[min(k + j, k_end - 1)];
                v[j] = (r < r_end && k + j < k_end) ? x : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float *lds, int t) const {
        float *p = lds + (t >> 1) * PK + (t & 1) * EPT;
#pragma unroll
        for (int q = 0; q < EPT / 4; ++q)
            *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
    // the 4 k-values of MFMA group g for the 32 rows starting at r0 (lane l: row r0 + (l & 31))
    static __device__ __forceinline__ float4 frag(const float *lds, int r0, int g, int lane) {
        return *reinterpret_cast<const float4 *>(lds + (r0 + (lane & 31)) * PK + 8 * g + (lane >> 5) * 4);
    }
};

// M/N-contiguous operand: element (k, c) at src[k * ld + c]; LDS image [k][c].  Thread t owns k = t/16 and
// EPT = 8 consecutive columns.
struct MContig {
    static constexpr int TPK = BM / EPT;   // threads per k-row
    float v[EPT];
    __device__ __forceinline__ void load(const float *src, long ld, long c0, long c_end, long k0, long k_end, int t,
                                         bool fast) {
        const long k = k0 + t / TPK;
        const long c = c0 + (t % TPK) * EPT;
        if (fast) {
            const float4 *p = reinterpret_cast<const float4 *>(src + k * ld + c);
#pragma unroll
            for (int q = 0; q < EPT / 4; ++q) {
                const float4 x = p[q];
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            }
        } else {
            const float *row = src + min(k, k_end - 1) * ld;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const float x = row[min(c + j, c_end - 1)];
                v[j] = (k < k_end && c + j < c_end) ? x : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float *lds, int t) const {
        float *p = lds + (t / TPK) * PM + (t % TPK) * EPT;
#pragma unroll
        for (int q = 0; q < EPT / 4; ++q)
            *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
    static __device__ __forceinline__ float4 frag(const float *lds, int c0, int g, int lane) {
        const float *p = lds + (8 * g + (lane >> 5) * 4) * PM + c0 + (lane & 31);
        return make_float4(p[0], p[PM], p[2 * PM], p[3 * PM]);
    }
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][IMG];
    __shared__ __attribute__((aligned(16))) float Bs[2][IMG];
    using LA = typename std::conditional<TA, MContig, KContig>::type;
    using LB = typename std::conditional<TB, KContig, MContig>::type;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap: workgroup b runs on XCD b % 8; give every XCD a contiguous range of logical tiles
    const int tiles = g.tiles_m * g.tiles_n;
    int tile = blockIdx.x;
    {
        const int cpx = tiles >> 3, rem = tiles & 7;
        const int x = tile & 7, slot = tile >> 3;
        tile = x * cpx + min(x, rem) + slot;
    }
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
    if (atomic_out && k_lo >= k_hi) return;

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

    LA la;
    LB lb;
    auto fetch = [&](long k0) {
        const bool k_full = k0 + BK <= k_hi;
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
#pragma unroll
        for (int grp = 0; grp < BK / 8; ++grp) {
            const float4 a0 = LA::frag(As[buf], wm * 64, grp, lane), a1 = LA::frag(As[buf], wm * 64 + 32, grp, lane);
            const float4 b0 = LB::frag(Bs[buf], wn * 64, grp, lane), b1 = LB::frag(Bs[buf], wn * 64 + 32, grp, lane);
#define LKG_STEP(F)                                                                         \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.F, b0.F, acc[0][0], 0, 0, 0);       \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.F, b1.F, acc[0][1], 0, 0, 0);       \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.F, b0.F, acc[1][0], 0, 0, 0);       \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.F, b1.F, acc[1][1], 0, 0, 0);
            LKG_STEP(x) LKG_STEP(y) LKG_STEP(z) LKG_STEP(w)
#undef LKG_STEP
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
