// Weight-gradient product on the fp16 matrix cores ("f16 x 2", the k-major twin of lkg_gemm_tall.hip):
//
//   C[m, n] = sum over the k rows of  A[k, m] * B[k, n]          (A = dY or [g_gpre | g_zpre], B = the layer input / a
//                                                                 literal panel; k = entities, millions; m, n <= ~1000)
//
// Both operands arrive k-major (a thread sees 8 consecutive COLUMNS of one k), so the scale that keeps the fp16 split
// exact has to be per COLUMN of each operand: column j is multiplied by the power of two that puts its largest
// magnitude into [2^13, 2^14), every element is split into hi = fp16(a') (round toward zero) and
// mid = fp16((a' - hi) * 2^11), and  a'.b' = hi_a hi_b + hs_a mid_b + mid_a hs_b  with hs = hi * 2^-11 (an exact exponent
// shift): THREE v_mfma_f32_32x32x16_f16 per 16 k and 32 x 32 tile into ONE f32 accumulator, unscaled by an exact ldexp
// by -(e_m + e_n) in the epilogue -- against the six bf16 MFMAs (and the three-plane split) of lkg_gemm_f32's engine 2.
// Elements down to 2^-16 of their column's maximum carry the full 22 bits, smaller ones lose bits of the two cross
// terms only (their absolute error stays below 2^-35 of the column maximum times |b|): normwise the result is as
// accurate as an f32 GEMM's (tests against f64).  The column maxima are inputs: the producers of the operands emit
// them while they write (lkg_gate_blend_bwd_f32, lkg_row_absmax_f32's column output) or they are computed once for
// constant tables (lkg_col_absmax_f32); without them the caller stays on the bf16 x 3 engine.
//
// Structure = lkg_gemm.hip's engine 2: 128 x 128 tile, 4 waves of 64 x 64, 16-k steps, planes [16 k][128 cols] written
// as they arrive, fragments by gfx950's transposing LDS read (ds_read_b64_tr_b16), one register set, one barrier per
// step (barrier -> split + write tile t+1 -> re-issue the loads of tile t+2 -> 12 MFMAs of tile t), split-K over the
// grid with f32 atomics into the zeroed output.  32 KB of LDS (two fp16 planes per operand and buffer).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "lkg_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, PLANE = BM * BK;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct WgArgs {
    const float *a, *b;
    long lda, ldb;
    const float *a_colmax, *b_colmax;
    float *c;
    long ldc;
    long m, n, k;
    int tiles_m, tiles_n, k_splits;
    int a_aligned, b_aligned;    // rows 16-byte aligned and k a multiple of 16: whole tiles may use unguarded 16-byte loads
};

// exponent e with max * 2^e in [2^13, 2^14)   (0 for max == 0 / denormal / non-finite; clamped so that ldexp stays finite)
__device__ __forceinline__ int scale_exponent(float mx) {
    const int ex = (__float_as_int(mx) >> 23) & 0xff;
    if (ex == 0 || ex == 0xff) return 0;
    return max(-100, min(100, 13 - (ex - 127)));
}

__device__ __forceinline__ int kmajor_off(int k, int c) {   // element offset of (k, c) inside a plane (lkg_gemm.hip)
    return k * BM + ((((c >> 5) ^ (k & 3)) << 5) | (c & 31));
}
// plane, the 32 columns starting at c0 (a multiple of 32): this lane's column c0 + (lane & 31), k = 8 (lane >> 5) ..
__device__ __forceinline__ f16x8 kmajor_frag(const _Float16 *plane, int c0, int lane) {
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int col = c0 + (lane & 16) + 4 * pp;
    const int k0 = 8 * (lane >> 5) + q;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(plane + kmajor_off(k0, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(plane + kmajor_off(k0 + 4, col)));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, both);
}

// 8 consecutive columns of one k row -> their hi / mid halves in the two planes of an operand
// (the column scales 2^e come from LDS at every step: 16 registers held across the MFMA block would cost an occupancy step)
__device__ __forceinline__ void split_store(const float (&v)[8], const float *scale8, _Float16 *planes, int t) {
    typedef __fp16 fp16x8 __attribute__((ext_vector_type(8)));
    fp16x8 hv, mv;
    const float4 s0 = *reinterpret_cast<const float4 *>(scale8), s1 = *reinterpret_cast<const float4 *>(scale8 + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float a0 = v[j] * sc[j], a1 = v[j + 1] * sc[j + 1];
        const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(a0, a1);
        const fp16x2 m = __builtin_amdgcn_cvt_pkrtz((a0 - (float)h[0]) * 2048.f, (a1 - (float)h[1]) * 2048.f);
        hv[j] = h[0]; hv[j + 1] = h[1];
        mv[j] = m[0]; mv[j + 1] = m[1];
    }
    _Float16 *p = planes + kmajor_off(t >> 4, (t & 15) * 8);
    *reinterpret_cast<fp16x8 *>(p) = hv;
    *reinterpret_cast<fp16x8 *>(p + PLANE) = mv;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgrad_f16x2_kernel(WgArgs g) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[2][2][2 * PLANE];     // [buffer][operand][hi, mid]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup -> (k split, tile): consecutive workgroup ids go round the 8 XCDs, and every tile of a split re-reads the
    // split's rows of one operand -- so all tiles of a split are placed on ONE XCD, next to each other in dispatch order:
    // the re-reads hit that XCD's L2 instead of going to the fabric again (split = 8 * (slot / tiles) + xcd).
    const int tiles = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = (slot / tiles) * 8 + xcd, tile = slot % tiles;
    if (split >= g.k_splits) return;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const long m0 = (long)tm * BM, n0 = (long)tn * BN;
    const long per = ((g.k + g.k_splits - 1) / g.k_splits + BK - 1) / BK * BK;
    const long k_lo = (long)split * per, k_hi = min(g.k, k_lo + per);
    if (k_lo >= k_hi) return;

    // this thread's 8 columns of either operand and their exponents (fixed for the whole k loop)
    const int kr = t >> 4, c8 = (t & 15) * 8;
    __shared__ __attribute__((aligned(16))) float scale_s[2][BM];              // 2^e of the tile's columns, per operand
    if (t < BM) scale_s[0][t] = ldexpf(1.f, scale_exponent(g.a_colmax[min(m0 + t, g.m - 1)]));
    else scale_s[1][t - BM] = ldexpf(1.f, scale_exponent(g.b_colmax[min(n0 + t - BM, g.n - 1)]));
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float la[8], lb[8];
    auto load8 = [&](float (&v)[8], const float *src, long ld, long c0, long c_end, long k, bool fast) {
        if (fast) {
            const float4 *p = reinterpret_cast<const float4 *>(src + k * ld + c0 + c8);
            const float4 x0 = p[0], x1 = p[1];
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        } else {
            const float *row = src + min(k, k_hi - 1) * ld;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = row[min(c0 + c8 + j, c_end - 1)];
                v[j] = (k < k_hi && c0 + c8 + j < c_end) ? x : 0.f;
            }
        }
    };
    auto stage = [&](_Float16 (*D)[2 * PLANE]) {     // one operand at a time: its planes are dead before the next split
        split_store(la, scale_s[0] + c8, D[0], t);
        __builtin_amdgcn_sched_barrier(0);
        split_store(lb, scale_s[1] + c8, D[1], t);
    };
    auto mma = [&](const _Float16 (*S)[2 * PLANE]) {
        f16x8 ah[2], am[2], bh[2], bm[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = kmajor_frag(S[0], wm * 64 + i * 32, lane);
            bh[i] = kmajor_frag(S[1], wn * 64 + i * 32, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        const _Float16 sc = (_Float16)(1.f / 2048.f);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bm[i] = kmajor_frag(S[1] + PLANE, wn * 64 + i * 32, lane);
            ah[i] = ah[i] * sc;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bm[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            am[i] = kmajor_frag(S[0] + PLANE, wm * 64 + i * 32, lane);
            bh[i] = bh[i] * sc;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[i], bh[j], acc[i][j], 0, 0, 0);
    };
    // fast_tag: bit 1 = the A tile is loaded with unguarded 16-byte loads, bit 0 = the B tile (compile-time constants: a
    // runtime flag would put a branch around the loads and cost a vmcnt(0) per step).  An edge column tile (n = 300: the
    // text-literal panel) keeps its other operand -- and every whole tile both -- on 16-byte loads this way.
    // Every step is the same code; past the last tile the staged / fetched tiles are duplicates of it (in bounds, never
    // multiplied).
    auto pipeline = [&](auto fast_tag) {
        constexpr int FAST = decltype(fast_tag)::value;
        auto fetch = [&](long tile_k) {
            const long k = k_lo + tile_k * BK + kr;
            load8(la, g.a, g.lda, m0, g.m, k, (FAST & 2) != 0);
            load8(lb, g.b, g.ldb, n0, g.n, k, (FAST & 1) != 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        const long count = (k_hi - k_lo + BK - 1) / BK;
        fetch(0);
        stage(smem[0]);
        fetch(min(1L, count - 1));
        for (long it = 0; it < count; ++it) {
            __syncthreads();
            stage(smem[(it + 1) & 1]);
            fetch(min(it + 2, count - 1));
            mma(smem[it & 1]);
        }
    };
    const bool a_fast = g.a_aligned && m0 + BM <= g.m, b_fast = g.b_aligned && n0 + BN <= g.n;
    if (a_fast && b_fast) pipeline(std::integral_constant<int, 3>{});
    else if (a_fast) pipeline(std::integral_constant<int, 2>{});
    else if (b_fast) pipeline(std::integral_constant<int, 1>{});
    else pipeline(std::integral_constant<int, 0>{});

    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const bool atomic_out = g.k_splits > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = n0 + wn * 64 + j * 32 + (lane & 31);
            const long row0 = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
            if (col >= g.n) continue;
            const int ecol = scale_exponent(g.b_colmax[col]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + (r & 3) + 8 * (r >> 2);
                if (row >= g.m) continue;
                const float v = ldexpf(acc[i][j][r], -(scale_exponent(g.a_colmax[row]) + ecol));
                if (atomic_out) atomicAdd(g.c + row * g.ldc + col, v);
                else g.c[row * g.ldc + col] = v;
            }
        }
}

// out[c] = max_r |x[r, c]|   (out is overwritten; non-negative floats order like their int bits)
__global__ __launch_bounds__(256) void col_absmax_kernel(long n, int d, const float *__restrict__ x, long ldx,
                                                          int *__restrict__ out, long rows_per_block) {
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;   // four rows in flight per thread
        long r = r0;
        for (; r + 4 <= r1; r += 4) {
            m0 = fmaxf(m0, fabsf(x[r * ldx + c]));
            m1 = fmaxf(m1, fabsf(x[(r + 1) * ldx + c]));
            m2 = fmaxf(m2, fabsf(x[(r + 2) * ldx + c]));
            m3 = fmaxf(m3, fabsf(x[(r + 3) * ldx + c]));
        }
        for (; r < r1; ++r) m0 = fmaxf(m0, fabsf(x[r * ldx + c]));
        atomicMax(out + c, __float_as_int(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3))));
    }
}

}  // namespace

extern "C" int lkg_col_absmax_f32(int64_t n, int32_t d, const float *x, int64_t ldx, float *out, void *stream) {
    LKG_REQUIRE(n >= 0 && d > 0 && ldx >= d && out, "lkg_col_absmax_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * d, s) != hipSuccess) {
        lkg_set_error("lkg_col_absmax_f32: hipMemsetAsync failed");
        return LKG_ERR_HIP;
    }
    if (n == 0) return LKG_OK;
    LKG_REQUIRE(x, "lkg_col_absmax_f32: null pointer");
    const long blocks = std::min<int64_t>((n + 63) / 64, 2048);
    const long rpb = (n + blocks - 1) / blocks;
    hipLaunchKernelGGL(col_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (long)n, d, x, (long)ldx,
                       reinterpret_cast<int *>(out), rpb);
    LKG_CHECK_LAUNCH("lkg_col_absmax_f32");
    return LKG_OK;
}

extern "C" int lkg_gemm_longk_ok(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb);
static int longk_launch(bool f16, int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *a_colmax,
                        const float *b, int64_t ldb, const float *b_colmax, float *c, int64_t ldc, hipStream_t s);

extern "C" int lkg_gemm_wgrad_f32(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *a_colmax,
                                  const float *b, int64_t ldb, const float *b_colmax, float *c, int64_t ldc,
                                  void *stream) {
    LKG_REQUIRE(m >= 0 && n >= 0 && k >= 0, "lkg_gemm_wgrad_f32: negative size");
    if (m == 0 || n == 0) return LKG_OK;
    LKG_REQUIRE(c && ldc >= n, "lkg_gemm_wgrad_f32: bad C (ldc=%lld, n=%lld)", (long long)ldc, (long long)n);
    if (a && b && a_colmax && b_colmax && lda >= m && ldb >= n && lkg_gemm_longk_ok(m, n, k, a, lda, b, ldb))
        return longk_launch(true, m, n, k, a, lda, a_colmax, b, ldb, b_colmax, c, ldc, (hipStream_t)stream);   // 256 x 128 tiles, 3 tiles in flight
    hipStream_t s = (hipStream_t)stream;
    WgArgs g{};
    g.a = a; g.b = b; g.lda = lda; g.ldb = ldb; g.a_colmax = a_colmax; g.b_colmax = b_colmax; g.c = c; g.ldc = ldc;
    g.m = m; g.n = n; g.k = k;
    g.tiles_m = (int)((m + BM - 1) / BM);
    g.tiles_n = (int)((n + BN - 1) / BN);
    const long tiles = (long)g.tiles_m * g.tiles_n;
    LKG_REQUIRE(tiles < INT32_MAX, "lkg_gemm_wgrad_f32: too many tiles");
    int splits = 1;
    if (tiles < 256 && k >= 8192) splits = (int)std::max<long>(1, std::min<long>(std::min<long>(1024 / tiles, k / 2048), 65535));
    g.k_splits = splits;
    if (splits > 1 || k == 0) {       // the partial sums meet in C by f32 atomics
        const hipError_t rc = ldc == n ? hipMemsetAsync(c, 0, sizeof(float) * m * n, s)
                                       : hipMemset2DAsync(c, sizeof(float) * ldc, 0, sizeof(float) * n, m, s);
        if (rc != hipSuccess) {
            lkg_set_error("lkg_gemm_wgrad_f32: hipMemsetAsync failed");
            return LKG_ERR_HIP;
        }
        if (k == 0) return LKG_OK;
    }
    LKG_REQUIRE(a && b && a_colmax && b_colmax && lda >= m && ldb >= n, "lkg_gemm_wgrad_f32: null operand / leading dimension too small");
    // unguarded 16-byte loads need aligned rows and whole k tiles in every split (whole column tiles: decided per tile)
    g.a_aligned = k % BK == 0 && lda % 4 == 0 && lkg_aligned16(a);
    g.b_aligned = k % BK == 0 && ldb % 4 == 0 && lkg_aligned16(b);
    const long groups = ((long)splits + 7) / 8;          // (split, tile) pairs padded to whole rounds of the 8 XCDs
    LKG_REQUIRE(groups * tiles * 8 < INT32_MAX, "lkg_gemm_wgrad_f32: grid too large");
    hipLaunchKernelGGL(wgrad_f16x2_kernel, dim3((unsigned)(groups * tiles * 8)), dim3(256), 0, s, g);
    LKG_CHECK_LAUNCH("lkg_gemm_wgrad_f32");
    return LKG_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// The same product without scales: "bf16 x 3" (every f32 is the exact sum of three bf16, six bf16 MFMAs per product:
// lkg_gemm.hip) on a 256 x 128 tile per CU fed by an LDS-DMA ring.
//
// The 128 x 128 long-k engines request every operand once per tile of the other (4 GB for a 256 x 256 gradient over
// 1 M rows) and run at 4.6-5.4 TB/s of requests with the matrix pipe a quarter busy.  Here one 8-wave workgroup owns a
// CU: 256 columns of A x 128 of B (25-45 % fewer requested bytes), THREE register sets of raw pieces keep three tiles
// (72 KB per CU) in flight, and the thread that loaded a piece splits it three ways and writes the bf16 planes in the
// shadow of the step's 24 MFMAs: the split is cut into 21 slices of <= 5 VALU instructions, one behind each MFMA (an MFMA
// holds the issue port for 8 of its 32 cycles), the 22nd re-issues the set's loads.  (An LDS-DMA ring instead of the
// register sets was measured first: the three DMA issues per wave and step cost more than they saved -- 0.99 ms.)
// LDS: planes 2 x 36 KB.  k must be a multiple of 16 (the caller falls back otherwise).
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RM = 256, RN = 128, RK = 16;
constexpr int RA_PLANE = RM * RK, RB_PLANE = RN * RK;                 // bf16 elements per plane
constexpr int RBUF = 3 * (RA_PLANE + RB_PLANE);                       // bf16 elements per plane buffer (36 KB)

template <int COLS>
__device__ __forceinline__ int roff(int k, int c) {   // element (k, c) of a [16 k][COLS] plane: 64-byte chunks XORed with k & 3
    return k * COLS + ((((c >> 5) ^ (k & 3)) << 5) | (c & 31));
}
template <int COLS>
__device__ __forceinline__ bf16x8 rfrag(const __bf16 *plane, int c0, int lane) {
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int col = c0 + (lane & 16) + 4 * pp;
    const int k0 = 8 * (lane >> 5) + q;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(plane + roff<COLS>(k0, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(plane + roff<COLS>(k0 + 4, col)));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_longk_kernel(WgArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char longk_smem[];
    __bf16 *planes = reinterpret_cast<__bf16 *>(longk_smem);                      // [2][A hi mid lo | B hi mid lo]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;                      // all tiles of a k split on one XCD (above)
    const int split = (slot / tiles) * 8 + xcd, tile = slot % tiles;
    if (split >= g.k_splits) return;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const long m0 = (long)tm * RM, n0 = (long)tn * RN;
    const long per = ((g.k + g.k_splits - 1) / g.k_splits + RK - 1) / RK * RK;
    const long k_lo = (long)split * per, k_hi = min(g.k, k_lo + per);            // (whole k tiles: k % 16 == 0)
    if (k_lo >= k_hi) return;
    const int count = (int)((k_hi - k_lo) / RK);

    // this thread's pieces (4 consecutive columns of one k row): A pieces t and t + 512 (k rows t >> 6 and 8 + (t >> 6),
    // columns 4 (t & 63) ..), B piece t (k row t >> 5, columns 4 (t & 31) ..).  A piece past the operand's width is
    // requested from its last valid piece instead and zeroed (bit mask) when it is split.
    const int ka = t >> 6, ca = 4 * (t & 63), kb = t >> 5, cb = 4 * (t & 31);
    const int keep_a = (m0 + ca + 3 < g.m) ? -1 : 0, keep_b = (n0 + cb + 3 < g.n) ? -1 : 0;
    const float *pa0 = g.a + (k_lo + ka) * g.lda + min(m0 + ca, g.m - 4);
    const float *pa1 = pa0 + 8 * g.lda;
    const float *pb = g.b + (k_lo + kb) * g.ldb + min(n0 + cb, g.n - 4);
    const long step_a = RK * g.lda, step_b = RK * g.ldb;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // THREE register sets of raw pieces: the loads of tile t+4 are issued when the split of tile t+1 has freed its set --
    // three tiles (72 KB per CU) in flight, counted vmcnt waits by the compiler (plain loads only)
    f32x4 rs0[3], rs1[3], rs2[3];
    int fetched = 0;                             // k tiles requested so far (past the last one the last is requested again)
    auto fetch = [&](f32x4 (&r)[3]) {
        r[0] = *reinterpret_cast<const f32x4 *>(pa0);
        r[1] = *reinterpret_cast<const f32x4 *>(pa1);
        r[2] = *reinterpret_cast<const f32x4 *>(pb);
        ++fetched;
        const bool more = fetched < count;       // (wave-uniform)
        pa0 += more ? step_a : 0;
        pa1 += more ? step_a : 0;
        pb += more ? step_b : 0;
    };

    // ---- the split of one element pair, in three slices of <= 5 VALU instructions (x -> hi, mid, lo; packed conversions)
    f32x2 px[6];                                  // pair q: piece q >> 1, elements 2 (q & 1), 2 (q & 1) + 1
    bf16x2 ph[6], pm[6], pl[6];
    auto s1 = [&](const f32x4 (&r)[3], int q, int live) {
        const f32x4 v = r[q >> 1];
        const int keep = ((q >> 1) == 2 ? keep_b : keep_a) & live;
        px[q][0] = __int_as_float(__float_as_int(v[2 * (q & 1)]) & keep);
        px[q][1] = __int_as_float(__float_as_int(v[2 * (q & 1) + 1]) & keep);
        ph[q] = __builtin_convertvector(px[q], bf16x2);
    };
    auto s2 = [&](int q) {
        px[q] = px[q] - __builtin_convertvector(ph[q], f32x2);
        pm[q] = __builtin_convertvector(px[q], bf16x2);
    };
    auto s3 = [&](int q) {
        px[q] = px[q] - __builtin_convertvector(pm[q], f32x2);
        pl[q] = __builtin_convertvector(px[q], bf16x2);
    };
    const int off_a0 = roff<RM>(ka, ca), off_a1 = roff<RM>(ka + 8, ca), off_b = roff<RN>(kb, cb);
    auto write_piece = [&](int p, __bf16 *D) {    // piece p of the next tile -> its three planes
        __bf16 *plane0 = p == 2 ? D + 3 * RA_PLANE : D;
        const int pe = p == 2 ? RB_PLANE : RA_PLANE, off = p == 0 ? off_a0 : (p == 1 ? off_a1 : off_b);
        const bf16x4 h = {ph[2 * p][0], ph[2 * p][1], ph[2 * p + 1][0], ph[2 * p + 1][1]};
        const bf16x4 m = {pm[2 * p][0], pm[2 * p][1], pm[2 * p + 1][0], pm[2 * p + 1][1]};
        const bf16x4 l = {pl[2 * p][0], pl[2 * p][1], pl[2 * p + 1][0], pl[2 * p + 1][1]};
        *reinterpret_cast<bf16x4 *>(plane0 + off) = h;
        *reinterpret_cast<bf16x4 *>(plane0 + pe + off) = m;
        *reinterpret_cast<bf16x4 *>(plane0 + 2 * pe + off) = l;
    };
    // slice number `i` of the 22 that follow the MFMAs (pairs in order; a piece is written once both its pairs are split;
    // the last one re-issues the set's loads)
    auto slice = [&](f32x4 (&r)[3], int i, __bf16 *D, int live) {
        if (i == 21) { fetch(r); return; }
        if (i > 21) return;
        const int p = i / 7, rr = i % 7;           // per piece: s1 s1 s2 s2 s3 s3 write
        if (rr < 2) s1(r, 2 * p + rr, live);
        else if (rr < 4) s2(2 * p + rr - 2);
        else if (rr < 6) s3(2 * p + rr - 4);
        else write_piece(p, D);
    };
#define LKG_PIN() __builtin_amdgcn_sched_barrier(0)
#define LKG_MF(I, J, PA, PB) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[I][PA], b[J][PB], acc[I][J], 0, 0, 0)
#define LKG_TERM4(PA, PB, S0)                                                                                       \
    LKG_MF(0, 0, PA, PB); slice(r, S0, D, live); LKG_PIN(); LKG_MF(0, 1, PA, PB); slice(r, S0 + 1, D, live); LKG_PIN();  \
    LKG_MF(1, 0, PA, PB); slice(r, S0 + 2, D, live); LKG_PIN(); LKG_MF(1, 1, PA, PB); slice(r, S0 + 3, D, live); LKG_PIN();
    // the 24 MFMAs of the tile in S with the split of the next tile (register set r -> D; live = 0: past the last tile,
    // zero planes) sliced between them
    auto step = [&](const __bf16 *S, __bf16 *D, f32x4 (&r)[3], int live) {
        const __bf16 *SA = S, *SB = S + 3 * RA_PLANE;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i][2] = rfrag<RM>(SA + 2 * RA_PLANE, wm * 64 + i * 32, lane);
            b[i][0] = rfrag<RN>(SB, wn * 64 + i * 32, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i][0] = rfrag<RM>(SA, wm * 64 + i * 32, lane);
            b[i][2] = rfrag<RN>(SB + 2 * RB_PLANE, wn * 64 + i * 32, lane);
            a[i][1] = rfrag<RM>(SA + RA_PLANE, wm * 64 + i * 32, lane);
            b[i][1] = rfrag<RN>(SB + RB_PLANE, wn * 64 + i * 32, lane);
        }
        LKG_PIN();
        LKG_TERM4(2, 0, 0)
        LKG_TERM4(0, 2, 4)
        LKG_TERM4(1, 1, 8)
        LKG_TERM4(1, 0, 12)
        LKG_TERM4(0, 1, 16)
        LKG_TERM4(0, 0, 20)
    };
#undef LKG_TERM4
#undef LKG_MF
#undef LKG_PIN

    // Tile j lives in register set j % 3.  Step t: barrier (tile t complete in planes t & 1) -> MFMAs of tile t with the
    // split of tile t+1 and, behind it, the loads of tile t+4 into the same set.  The loop runs in threes (static set
    // names); the up to two extra steps multiply planes that were staged as zeros (live = 0).
    fetch(rs0);
    fetch(rs1);
    fetch(rs2);
    for (int i = 0; i < 21; ++i) slice(rs0, i, planes, -1);
    fetch(rs0);
    auto P = [&](int j) { return planes + (j & 1) * RBUF; };
    for (int it = 0; it < count; it += 3) {
        __syncthreads();
        step(P(it), P(it + 1), rs1, it + 1 < count ? -1 : 0);
        __syncthreads();
        step(P(it + 1), P(it + 2), rs2, it + 2 < count ? -1 : 0);
        __syncthreads();
        step(P(it + 2), P(it + 3), rs0, it + 3 < count ? -1 : 0);
    }

    const bool atomic_out = g.k_splits > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = n0 + wn * 64 + j * 32 + (lane & 31);
            const long row0 = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
            if (col >= g.n) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + (r & 3) + 8 * (r >> 2);
                if (row >= g.m) continue;
                if (atomic_out) atomicAdd(g.c + row * g.ldc + col, acc[i][j][r]);
                else g.c[row * g.ldc + col] = acc[i][j][r];
            }
        }
}

}  // namespace

// The f16 x 2 arithmetic of wgrad_f16x2_kernel (column scales, exact hi / mid split, 12 MFMAs per step instead of 24) in the
// structure of wgrad_longk_kernel: with half the MFMAs the step is bound by its 24 KB of loads, three tiles in flight.
namespace {
constexpr int HBUF = 2 * (RA_PLANE + RB_PLANE);                        // fp16 elements per plane buffer (24 KB)

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_longk_f16_kernel(WgArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char longk_smem[];
    _Float16 *planes = reinterpret_cast<_Float16 *>(longk_smem);                 // [2][A hi mid | B hi mid]
    typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = (slot / tiles) * 8 + xcd, tile = slot % tiles;
    if (split >= g.k_splits) return;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const long m0 = (long)tm * RM, n0 = (long)tn * RN;
    const long per = ((g.k + g.k_splits - 1) / g.k_splits + RK - 1) / RK * RK;
    const long k_lo = (long)split * per, k_hi = min(g.k, k_lo + per);
    if (k_lo >= k_hi) return;
    const int count = (int)((k_hi - k_lo) / RK);
    // A tile at the edge of C whose real rows / columns fill at most half of it (600 x 300 in 256 x 128 tiles: 88 of 256, 44 of 128):
    // the waves whose 64 x 64 block is all padding run no MFMAs.  The 4 x 2 wave grid is laid out so that the live waves sit one
    // per SIMD (waves 0..3): by rows (wm = wave >> 1) when the rows are short, by columns (wn = wave >> 2) when the columns are.
    const int m_live = (int)min(4l, (g.m - m0 + 63) / 64), n_live = (int)min(2l, (g.n - n0 + 63) / 64);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool by_cols = n_live == 1;
    const int wm = by_cols ? (wave & 3) : (wave >> 1), wn = by_cols ? (wave >> 2) : (wave & 1);
    const bool wave_live = (by_cols ? (wave_u & 3) : (wave_u >> 1)) < m_live && (by_cols ? (wave_u >> 2) : (wave_u & 1)) < n_live;

    const int ka = t >> 6, ca = 4 * (t & 63), kb = t >> 5, cb = 4 * (t & 31);
    const int keep_a = (m0 + ca + 3 < g.m) ? -1 : 0, keep_b = (n0 + cb + 3 < g.n) ? -1 : 0;
    const float *pa0 = g.a + (k_lo + ka) * g.lda + min(m0 + ca, g.m - 4);
    const float *pa1 = pa0 + 8 * g.lda;
    const float *pb = g.b + (k_lo + kb) * g.ldb + min(n0 + cb, g.n - 4);
    const long step_a = RK * g.lda, step_b = RK * g.ldb;
    float sa[4], sb[4];                           // 2^e of this thread's columns (both A pieces share theirs)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sa[j] = ldexpf(1.f, scale_exponent(g.a_colmax[min(m0 + ca + j, g.m - 1)]));
        sb[j] = ldexpf(1.f, scale_exponent(g.b_colmax[min(n0 + cb + j, g.n - 1)]));
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 rs0[3], rs1[3], rs2[3];
    int fetched = 0;
    auto fetch = [&](f32x4 (&r)[3]) {
        r[0] = *reinterpret_cast<const f32x4 *>(pa0);
        r[1] = *reinterpret_cast<const f32x4 *>(pa1);
        r[2] = *reinterpret_cast<const f32x4 *>(pb);
        ++fetched;
        const bool more = fetched < count;
        pa0 += more ? step_a : 0;
        pa1 += more ? step_a : 0;
        pb += more ? step_b : 0;
    };
    float px[6][2];
    fp16x2 ph[6], pm[6];
    auto t1 = [&](const f32x4 (&r)[3], int q, int live) {      // pair q: piece q >> 1, elements 2 (q & 1), 2 (q & 1) + 1
        const f32x4 v = r[q >> 1];
        const bool isb = (q >> 1) == 2;
        const int keep = (isb ? keep_b : keep_a) & live, e0 = 2 * (q & 1);
        px[q][0] = __int_as_float(__float_as_int(v[e0]) & keep) * (isb ? sb[e0] : sa[e0]);
        px[q][1] = __int_as_float(__float_as_int(v[e0 + 1]) & keep) * (isb ? sb[e0 + 1] : sa[e0 + 1]);
        ph[q] = __builtin_amdgcn_cvt_pkrtz(px[q][0], px[q][1]);
    };
    auto t2 = [&](int q) {
        pm[q] = __builtin_amdgcn_cvt_pkrtz((px[q][0] - (float)ph[q][0]) * 2048.f, (px[q][1] - (float)ph[q][1]) * 2048.f);
    };
    const int off_a0 = roff<RM>(ka, ca), off_a1 = roff<RM>(ka + 8, ca), off_b = roff<RN>(kb, cb);
    auto write_piece = [&](int p, _Float16 *D) {
        _Float16 *plane0 = p == 2 ? D + 2 * RA_PLANE : D;
        const int pe = p == 2 ? RB_PLANE : RA_PLANE, off = p == 0 ? off_a0 : (p == 1 ? off_a1 : off_b);
        const fp16x4 h = {ph[2 * p][0], ph[2 * p][1], ph[2 * p + 1][0], ph[2 * p + 1][1]};
        const fp16x4 m = {pm[2 * p][0], pm[2 * p][1], pm[2 * p + 1][0], pm[2 * p + 1][1]};
        *reinterpret_cast<fp16x4 *>(plane0 + off) = h;
        *reinterpret_cast<fp16x4 *>(plane0 + pe + off) = m;
    };
#define LKG_PIN() __builtin_amdgcn_sched_barrier(0)
#define LKG_MF(C, A_, B_) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C, 0, 0, 0)
    auto hfrag = [&](const _Float16 *plane, int cols, int c0) {
        return cols == RM ? __builtin_bit_cast(f16x8, rfrag<RM>(reinterpret_cast<const __bf16 *>(plane), c0, lane))
                          : __builtin_bit_cast(f16x8, rfrag<RN>(reinterpret_cast<const __bf16 *>(plane), c0, lane));
    };
    // the 12 MFMAs of the tile in S with the split of the next tile (register set r -> D) and the 2^-11 copies of the hi
    // fragments sliced between them
    auto step = [&](const _Float16 *S, _Float16 *D, f32x4 (&r)[3], int live, auto LIVE) {
        if constexpr (!decltype(LIVE)::value) {       // a wave of padding: its share of the split, nothing else
#pragma unroll
            for (int q = 0; q < 6; ++q) { t1(r, q, live); t2(q); }
#pragma unroll
            for (int p = 0; p < 3; ++p) write_piece(p, D);
            fetch(r);
            return;
        }
        const _Float16 *SA = S, *SB = S + 2 * RA_PLANE;
        f16x8 ah[2], am[2], bh[2], bm[2], ahs[2], bhs[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = hfrag(SA, RM, wm * 64 + i * 32);
            bh[i] = hfrag(SB, RN, wn * 64 + i * 32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bm[i] = hfrag(SB + RB_PLANE, RN, wn * 64 + i * 32);
            am[i] = hfrag(SA + RA_PLANE, RM, wm * 64 + i * 32);
        }
        const _Float16 sc = (_Float16)(1.f / 2048.f);
        LKG_PIN();
        LKG_MF(acc[0][0], ah[0], bh[0]); ahs[0] = ah[0] * sc; t1(r, 0, live); LKG_PIN();
        LKG_MF(acc[0][1], ah[0], bh[1]); ahs[1] = ah[1] * sc; t1(r, 1, live); LKG_PIN();
        LKG_MF(acc[1][0], ah[1], bh[0]); t1(r, 2, live); t1(r, 3, live); LKG_PIN();
        LKG_MF(acc[1][1], ah[1], bh[1]); t1(r, 4, live); t1(r, 5, live); LKG_PIN();
        LKG_MF(acc[0][0], ahs[0], bm[0]); bhs[0] = bh[0] * sc; t2(0); LKG_PIN();
        LKG_MF(acc[0][1], ahs[0], bm[1]); bhs[1] = bh[1] * sc; t2(1); LKG_PIN();
        LKG_MF(acc[1][0], ahs[1], bm[0]); t2(2); write_piece(0, D); LKG_PIN();
        LKG_MF(acc[1][1], ahs[1], bm[1]); t2(3); LKG_PIN();
        LKG_MF(acc[0][0], am[0], bhs[0]); t2(4); write_piece(1, D); LKG_PIN();
        LKG_MF(acc[0][1], am[0], bhs[1]); t2(5); LKG_PIN();
        LKG_MF(acc[1][0], am[1], bhs[0]); write_piece(2, D); LKG_PIN();
        LKG_MF(acc[1][1], am[1], bhs[1]); fetch(r);
    };
#undef LKG_MF
#undef LKG_PIN

    fetch(rs0);
    fetch(rs1);
    fetch(rs2);
    for (int q = 0; q < 6; ++q) { t1(rs0, q, -1); t2(q); }
    for (int p = 0; p < 3; ++p) write_piece(p, planes);
    fetch(rs0);
    auto P = [&](int j) { return planes + (j & 1) * HBUF; };
    auto k_loop = [&](auto LIVE) {                     // (the whole loop twice: no branch between the accumulators' uses)
        for (int it = 0; it < count; it += 3) {
            __syncthreads();
            step(P(it), P(it + 1), rs1, it + 1 < count ? -1 : 0, LIVE);
            __syncthreads();
            step(P(it + 1), P(it + 2), rs2, it + 2 < count ? -1 : 0, LIVE);
            __syncthreads();
            step(P(it + 2), P(it + 3), rs0, it + 3 < count ? -1 : 0, LIVE);
        }
    };
    if (wave_live) k_loop(std::true_type{});
    else { k_loop(std::false_type{}); return; }        // (nothing to store)

    const bool atomic_out = g.k_splits > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = n0 + wn * 64 + j * 32 + (lane & 31);
            const long row0 = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
            if (col >= g.n) continue;
            const int ecol = scale_exponent(g.b_colmax[col]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + (r & 3) + 8 * (r >> 2);
                if (row >= g.m) continue;
                const float v = ldexpf(acc[i][j][r], -(scale_exponent(g.a_colmax[row]) + ecol));
                if (atomic_out) atomicAdd(g.c + row * g.ldc + col, v);
                else g.c[row * g.ldc + col] = v;
            }
        }
}
}  // namespace

// 1 when lkg_gemm_longk_f32 takes this product (else the caller uses lkg_gemm_f32): long k in whole 16-row tiles,
// 16-byte aligned rows, widths that are multiples of 4
extern "C" int lkg_gemm_longk_ok(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb) {
    return k >= 8192 && k % RK == 0 && m >= 4 && n >= 4 && m % 4 == 0 && n % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
           lkg_aligned16(a) && lkg_aligned16(b);
}

static int longk_launch(bool f16, int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *a_colmax,
                        const float *b, int64_t ldb, const float *b_colmax, float *c, int64_t ldc, hipStream_t s) {
    WgArgs g{};
    g.a = a; g.b = b; g.lda = lda; g.ldb = ldb; g.a_colmax = a_colmax; g.b_colmax = b_colmax; g.c = c; g.ldc = ldc;
    g.m = m; g.n = n; g.k = k;
    g.tiles_m = (int)((m + RM - 1) / RM);
    g.tiles_n = (int)((n + RN - 1) / RN);
    const long tiles = (long)g.tiles_m * g.tiles_n;
    LKG_REQUIRE(tiles < 65536, "lkg_gemm_longk_f32: too many tiles");
    // one workgroup owns a CU: about two rounds of the 256 CUs, at least 64 steps per workgroup
    int splits = (int)std::max<long>(1, std::min<long>((512 + tiles - 1) / tiles, k / 1024));
    g.k_splits = splits;
    if (splits > 1) {
        const hipError_t rc = ldc == n ? hipMemsetAsync(c, 0, sizeof(float) * m * n, s)
                                       : hipMemset2DAsync(c, sizeof(float) * ldc, 0, sizeof(float) * n, m, s);
        if (rc != hipSuccess) {
            lkg_set_error("lkg_gemm_longk_f32: hipMemsetAsync failed");
            return LKG_ERR_HIP;
        }
    }
    const int lds = (f16 ? HBUF : RBUF) * 2 * 2;
    static bool raised[2] = {false, false};
    if (!raised[f16]) {
        const void *fn = f16 ? reinterpret_cast<const void *>(wgrad_longk_f16_kernel) : reinterpret_cast<const void *>(wgrad_longk_kernel);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            lkg_set_error("lkg_gemm_longk_f32: cannot raise the dynamic LDS limit");
            return LKG_ERR_HIP;
        }
        raised[f16] = true;
    }
    const long groups = ((long)splits + 7) / 8;
    if (f16) hipLaunchKernelGGL(wgrad_longk_f16_kernel, dim3((unsigned)(groups * tiles * 8)), dim3(512), lds, s, g);
    else hipLaunchKernelGGL(wgrad_longk_kernel, dim3((unsigned)(groups * tiles * 8)), dim3(512), lds, s, g);
    LKG_CHECK_LAUNCH("lkg_gemm_longk_f32");
    return LKG_OK;
}

extern "C" int lkg_gemm_longk_f32(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb,
                                  float *c, int64_t ldc, void *stream) {
    LKG_REQUIRE(m > 0 && n > 0 && k > 0 && a && b && c && ldc >= n && lda >= m && ldb >= n, "lkg_gemm_longk_f32: bad arguments");
    LKG_REQUIRE(lkg_gemm_longk_ok(m, n, k, a, lda, b, ldb), "lkg_gemm_longk_f32: needs k >= 8192 in whole 16-row tiles, widths and "
                "row strides that are multiples of 4 floats and 16-byte aligned operands (lkg_gemm_longk_ok)");
    return longk_launch(false, m, n, k, a, lda, nullptr, b, ldb, nullptr, c, ldc, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradients with a NARROW dY: C[m, n] = sum over the k rows of A[k, m] * B[k, n] for m <= 64 (an aggregation
// layer of conv_dim 32 -- the reference's default is eight of them, argument_pretraining.py:54-58 -- has dW = dY^T X
// with dY 32 wide).  The matrix-core engines above spend a 256 x 128 (or 128 x 128) tile on it: 0.7 ms for a 32 x 32
// product over 1 M rows that moves 256 MB.  Here the product is plain f32 FMAs on the VALU: a workgroup walks its
// slice of the k rows in tiles of 32 rows staged in LDS, every thread keeps an (MT / 16) x 4 block of the MT x 64
// output chunk in registers (two LDS reads per row and thread, 16-byte, conflict-free), the slices are combined by f32
// atomics into the zeroed output.  n is covered in chunks of 64 columns (grid.y).  Exact f32 products, f32 sums.
namespace {

template <int MT>
__global__ __launch_bounds__(256) void smallm_wgrad_kernel(long m, long n, long k, const float *__restrict__ a, long lda,
                                                            const float *__restrict__ b, long ldb, float *__restrict__ c,
                                                            long ldc, long rows_per_block) {
    constexpr int RM_ = MT / 16;                   // output rows per thread (2 or 4)
    constexpr int R = 32;                          // k rows per staged tile
    __shared__ __attribute__((aligned(16))) float as[R][MT], bs[R][64];
    const int t = threadIdx.x;
    const int bi = t >> 4, bj = t & 15;            // output block: rows RM_ * bi .., columns 4 * bj .. of this chunk
    const long n0 = (long)blockIdx.y * 64;
    const long k_lo = (long)blockIdx.x * rows_per_block, k_hi = min(k, k_lo + rows_per_block);
    if (k_lo >= k_hi) return;
    float acc[RM_][4];
#pragma unroll
    for (int i = 0; i < RM_; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // staging: the A tile is R x MT floats = R * MT / 4 16-byte pieces (1 or 2 per thread), the B tile R x 64 = 512 pieces
    // (2 per thread).  The pieces of tile t+1 are loaded into registers BEFORE tile t is multiplied and written to LDS
    // after it: the global round trip runs behind the FMAs.
    constexpr int APIECES = R * MT / 4, A_PER_ROW = MT / 4, NA = APIECES / 256, NB = 2;
    float4 ra[NA], rb[NB];
    auto load_tile = [&](long k0) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int p = t + 256 * q, r = p / A_PER_ROW, cq = (p % A_PER_ROW) * 4;
            const long row = min(k0 + r, k_hi - 1);
            const bool on = k0 + r < k_hi && cq < m;               // (m, n are multiples of 4: a piece is in or out)
            const float4 v = *reinterpret_cast<const float4 *>(a + row * lda + (cq < m ? cq : 0));
            ra[q] = on ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int p = t + 256 * q, r = p >> 4, cq = (p & 15) * 4;
            const long row = min(k0 + r, k_hi - 1);
            const bool on = k0 + r < k_hi && n0 + cq < n;
            const float4 v = *reinterpret_cast<const float4 *>(b + row * ldb + (n0 + cq < n ? n0 + cq : 0));
            rb[q] = on ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int p = t + 256 * q;
            *reinterpret_cast<float4 *>(&as[p / A_PER_ROW][(p % A_PER_ROW) * 4]) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int p = t + 256 * q;
            *reinterpret_cast<float4 *>(&bs[p >> 4][(p & 15) * 4]) = rb[q];
        }
    };
    load_tile(k_lo);
    store_tile();
    __syncthreads();
    for (long k0 = k_lo; k0 < k_hi; k0 += R) {
        const bool more = k0 + R < k_hi;           // (workgroup-uniform)
        if (more) load_tile(k0 + R);
#pragma unroll 8
        for (int r = 0; r < R; ++r) {
            const float4 bv = *reinterpret_cast<const float4 *>(&bs[r][4 * bj]);
            float av[RM_];
            if constexpr (RM_ == 4) {
                const float4 x = *reinterpret_cast<const float4 *>(&as[r][4 * bi]);
                av[0] = x.x; av[1] = x.y; av[2] = x.z; av[3] = x.w;
            } else {
                const float2 x = *reinterpret_cast<const float2 *>(&as[r][2 * bi]);
                av[0] = x.x; av[1] = x.y;
            }
#pragma unroll
            for (int i = 0; i < RM_; ++i) {
                acc[i][0] = fmaf(av[i], bv.x, acc[i][0]);
                acc[i][1] = fmaf(av[i], bv.y, acc[i][1]);
                acc[i][2] = fmaf(av[i], bv.z, acc[i][2]);
                acc[i][3] = fmaf(av[i], bv.w, acc[i][3]);
            }
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RM_; ++i) {
        const long row = RM_ * bi + i;
        if (row >= m) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long col = n0 + 4 * bj + j;
            if (col < n) atomicAdd(c + row * ldc + col, acc[i][j]);
        }
    }
}

}  // namespace

// 1 when lkg_gemm_smallm_f32 takes this product: a narrow A (m <= 64) over many rows, widths and row strides multiples of
// 4 floats, 16-byte aligned operands
extern "C" int lkg_gemm_smallm_ok(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb) {
    return k >= 4096 && m >= 4 && m <= 64 && n >= 4 && m % 4 == 0 && n % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
           lkg_aligned16(a) && lkg_aligned16(b);
}

extern "C" int lkg_gemm_smallm_f32(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb,
                                   float *c, int64_t ldc, void *stream) {
    LKG_REQUIRE(m > 0 && n > 0 && k > 0 && a && b && c && ldc >= n && lda >= m && ldb >= n, "lkg_gemm_smallm_f32: bad arguments");
    LKG_REQUIRE(lkg_gemm_smallm_ok(m, n, k, a, lda, b, ldb), "lkg_gemm_smallm_f32: needs m <= 64, k >= 4096, widths and row "
                "strides that are multiples of 4 floats and 16-byte aligned operands (lkg_gemm_smallm_ok)");
    hipStream_t s = (hipStream_t)stream;
    const hipError_t rc = ldc == n ? hipMemsetAsync(c, 0, sizeof(float) * m * n, s)
                                   : hipMemset2DAsync(c, sizeof(float) * ldc, 0, sizeof(float) * n, m, s);
    if (rc != hipSuccess) {
        lkg_set_error("lkg_gemm_smallm_f32: hipMemsetAsync failed");
        return LKG_ERR_HIP;
    }
    const int64_t chunks = (n + 63) / 64;
    // slices of whole 32-row tiles, at least 256 rows each; every slice ends in m x 64 atomics on the same few KB of C
    // (contended float atomics run an order of magnitude below the streaming rate), so fewer and longer slices than the
    // CUs could hold: LKG_SMALLM_BLOCKS (tuning aid) or 768 over the grid
    static const int64_t target = getenv("LKG_SMALLM_BLOCKS") ? atoll(getenv("LKG_SMALLM_BLOCKS")) : 768;
    int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(target / chunks + 1, k / 256));
    int64_t per = ((k + blocks - 1) / blocks + 31) / 32 * 32;
    blocks = (k + per - 1) / per;
    const dim3 grid((unsigned)blocks, (unsigned)chunks);
    if (m <= 32)
        hipLaunchKernelGGL((smallm_wgrad_kernel<32>), grid, dim3(256), 0, s, (long)m, (long)n, (long)k, a, (long)lda, b, (long)ldb,
                           c, (long)ldc, (long)per);
    else
        hipLaunchKernelGGL((smallm_wgrad_kernel<64>), grid, dim3(256), 0, s, (long)m, (long)n, (long)k, a, (long)lda, b, (long)ldb,
                           c, (long)ldc, (long)per);
    LKG_CHECK_LAUNCH("lkg_gemm_smallm_f32");
    return LKG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Skinny products over many rows: C[m, n] = A[m, k] . op(B) (+ bias) (+ beta C) with k, n <= 64 -- the 32 x 32 Linears, their
// data gradients and the residual mix of narrow aggregation layers (the reference's defaults stack eight layers of 32:
// argument.py:56-58).  1 M x 32 x 32 moves 256 MB and 2 GFLOP: the matrix-core engines spend a 128-column tile (and
// an operand split, or a row-scale pass) on it, 0.22 ms; here it is exact f32 on the VALU at the rate the rows stream:
// a persistent workgroup keeps op(B) in LDS, stages 64 rows of A per turn, every thread owns one row x n / 4 columns.
namespace {

template <int CPT>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(long m, int n, int k, const float *__restrict__ a, long lda,
                                                           const float *__restrict__ b, long ldb, int trans_b, float beta,
                                                           float *__restrict__ c, long ldc, const float *__restrict__ bias) {
    constexpr int KP = 68;                              // LDS row pitch of the A tile (k <= 64, +4: conflict-free 16-byte reads)
    __shared__ __attribute__((aligned(16))) float ws[64][4 * CPT], xs[64][KP];
    const int t = threadIdx.x;
    const int r = t >> 2, cg = t & 3;                   // this thread: row r of the tile, columns cg * CPT .. + CPT
    // op(B)[kk][col] into LDS once (zero-padded to 4 * CPT columns)
    for (int i = t; i < k * 4 * CPT; i += 256) {
        const int kk = i / (4 * CPT), col = i % (4 * CPT);
        ws[kk][col] = col < n ? (trans_b ? b[(long)col * ldb + kk] : b[(long)kk * ldb + col]) : 0.f;
    }
    float bv[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) bv[j] = (bias && cg * CPT + j < n) ? bias[cg * CPT + j] : 0.f;
    const int k4 = k >> 2;
    for (long m0 = (long)blockIdx.x * 64; m0 < m; m0 += (long)gridDim.x * 64) {
        __syncthreads();                                // (the previous tile's readers are done; B is in place on the first turn)
        for (int p = t; p < 64 * k4; p += 256) {        // 16-byte pieces of the 64 x k tile
            const int rr = p / k4, q = p % k4;
            const long row = min(m0 + rr, m - 1);
            *reinterpret_cast<float4 *>(&xs[rr][4 * q]) = *reinterpret_cast<const float4 *>(a + row * lda + 4 * q);
        }
        __syncthreads();
        float acc[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[j] = bv[j];
        for (int q = 0; q < k4; ++q) {
            const float4 xv = *reinterpret_cast<const float4 *>(&xs[r][4 * q]);
            const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < CPT; j += 4) {
                    const float4 wv = *reinterpret_cast<const float4 *>(&ws[4 * q + e][cg * CPT + j]);
                    acc[j] = fmaf(xe[e], wv.x, acc[j]);
                    acc[j + 1] = fmaf(xe[e], wv.y, acc[j + 1]);
                    acc[j + 2] = fmaf(xe[e], wv.z, acc[j + 2]);
                    acc[j + 3] = fmaf(xe[e], wv.w, acc[j + 3]);
                }
        }
        const long row = m0 + r;
        if (row < m) {
            float *dst = c + row * ldc + cg * CPT;
#pragma unroll
            for (int j = 0; j < CPT; j += 4) {
                if (cg * CPT + j >= n) break;           // (n is a multiple of 4: a group of four is in or out)
                float4 o = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
                if (beta != 0.f) {
                    const float4 old = *reinterpret_cast<const float4 *>(dst + j);
                    o.x = fmaf(beta, old.x, o.x); o.y = fmaf(beta, old.y, o.y); o.z = fmaf(beta, old.z, o.z); o.w = fmaf(beta, old.w, o.w);
                }
                *reinterpret_cast<float4 *>(dst + j) = o;
            }
        }
    }
}

}  // namespace

// 1 when lkg_gemm_skinny_f32 takes this product: many rows, k and n <= 64 in multiples of 4, 16-byte aligned rows of A and C
extern "C" int lkg_gemm_skinny_ok(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *c, int64_t ldc) {
    return m >= 4096 && k >= 4 && k <= 64 && n >= 4 && n <= 64 && k % 4 == 0 && n % 4 == 0 && lda % 4 == 0 && ldc % 4 == 0 &&
           lkg_aligned16(a) && lkg_aligned16(c);
}

extern "C" int lkg_gemm_skinny_f32(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb,
                                   int32_t trans_b, float beta, float *c, int64_t ldc, const float *bias, void *stream) {
    LKG_REQUIRE(m > 0 && a && b && c && lda >= k && ldc >= n && ldb >= (trans_b ? k : n), "lkg_gemm_skinny_f32: bad arguments");
    LKG_REQUIRE(lkg_gemm_skinny_ok(m, n, k, a, lda, c, ldc), "lkg_gemm_skinny_f32: needs m >= 4096, k and n <= 64 in multiples "
                "of 4 and 16-byte aligned rows (lkg_gemm_skinny_ok)");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)std::min<int64_t>((m + 63) / 64, 256 * 8);
#define LKG_SKINNY(CPT_)                                                                                                    \
    hipLaunchKernelGGL((skinny_gemm_kernel<CPT_>), dim3(blocks), dim3(256), 0, s, (long)m, (int)n, (int)k, a, (long)lda, b,  \
                       (long)ldb, (int)trans_b, beta, c, (long)ldc, bias)
    if (n <= 16) LKG_SKINNY(4);
    else if (n <= 32) LKG_SKINNY(8);
    else if (n <= 48) LKG_SKINNY(12);
    else LKG_SKINNY(16);
#undef LKG_SKINNY
    LKG_CHECK_LAUNCH("lkg_gemm_skinny_f32");
    return LKG_OK;
}

// lkg_preload(): HIP loads a translation unit's code object on the first use of one of its kernels; asking for a kernel's
// attributes is such a use (no launch).
int lkg_internal_preload_gemm_wgrad() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&wgrad_f16x2_kernel)) == hipSuccess ? 0 : 1;
}
