// Dense fp32 GEMM on the gfx950 matrix cores, plus the two grouped forms the TransR projection needs.
// Three engines behind one entry point (lkg_gemm_f32 picks by shape; all take and return f32, accumulate in f32):
//   f32 engine     v_mfma_f32_32x32x2_f32 (exact f32, bitwise a k-ordered fmaf chain): every shape, every mode;
//   split engine 1 v_mfma_f32_32x32x16_bf16 x 6 over a three-way bf16 split: A row-major, small pre-split B
//                  (nn.Linear forward and data gradient);
//   split engine 2 the same arithmetic, both operands k-major (weight gradients), transposing LDS reads.
// The description below is the f32 engine's; the split engines are described where they are defined.
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
//   rows mode : rows [seg[g], seg[g+1]) of A and C use B + (g % b_period) * stride_b     (projection, data gradient)
//   k mode    : the reduction runs over rows [seg[g], seg[g+1]) of A^T and B, output C + g * stride_c
//               (weight gradient  g_W[r] = X_r^T G_r)
// Plain long-K / small-output products (nn.Linear weight gradients, K = n_entities) are split over K
// with f32 atomic accumulation into a zeroed C.
#include <stdlib.h>

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
    int b_period;        // rows mode: group z uses B block z % b_period (0: block z)
    int mode;            // 0 plain, 1 rows grouped, 2 k grouped
    int k_splits;        // plain mode only (atomic accumulation when > 1)
    int tiles_m, tiles_n;
    const __bf16 *bp;    // split engine: B as bf16 planes, tile-major [tiles_n][ktiles_b][3][PLANE] (presplit_b_kernel)
    int ktiles_b;
    int split_km;        // split engine 2: both operands k-major, split on the fly
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

// ---- split-operand engine: f32 GEMM on the bf16 matrix cores ("bf16 x 3") ------------------------------
// The f32-input MFMA runs at 1/16 of the bf16 rate.  Every f32 x is the exact sum hi + mid + lo of three bf16
// numbers (8 + 8 + 8 mantissa bits, round-to-nearest at each step; the exponent range is the same), so
//   a * b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh)  + O(2^-25 |a b|)
// and six v_mfma_f32_32x32x16_bf16 (products exact, f32 accumulation) replace eight 32x32x2 f32 MFMAs per 16 k
// at 2.7x the peak rate.  The dropped terms (am*bl, al*bm, al*bl) are below f32 rounding; measured against f64
// the result is as accurate as the f32 MFMA chain (fewer accumulator roundings per output: 6 per 16 k, not 8).
// K-contiguous operands only (A row-major, B = nn.Linear weight): image [3 planes][128 rows][16 k] of bf16, one
// ds_read_b128 per lane and plane fetches the 8 consecutive k of v_mfma_f32_32x32x16_bf16 (lane l: row l & 31,
// k = 8 (l >> 5) + j).  The two 16-byte halves of a row are swapped on rows 16-31 of every 32, which makes the
// reads conflict-free for the b128 lane groups ({0-3, 12-15, 20-27}, ...) without padding: 48 KB for both
// operands, double-buffered.  The split costs ~44 VALU instructions per 8 elements at staging time, once per
// element per workgroup, hidden behind the MFMAs of the other waves.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PLANE = BM * BK;   // bf16 elements of one plane of one operand tile

__device__ __forceinline__ void split_planes(const float (&v)[EPT], bf16x8 (&pl)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        const float r1 = v[j] - (float)h;
        const __bf16 m = (__bf16)r1;
        pl[0][j] = h;
        pl[1][j] = m;
        pl[2][j] = (__bf16)(r1 - (float)m);
    }
}
__device__ __forceinline__ void planes_store(const bf16x8 (&pl)[3], __bf16 *planes, int t) {
    const int row = t >> 1;
    __bf16 *p = planes + row * BK + (((t & 1) ^ ((row >> 4) & 1)) << 3);
    *reinterpret_cast<bf16x8 *>(p) = pl[0];
    *reinterpret_cast<bf16x8 *>(p + PLANE) = pl[1];
    *reinterpret_cast<bf16x8 *>(p + 2 * PLANE) = pl[2];
}
// plane `pl` of the 32 rows starting at r0 (a multiple of 32)
__device__ __forceinline__ bf16x8 split_frag(const __bf16 *planes, int pl, int r0, int lane) {
    return *reinterpret_cast<const bf16x8 *>(planes + pl * PLANE + (r0 + (lane & 31)) * BK +
                                             (((lane >> 5) ^ ((lane >> 4) & 1)) << 3));
}

// ---- k-major operands (A given transposed, B row-major: the weight gradient  dW = dY^T X, k = rows) ------
// Image per plane [16 k][128 cols] bf16, 256-byte rows, written as it arrives (a thread owns 8 consecutive columns of
// one k).  The MFMA wants 8 consecutive k of ONE column per lane: ds_read_b64_tr_b16 (gfx950) reads, per 16 lanes, a
// 4-row x 16-column block and hands every lane one column of it -- two of them make the fragment, no transpose pass.
// Lane 4q + p of a 16-lane group supplies the address of row q, columns 4p .. 4p+3 of its block; lane i receives
// column i.  The 64-byte chunk index of a row is XORed with (k & 3): the four rows of a block then sit in different
// banks (256-byte rows would otherwise put them on the same 16), reads and the 16-byte writes are conflict-free.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int kmajor_off(int k, int c) {   // element offset of (k, c) inside a plane
    return k * BM + ((((c >> 5) ^ (k & 3)) << 5) | (c & 31));
}
__device__ __forceinline__ void kmajor_store(const bf16x8 (&pl)[3], __bf16 *planes, int t) {
    __bf16 *p = planes + kmajor_off(t >> 4, (t & 15) * 8);
    *reinterpret_cast<bf16x8 *>(p) = pl[0];
    *reinterpret_cast<bf16x8 *>(p + PLANE) = pl[1];
    *reinterpret_cast<bf16x8 *>(p + 2 * PLANE) = pl[2];
}
// plane `pl`, the 32 columns starting at c0 (a multiple of 32): this lane's column c0 + (lane & 31), k = 8 (lane >> 5) ..
__device__ __forceinline__ bf16x8 kmajor_frag(const __bf16 *planes, int pl, int c0, int lane) {
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int col = c0 + (lane & 16) + 4 * pp;
    const int k0 = 8 * (lane >> 5) + q;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const __bf16 *base = planes + pl * PLANE;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(base + kmajor_off(k0, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(base + kmajor_off(k0 + 4, col)));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// B of the split engine: the (small) weight operand is split ONCE per call into the exact LDS image of every
// (n tile, k tile) -- [tiles_n][ktiles][3 planes][128 rows][16 k, halves swizzled], zero-padded past N and K -- so the
// GEMM stages it with straight 16-byte copies and no VALU work.  tb: B given as [N][K] (nn.Linear weight), else [K][N].
__global__ __launch_bounds__(256) void presplit_b_kernel(const float *__restrict__ b, long ldb, int tb, long n, long k,
                                                         int ktiles, __bf16 *__restrict__ out) {
    const int t = threadIdx.x;
    const long tn = blockIdx.x / ktiles, kt = blockIdx.x % ktiles;
    const long row = tn * BN + (t >> 1), k0 = kt * BK + (t & 1) * 8;
    float v[EPT];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long kk = k0 + j;
        const bool ok = row < n && kk < k;
        const long off = tb ? min(row, n - 1) * ldb + min(kk, k - 1) : min(kk, k - 1) * ldb + min(row, n - 1);
        const float x = b[off];
        v[j] = ok ? x : 0.f;
    }
    bf16x8 pl[3];
    split_planes(v, pl);
    planes_store(pl, out + (long)blockIdx.x * (3 * PLANE), t);
}

template <bool TA, bool TB, int SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_kernel(GemmArgs g) {
    static_assert(SPLIT == 0 || (BK == 16 && EPT == 8 && (SPLIT == 1 ? !TA : (TA && !TB))),
                  "split engines: 1 = A row-major + pre-split B, 2 = both operands k-major (weight gradient)");
    constexpr int LDS_BYTES = SPLIT != 0 ? 2 * 2 * 3 * PLANE * 2 : 4 * IMG * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    float(*As)[IMG] = reinterpret_cast<float(*)[IMG]>(smem);                 // f32 engine: As[2], Bs[2]
    float(*Bs)[IMG] = reinterpret_cast<float(*)[IMG]>(smem) + 2;
    __bf16(*Sp)[2][3 * PLANE] = reinterpret_cast<__bf16(*)[2][3 * PLANE]>(smem);   // split engine: Sp[buf][A/B]
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
        B += (long)(g.b_period > 0 ? z % g.b_period : z) * g.stride_b;
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

    float bias_v[2] = {0.f, 0.f};       // this lane's two output columns; fetched now, long complete at the epilogue
    if (g.bias) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = n0 + wn * 64 + j * 32 + (lane & 31);
            bias_v[j] = g.bias[min(col, g.n - 1)];
        }
    }

    LA la;
    LB lb;
    auto fetch = [&](LA &ra, LB &rb, long k0) {
        const bool k_full = k0 + BK <= k_hi;
        if constexpr (TA)
            ra.load(A, g.lda, m0, m_hi, k0, k_hi, t, a_al && m_full && k_full && (m0 % 4 == 0));
        else
            ra.load(A, g.lda, m0, m_hi, k0, k_hi, t, a_al && m_full && k_full && (k0 % 4 == 0));
        if constexpr (TB)
            rb.load(B, g.ldb, n0, g.n, k0, k_hi, t, b_al && n_full && k_full && (k0 % 4 == 0));
        else
            rb.load(B, g.ldb, n0, g.n, k0, k_hi, t, b_al && n_full && k_full);
    };

    if constexpr (SPLIT == 1) {
        // One register set, one barrier per step.  Step t: barrier (tile t is complete in LDS buffer t & 1, nobody
        // still reads the other buffer) -> split + write tile t + 1 (its loads were issued a whole step ago) into the
        // other buffer -> re-issue the loads for tile t + 2 into the same registers -> fragment reads + MFMAs of
        // tile t.  B arrives as ready-made bf16 planes (a straight 48-byte copy per thread, no VALU work).
        // Variants measured on x[1M,256] @ W^T and NOT kept (all within +-5 % of this one, 1.0-1.1 ms): loads two tiles
        // ahead with two register sets (occupancy 2 instead of 3 waves per SIMD), the split spread between the MFMAs
        // with sched_group_barrier, on-the-fly splitting of B.  Traps met on the way: a guarded load inside the loop
        // makes hipcc wait vmcnt(0) every step; a small register ARRAY for the staged planes became an LDS-resident
        // alloca; an `asm volatile("" ::: "memory")` fence forced that state through memory.
        uint4 qb0, qb1, qb2;   // (scalars on purpose, see above)
        const uint4 *bsrc = reinterpret_cast<const uint4 *>(g.bp) +
                            ((long)tn * g.ktiles_b + k_lo / BK) * (3 * PLANE / 8) + t;
#define LKG_TERM(PA, PB)                                                                                    \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA], b[0][PB], acc[0][0], 0, 0, 0);            \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA], b[1][PB], acc[0][1], 0, 0, 0);            \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA], b[0][PB], acc[1][0], 0, 0, 0);            \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA], b[1][PB], acc[1][1], 0, 0, 0);
        auto stage = [&](__bf16(*D)[3 * PLANE]) {      // registers -> LDS image of one tile
            bf16x8 pa[3];
            split_planes(la.v, pa);
            planes_store(pa, D[0], t);
            uint4 *d = reinterpret_cast<uint4 *>(D[1]) + t;
            d[0] = qb0;
            d[PLANE / 8] = qb1;
            d[2 * (PLANE / 8)] = qb2;
        };
        auto mma = [&](const __bf16(*S)[3 * PLANE]) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[i][pl] = split_frag(S[0], pl, wm * 64 + i * 32, lane);
                    b[i][pl] = split_frag(S[1], pl, wn * 64 + i * 32, lane);
                }
            // smallest terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
            LKG_TERM(2, 0) LKG_TERM(0, 2) LKG_TERM(1, 1) LKG_TERM(1, 0) LKG_TERM(0, 1) LKG_TERM(0, 0)
        };
#undef LKG_TERM
        // tiles [first, first + count) of the k range.  FAST: every load is an unguarded 16-byte load and the
        // steady-state loop carries no branch around a load (the compiler then waits with counted vmcnt).
        auto pipeline = [&](auto fast_tag, long first, long count) {
            constexpr bool FAST = decltype(fast_tag)::value;
            auto fetch_tile = [&](long tile) {
                const long k0 = k_lo + tile * BK;
                la.load(A, g.lda, m0, m_hi, k0, k_hi, t, FAST);   // a CONSTANT flag: the guarded form is branch-free too
                const uint4 *src = bsrc + tile * (3 * PLANE / 8);
                qb0 = src[0];
                qb1 = src[PLANE / 8];
                qb2 = src[2 * (PLANE / 8)];
                __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: sunk towards their use they lose the prefetch
            };
            if (count <= 0) return;
            fetch_tile(first);
            stage(Sp[0]);
            if (count > 1) fetch_tile(first + 1);
            // invariant at the top of a trip: buffer 0 holds tile `it`, the registers hold (or are receiving) tile it + 1
            long it = 0;
            for (; it + 3 < count; it += 2) {       // unconditional body, two steps per trip (static buffer indices)
                __syncthreads();
                stage(Sp[1]);
                fetch_tile(first + it + 2);
                mma(Sp[0]);
                __syncthreads();
                stage(Sp[0]);
                fetch_tile(first + it + 3);
                mma(Sp[1]);
            }
            const long rem = count - it;            // 1, 2 or 3 tiles left
            __syncthreads();
            if (rem >= 2) stage(Sp[1]);
            if (rem == 3) fetch_tile(first + it + 2);
            mma(Sp[0]);
            if (rem >= 2) {
                __syncthreads();
                if (rem == 3) stage(Sp[0]);
                mma(Sp[1]);
            }
            if (rem == 3) {
                __syncthreads();
                mma(Sp[0]);
            }
            __syncthreads();
        };
        const long nt_all = (k_hi - k_lo + BK - 1) / BK, nt_full = (k_hi - k_lo) / BK;
        if (a_al && m_full && (k_lo % 4 == 0)) {
            pipeline(std::true_type{}, 0, nt_full);
            if (nt_all > nt_full) pipeline(std::false_type{}, nt_full, 1);   // the partial last k tile
        } else {
            pipeline(std::false_type{}, 0, nt_all);
        }
    } else if constexpr (SPLIT == 2) {
        // Both operands k-major and large (weight gradient): the same one-register-set pipeline, both operands split
        // on the fly (about 88 VALU per thread and step), fragments by transposing LDS reads (kmajor_frag).
#define LKG_TERM(PA, PB)                                                                                    \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA], b[0][PB], acc[0][0], 0, 0, 0);            \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA], b[1][PB], acc[0][1], 0, 0, 0);            \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA], b[0][PB], acc[1][0], 0, 0, 0);            \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA], b[1][PB], acc[1][1], 0, 0, 0);
        auto stage = [&](__bf16(*D)[3 * PLANE]) {      // one operand at a time: its three planes are dead before the next split
            {
                bf16x8 pa[3];
                split_planes(la.v, pa);
                kmajor_store(pa, D[0], t);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                bf16x8 pb[3];
                split_planes(lb.v, pb);
                kmajor_store(pb, D[1], t);
            }
        };
        auto mma = [&](const __bf16(*S)[3 * PLANE]) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[i][pl] = kmajor_frag(S[0], pl, wm * 64 + i * 32, lane);
                    b[i][pl] = kmajor_frag(S[1], pl, wn * 64 + i * 32, lane);
                }
            LKG_TERM(2, 0) LKG_TERM(0, 2) LKG_TERM(1, 1) LKG_TERM(1, 0) LKG_TERM(0, 1) LKG_TERM(0, 0)
        };
#undef LKG_TERM
        // fast_tag: bit 1 = A tile loads unguarded, bit 0 = B tile loads unguarded (compile-time constants: a runtime
        // flag would put a branch around the loads and cost a vmcnt(0) per step).  An edge column tile (n = 300: the
        // text-literal weight gradient) keeps its A side on 16-byte loads this way.
        auto pipeline = [&](auto fast_tag, long first, long count) {
            constexpr int FAST = decltype(fast_tag)::value;
            auto fetch_tile = [&](long tile) {
                const long k0 = k_lo + tile * BK;
                la.load(A, g.lda, m0, m_hi, k0, k_hi, t, (FAST & 2) != 0);
                lb.load(B, g.ldb, n0, g.n, k0, k_hi, t, (FAST & 1) != 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            if (count <= 0) return;
            fetch_tile(first);
            stage(Sp[0]);
            if (count > 1) fetch_tile(first + 1);
            long it = 0;
            for (; it + 3 < count; it += 2) {
                __syncthreads();
                stage(Sp[1]);
                fetch_tile(first + it + 2);
                mma(Sp[0]);
                __syncthreads();
                stage(Sp[0]);
                fetch_tile(first + it + 3);
                mma(Sp[1]);
            }
            const long rem = count - it;
            __syncthreads();
            if (rem >= 2) stage(Sp[1]);
            if (rem == 3) fetch_tile(first + it + 2);
            mma(Sp[0]);
            if (rem >= 2) {
                __syncthreads();
                if (rem == 3) stage(Sp[0]);
                mma(Sp[1]);
            }
            if (rem == 3) {
                __syncthreads();
                mma(Sp[0]);
            }
            __syncthreads();
        };
        const long nt_all = (k_hi - k_lo + BK - 1) / BK, nt_full = (k_hi - k_lo) / BK;
        const bool a_fast = a_al && m_full && (m0 % 4 == 0), b_fast = b_al && n_full;
        if (a_fast && b_fast)
            pipeline(std::integral_constant<int, 3>{}, 0, nt_full);
        else if (a_fast)
            pipeline(std::integral_constant<int, 2>{}, 0, nt_full);
        else if (b_fast)
            pipeline(std::integral_constant<int, 1>{}, 0, nt_full);
        else
            pipeline(std::integral_constant<int, 0>{}, 0, nt_full);
        if (nt_all > nt_full) pipeline(std::integral_constant<int, 0>{}, nt_full, 1);   // the partial last k tile
    } else {
        int buf = 0;
        if (k_lo < k_hi) {
            fetch(la, lb, k_lo);
            la.store(As[0], t);
            lb.store(Bs[0], t);
        }
        __syncthreads();
        for (long k0 = k_lo; k0 < k_hi; k0 += BK) {
            const bool more = k0 + BK < k_hi;
            if (more) fetch(la, lb, k0 + BK);
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
    }

    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // The 16 stores of a tile are issued back to back, all reads they need (bias: fetched before the k loop; old C for
    // beta != 0: the tile's 16 loads first) done beforehand: stores count in vmcnt on gfx9, so a load pending anywhere
    // between them made hipcc drain with vmcnt(0) before EVERY store -- 64 serialised write round trips per thread,
    // 42 of the 81 k-cycles of a workgroup on x[1M,256] @ W^T.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.n) continue;
            const float bv = (!atomic_out || z == 0) ? bias_v[j] : 0.f;
            const long row0 = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
            float *dst0 = C + row0 * g.ldc + col;
            const bool all_rows = m0 + wm * 64 + i * 32 + 32 <= m_hi;      // wave-uniform
            float out[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) out[r] = g.alpha * acc[i][j][r] + bv;
            if (atomic_out) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (all_rows || row0 + dr < m_hi) atomicAdd(dst0 + dr * g.ldc, out[r]);
                }
            } else {
                if (g.beta != 0.f) {
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        old[r] = (all_rows || row0 + dr < m_hi) ? dst0[dr * g.ldc] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) out[r] = fmaf(g.beta, old[r], out[r]);
                }
                if (all_rows) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst0[((r & 3) + 8 * (r >> 2)) * g.ldc] = out[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        if (row0 + dr < m_hi) dst0[dr * g.ldc] = out[r];
                    }
                }
            }
        }
}

int run(bool ta, bool tb, const GemmArgs &g, dim3 grid, hipStream_t s) {
    if (g.bp)             // split engine 1 (A row-major, B pre-split)
        hipLaunchKernelGGL((gemm_kernel<false, true, 1>), grid, dim3(256), 0, s, g);
    else if (g.split_km)  // split engine 2 (both operands k-major)
        hipLaunchKernelGGL((gemm_kernel<true, false, 2>), grid, dim3(256), 0, s, g);
    else if (!ta && !tb)
        hipLaunchKernelGGL((gemm_kernel<false, false, 0>), grid, dim3(256), 0, s, g);
    else if (!ta && tb)
        hipLaunchKernelGGL((gemm_kernel<false, true, 0>), grid, dim3(256), 0, s, g);
    else if (ta && !tb)
        hipLaunchKernelGGL((gemm_kernel<true, false, 0>), grid, dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_kernel<true, true, 0>), grid, dim3(256), 0, s, g);
    LKG_CHECK_LAUNCH("lkg_gemm_f32");
    return LKG_OK;
}

}  // namespace

// bytes of workspace with which lkg_gemm_f32 can run this product on split engine 1 (0: the engine does not apply)
extern "C" int64_t lkg_gemm_workspace(int32_t trans_a, int64_t m, int64_t n, int64_t k) {
    if (trans_a || k <= 0 || m < 16384 || n * k > (1L << 22)) return 0;
    const long tiles_n = (n + BN - 1) / BN, ktiles = (k + BK - 1) / BK;
    return tiles_n * ktiles * 3 * PLANE * (long)sizeof(__bf16);
}

extern "C" int lkg_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, float alpha,
                            const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                            int64_t ldc, const float *bias, void *workspace, int64_t workspace_bytes,
                            void *stream) {
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
    } else if (beta == 0.f && tiles <= 32 && k >= 2048) {
        // a few thousand rows (the weight gradient over the rows a loss reaches): a handful of tiles would walk k alone
        splits = (int)std::min<long>(256 / tiles, k / 256);
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
    // LKG_GEMM_F32_ONLY=1 in the environment keeps every product on the f32-input MFMA (the bit-exact k-ordered fmaf
    // chain), e.g. to bisect a numerical difference; read once.
    static const bool f32_only = [] {
        const char *e = getenv("LKG_GEMM_F32_ONLY");
        return e && e[0] == '1';
    }();
    const int ktiles = (int)((k + BK - 1) / BK);
    // Split engine 1: A row-major and B small enough to pre-split per call (a weight matrix), enough rows to pay for
    // the extra launch -- when the CALLER handed over the workspace for B's planes (lkg_gemm_workspace; no allocation
    // here).  (below ~16 k rows the call is bound by its host-side issue, ~12 us)
    void *ws = nullptr;
    const int64_t need = lkg_gemm_workspace(trans_a, m, n, k);
    if (!f32_only && need > 0 && workspace && workspace_bytes >= need) ws = workspace;
    if (ws) {
        hipLaunchKernelGGL(presplit_b_kernel, dim3((unsigned)(g.tiles_n * ktiles)), dim3(256), 0, s, b, (long)ldb,
                           (int)(trans_b != 0), (long)n, (long)k, ktiles, reinterpret_cast<__bf16 *>(ws));
        g.bp = reinterpret_cast<const __bf16 *>(ws);
        g.ktiles_b = ktiles;
    }
    g.split_km = (!f32_only && trans_a && !trans_b && k >= 2048) ? 1 : 0;   // long reductions over rows: weight gradients
    return run(trans_a != 0, trans_b != 0, g, dim3((unsigned)tiles, (unsigned)splits), s);
}

extern "C" int lkg_grouped_gemm_f32(int32_t mode, int32_t n_groups, const int32_t *seg, int64_t max_seg_len,
                                    int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, float alpha,
                                    const float *a, int64_t lda, const float *b, int64_t ldb, int64_t stride_b,
                                    int32_t b_period, float beta, float *c, int64_t ldc, int64_t stride_c,
                                    void *stream) {
    LKG_REQUIRE(mode == 1 || mode == 2, "lkg_grouped_gemm_f32: mode must be 1 (rows) or 2 (k)");
    LKG_REQUIRE(n_groups >= 0 && n_groups <= 65535 && max_seg_len >= 0, "lkg_grouped_gemm_f32: bad group count");
    if (n_groups == 0 || n == 0) return LKG_OK;
    LKG_REQUIRE(seg && a && b && c && ldc >= n, "lkg_grouped_gemm_f32: null pointer / bad ldc");
    LKG_REQUIRE(mode == 1 ? !trans_a : (trans_a && !trans_b),
                "lkg_grouped_gemm_f32: rows mode needs A row-major, k mode needs A^T and B stored k-major");
    GemmArgs g{};
    g.alpha = alpha; g.beta = beta; g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.c = c; g.ldc = ldc;
    g.bias = nullptr; g.seg = seg; g.stride_b = stride_b; g.stride_c = stride_c; g.mode = mode; g.k_splits = 1;
    g.b_period = mode == 1 ? b_period : 0;
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

// ---------------------------------------------------------------------------------------------------
// Small products with float64 accumulation, rounded to float32 ONCE: C = op(A) op(B).  For the GCNII-style residual's weight
// fold W_lin (W')^T (model.py:95-98: d x d x out, once per layer and step): W' = (1 - b) + b W has near-equal entries, so
// the product is a large common part plus a small informative one, and the LayerNorm behind it removes the common part --
// an fp32 dot product's accumulated rounding there is what the scores of ill-conditioned residual configurations see.
// One thread per output element, 16 x 16 outputs per workgroup; a few MFLOP: the launch is its cost.
namespace {
__global__ __launch_bounds__(256) void gemm_f64acc_kernel(int ta, int tb, long m, long n, long k, const float *__restrict__ a,
                                                          long lda, const float *__restrict__ b, long ldb,
                                                          float *__restrict__ c, long ldc) {
    const long i = (long)blockIdx.y * 16 + (threadIdx.x >> 4), j = (long)blockIdx.x * 16 + (threadIdx.x & 15);
    if (i >= m || j >= n) return;
    double acc = 0.0;
    for (long p = 0; p < k; ++p) {
        const float av = ta ? a[p * lda + i] : a[i * lda + p];
        const float bv = tb ? b[j * ldb + p] : b[p * ldb + j];
        acc = fma((double)av, (double)bv, acc);
    }
    c[i * ldc + j] = (float)acc;
}
}  // namespace

extern "C" int lkg_gemm_f64acc_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, const float *a,
                                   int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc, void *stream) {
    LKG_REQUIRE(m >= 0 && n >= 0 && k >= 0, "lkg_gemm_f64acc_f32: negative size");
    if (m == 0 || n == 0) return LKG_OK;
    LKG_REQUIRE(c && ldc >= n && (k == 0 || (a && b)), "lkg_gemm_f64acc_f32: null pointer / ldc smaller than n");
    LKG_REQUIRE(k == 0 || (lda >= (trans_a ? m : k) && ldb >= (trans_b ? k : n)), "lkg_gemm_f64acc_f32: row strides smaller "
                "than the stored rows");
    LKG_REQUIRE(m * n <= (1L << 24) && m * n * k <= (1L << 34), "lkg_gemm_f64acc_f32: a helper for SMALL products (weight "
                "folds); %lld x %lld x %lld belongs on lkg_gemm_f32", (long long)m, (long long)n, (long long)k);
    hipLaunchKernelGGL(gemm_f64acc_kernel, dim3((unsigned)((n + 15) / 16), (unsigned)((m + 15) / 16)), dim3(256), 0,
                       (hipStream_t)stream, trans_a, trans_b, (long)m, (long)n, (long)k, a, (long)lda, b, (long)ldb, c, (long)ldc);
    LKG_CHECK_LAUNCH("lkg_gemm_f64acc_f32");
    return LKG_OK;
}

// lkg_preload(): HIP loads a translation unit's code object on the first use of one of its kernels; asking for a kernel's
// attributes is such a use (no launch).
int lkg_internal_preload_gemm() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&presplit_b_kernel)) == hipSuccess ? 0 : 1;
}
