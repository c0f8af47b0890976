// Tall f32 GEMM for the dense part of the layers (nn.Linear forward / data gradient, the literal gate): M = entities
// (millions), N <= a few hundred, K <= ~1000, f32 in / f32 out.
//
//   C[m, n] = epilogue( sum over K-panels p of  A_p[m, K_p] . B_p[n, K_p]^T )
//
// Arithmetic: "f16 x 2" on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, 2.5 PF dense) with f32 accumulation.
//   Every row of A (and every row of B = output column) is scaled by a power of two so that its largest magnitude lies
//   in [2^13, 2^14), then every element is split into  hi = fp16(a') (round toward zero: the residual is then exact)  and
//   mid = fp16((a' - hi) * 2^11)  -- 11 + 11 significant bits, both in fp16's NORMAL range for every element down to
//   2^-27 of its row's maximum (below that the absolute error is < 2^-39 of the row maximum).  Then
//       a'.b' = hi_a hi_b + 2^-11 (hi_a mid_b + mid_a hi_b) + O(2^-22 |a'||b'|)
//   three MFMAs per 16 k and 32x32 tile -- into TWO accumulators (main, correction), combined and unscaled (an exact
//   ldexp by -(e_row + e_col)) in the epilogue -- instead of the six of the bf16 x 3 engine or the eight 32x32x2 f32
//   MFMAs.  Against f64 the result is within ~2x of an f32 GEMM's own rounding error (products carry 2^-22 instead of
//   2^-24; the f32 accumulation over K dominates either way).
//
// K-panels: A may be given as up to three column panels from DIFFERENT arrays (the gate's [x | num | txt], gate.py:23,
// graphsage's [ego | side]); the accumulators stay in registers across panels, so no concatenation and no C
// read-modify-write.  B rows may come from two stacked weight groups (the gate's g and z projections) interleaved in
// blocks of 32 so that one lane holds g[row, c] and z[row, c]: the blend (gate.py:24-26) happens in the epilogue.
//
// Tile: 128 rows x BN columns (BN = 256: a layer's whole width, A is read and split ONCE; 128 for narrow outputs),
// 16-k steps, 2 x BN/64 waves of 64 x 64, double-buffered LDS planes, one barrier per step.  Every load of the k loop is
// an LDS-DMA (global_load_lds_dwordx4, no VGPR destination, waits counted by hand): B arrives as ready-made fp16 planes
// (prepared once per call in the caller's workspace, an exact image of the LDS tile) straight into the other buffer; A's
// raw f32 windows land in a 3-deep ring, two steps ahead of their use, and the thread that asked for a window reads it
// back, splits it and writes its piece of the planes IN THE SHADOW of the step's 12 MFMAs (the split is cut into pieces
// pinned between them: an MFMA holds the matrix pipe for 32 cycles but the issue port for 8).
// Per workgroup at 1 M x 256 x 256 (s_memtime stamps, 1.8-1.95 GHz under this load): prologue 7-8 k cycles (one HBM
// round trip), k loop 42 k (16 steps of 2.6 k with two workgroups per CU: the matrix pipe 58 % busy), epilogue 11-13 k.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "lkg_common.h"

namespace {

#ifndef LKG_TALL_DEFAULT_VARIANT
#define LKG_TALL_DEFAULT_VARIANT 2      /* 0 "256x2", 1 "128x1", 2 "256x1" (8 waves of 64 x 64), 3 "256x1w" (4 waves of 64 x 128) */
#endif
constexpr int TM = 128, TK = 16, MAX_PANELS = 3;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_PLAIN = 0, EPI_GATE = 1, EPI_ACTLN = 2 };

// In-kernel stamps (a DIAGNOSTIC build only: LKG_EXTRA_HIPCC_FLAGS=-DLKG_WS_STAMPS; tools/ws_stamps.py): every wave sums the
// cycles it spends waiting for memory, waiting at the step barrier and working; lane 0 adds the sums to the buffer the
// caller hands in as the gate's (otherwise unused) z_out of a PLAIN product.  No stamp executes in the product build.
#ifdef LKG_WS_STAMPS
#define LKG_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define LKG_STAMP_ADD(acc, a, b) acc += (b) - (a)
#else
#define LKG_STAMP(var)
#define LKG_STAMP_ADD(acc, a, b)
#endif

// LDS reads of an EPILOGUE, hidden from the compiler.  hipcc cannot tell which LDS-DMA (global_load ... lds) wrote the bytes a
// ds_read it can see may alias, so it puts `s_waitcnt vmcnt(0)` in front of every such read of the staging array -- and vmcnt
// counts STORES too: in a loop "read 16 bytes back from the transpose, store them" every read then waits for the previous
// store's round trip to memory (~1000 cycles each; measured with the stamps below: 16-20 k of a tile's 56 k cycles).  The
// bytes these reads want were written by ds_write of the same wave (the LDS queue is in order per wave) or sit behind a
// workgroup barrier; no DMA targets them while an epilogue runs (the next tile's first loads land beyond the transposes).
typedef float lkg_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const void lkg_lds_cvoid;
__device__ __forceinline__ unsigned lds_addr_of(const void *p) { return (unsigned)(unsigned long)(lkg_lds_cvoid *)p; }
__device__ __forceinline__ void lds_read16_issue(lkg_f32x4 &v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
}
#define LKG_LDS_READS_DONE_4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory")
#define LKG_LDS_READS_DONE_3(a, b, c) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) :: "memory")
#define LKG_LDS_READS_DONE_2(a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) :: "memory")


struct TallArgs {
    long m;
    int n;                       // GEMM width (stacked width 2 d for the gate)
    int n_panels;
    const float *a[MAX_PANELS];
    long lda[MAX_PANELS];
    int ka[MAX_PANELS];
    int ktiles[MAX_PANELS];
    int ktiles_total;
    const float *a_rowmax;       // float[m]: max |A[i, :]| over all panels
    const _Float16 *bp;          // planes [tiles_n][ktiles_total][2][BN][16]
    const int *eb;               // int32[tiles_n * BN] exponents of the (stacked, interleaved) B rows
    float alpha, beta;
    float *c;
    long ldc;
    const float *bias;           // float[n] in GEMM column order (stacked for the gate), nullable
    // gate epilogue: out = (1 - sigmoid(z)) x + sigmoid(z) tanh(g)
    const float *x;
    long ldx;
    float *g_out, *z_out;        // nullable: tanh(g) / sigmoid(z) kept for the backward
    long ldg, ldz;
    int tiles_m, tiles_n;
    // act + LayerNorm epilogue (EPI_ACTLN, one column tile: n <= 256): y = Dropout(LayerNorm(LeakyReLU(A B^T + bias))) -> c
    // (nullable), yn = y / max(|y|_2, norm_eps) -> yn (nullable), the row statistics the backward needs -> mean / rstd
    float slope, ln_eps, norm_eps, drop_p;
    unsigned long long seed;
    const float *gamma, *beta_ln;
    float *yn;
    long ldyn;
    float *mean, *rstd;
};

// exponent e with max * 2^e in [2^13, 2^14)   (0 for max == 0 / denormal; clamped so that ldexp stays finite)
__device__ __forceinline__ int scale_exponent(float mx) {
    const int ex = (__float_as_int(mx) >> 23) & 0xff;
    if (ex == 0 || ex == 0xff) return 0;
    return max(-100, min(100, 13 - (ex - 127)));
}

// one element group -> hi / mid fp16 pairs.  PRE: the residual is stored as it is (mid' = a' - hi, the 2^-11 of the cross
// terms already inside the operand: fp16 subnormals for elements below 2^-17 of their row maximum) instead of scaled by
// 2^11 into fp16's normal range -- the one-accumulator kernels of 64 x 128 wave tiles use it and need no scaled copy of hi.
template <bool PRE = false>
__device__ __forceinline__ void split2(float a0, float a1, fp16x2 &hi, fp16x2 &mid) {
    hi = __builtin_amdgcn_cvt_pkrtz(a0, a1);
    constexpr float sc = PRE ? 1.f : 2048.f;
    const float r0 = (a0 - (float)hi[0]) * sc, r1 = (a1 - (float)hi[1]) * sc;
    mid = __builtin_amdgcn_cvt_pkrtz(r0, r1);
}

// element (row, k) of a plane: rows of 16 halves, the two 8-half groups swapped on rows 16-31 of every 32 (conflict-free
// ds_read_b128 fragments, as in lkg_gemm.hip)
__device__ __forceinline__ int plane_off(int row, int k) { return row * TK + ((((k >> 3) ^ ((row >> 4) & 1)) << 3) | (k & 7)); }

template <int BN>
__device__ __forceinline__ f16x8 frag(const _Float16 *plane, int r0, int lane) {
    return *reinterpret_cast<const f16x8 *>(plane + (r0 + (lane & 31)) * TK + (((lane >> 5) ^ ((lane >> 4) & 1)) << 3));
}

// ---------------------------------------------------------------------------------------------- B preparation
struct BDesc {
    const float *ptr[2][MAX_PANELS];     // [group][panel]
    long ld[2][MAX_PANELS];
    int k[MAX_PANELS];
    int n_groups, n_panels, rows_per_group, trans_b;   // trans_b: 1 = stored [rows][K] (nn.Linear weight), 0 = [K][rows]
    int interleave;                      // 2 groups: tile column s -> group (s / 32) & 1, row (s / 64) * 32 + s % 32
};

__device__ __forceinline__ bool b_source(const BDesc &b, int s, int &group, int &row) {
    if (b.interleave) {
        group = (s >> 5) & 1;
        row = (s >> 6) * 32 + (s & 31);
    } else {
        group = 0;
        row = s;
    }
    return row < b.rows_per_group && group < b.n_groups;
}

__device__ __forceinline__ float b_elem(const BDesc &b, int group, int row, int p, int k) {
    const float *src = b.ptr[group][p];
    return b.trans_b ? src[(long)row * b.ld[group][p] + k] : src[(long)k * b.ld[group][p] + row];
}

// exponent of every stacked B row: one wave per row, lanes stride over k
__global__ __launch_bounds__(256) void b_exponent_kernel(BDesc b, int n_stacked, int *__restrict__ eb) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_stacked) return;
    int group, row;
    float mx = 0.f;
    if (b_source(b, s, group, row))
        for (int p = 0; p < b.n_panels; ++p)
            for (int k = lane; k < b.k[p]; k += 64) mx = fmaxf(mx, fabsf(b_elem(b, group, row, p, k)));
    mx = wave_max(mx);
    if (lane == 0) eb[s] = scale_exponent(mx);
}

// planes of one (n tile, k tile): thread = tile row
template <int BN>
__global__ __launch_bounds__(BN) void b_planes_kernel(BDesc b, int ktiles_total, const int *__restrict__ eb,
                                                      _Float16 *__restrict__ out, int pre) {
    const int tn = blockIdx.x / ktiles_total, kt = blockIdx.x % ktiles_total;
    int p = 0, k0 = kt;
    while (p < b.n_panels - 1 && k0 >= (b.k[p] + TK - 1) / TK) {
        k0 -= (b.k[p] + TK - 1) / TK;
        ++p;
    }
    k0 *= TK;
    const int s = tn * BN + threadIdx.x;
    int group, row;
    const bool ok = b_source(b, s, group, row);
    const int e = eb[s];
    _Float16 *dst = out + (long)blockIdx.x * (2 * BN * TK);
#pragma unroll
    for (int k = 0; k < TK; k += 2) {
        float v0 = 0.f, v1 = 0.f;
        if (ok && k0 + k < b.k[p]) v0 = ldexpf(b_elem(b, group, row, p, k0 + k), e);
        if (ok && k0 + k + 1 < b.k[p]) v1 = ldexpf(b_elem(b, group, row, p, k0 + k + 1), e);
        fp16x2 hi, mid;
        if (pre) split2<true>(v0, v1, hi, mid); else split2<false>(v0, v1, hi, mid);
        const int off = plane_off(threadIdx.x, k);
        *reinterpret_cast<fp16x2 *>(dst + off) = hi;
        *reinterpret_cast<fp16x2 *>(dst + BN * TK + off) = mid;
    }
}

// ---------------------------------------------------------------------------------------------- the GEMM
// ONE: a single accumulator per tile.  The 2^-11 of the two cross terms is then applied to an OPERAND instead of to a
// second accumulator: hs = hi * 2^-11 (an exact exponent shift, one v_pk_mul_f16 per register, down to fp16's
// subnormals for elements below 2^-17 of their row maximum) and  a'.b' = hi_a hi_b + hs_a mid_b + mid_a hs_b.  64
// accumulator registers less per lane: three 4-wave workgroups (or two 8-wave ones) share a CU and their phases overlap.
// WN: columns per wave.  64: 2 x BN/64 waves of 64 x 64 (the forms above).  128 (BN = 256, one accumulator, prescaled
// mids): FOUR waves of 64 x 128 -- one wave per SIMD and workgroup, two workgroups per CU at up to 256 VGPRs each, 24 MFMAs
// per wave and barrier instead of 12, 48 KB of fragment reads per step instead of 64 KB, no scaled operand copies: while one
// workgroup of a CU stores its tile the other has the matrix pipe of all four SIMDs to itself.
// TM_: rows per tile.  128 (the forms above), or 256 with 64 x 128 wave tiles: EIGHT waves (4 x 2), one workgroup per CU.  Why:
// in-kernel stamps (tools/ws_stamps.py, profiles/r04_tall_stamps.log) put the 128-row forms at the rate at which a CU's vector
// memory pipe takes requests -- per 128-row tile at K = 256 it moves 128 KB of A, 256 KB of B planes (re-streamed from the
// L2 for every row tile: HALF of the bytes) and 128 KB of C, ~18 B per cycle; waves spend 20 % of a step issuing requests
// and an epilogue of 19 k cycles on 16 store instructions.  A 256-row tile reads B's planes once per 256 rows.
template <int BN, int EPI, bool ONE, int WN = 64, int TM_ = 128>
__global__ __launch_bounds__((TM_ / 64) * (BN / WN) * 64) __attribute__((amdgpu_waves_per_eu(WN == 128 ? 2 : (ONE ? (BN == 128 ? 3 : 4) : 2), WN == 128 ? 2 : (ONE ? (BN == 128 ? 3 : 4) : 2))))
void gemm_tall_kernel(TallArgs g) {
    static_assert(WN == 64 || (WN == 128 && BN == 256 && ONE), "64 x 128 wave tiles: 256-column tiles, one accumulator");
    static_assert(TM_ == 128 || (TM_ == 256 && WN == 128), "256-row tiles: 64 x 128 wave tiles");
    constexpr int TM = TM_;                       // (shadows the namespace's 128 for everything below)
    constexpr bool PRE = WN == 128;               // mid planes hold the residual itself (split2<true>)
    constexpr int NJ = WN / 32;                   // 32-column blocks per wave
    constexpr int NT = (TM / 64) * (BN / WN) * 64;        // threads
    constexpr int EPT = TM * TK / NT;             // A floats per thread per k tile: 4 (BN = 256) or 8 (BN = 128)
    constexpr int TPR = TK / EPT;                 // threads per A row
    constexpr int APL = TM * TK, BPL = BN * TK;   // halves per plane
    constexpr int BUF = 2 * APL + 2 * BPL;        // halves per buffer
    // ONE dynamic LDS object: staging planes, the raw A ring, the tile's row / column exponents and bias (76 KB at BN = 256:
    // two workgroups per CU, the gate's included)
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    constexpr int RING = 3;                        // raw A tiles (f32, as loaded) in flight
    float *raw_s = reinterpret_cast<float *>(smem + 2 * BUF);
    // exponents / bias of the tile's rows and BN stacked columns: read in its epilogue without a global round trip
    int *ea_s = reinterpret_cast<int *>(raw_s + RING * TM * TK);
    int *eb_s = ea_s + TM;
    float *bias_s = reinterpret_cast<float *>(eb_s + BN);
    // ... and the landing area of the NEXT tile's scalars: requested by LDS-DMA before the current tile's epilogue (no register
    // lives across it for them), turned into the arrays above by tile_open() once that epilogue is through
    float *pre_rm_s = bias_s + BN;                 // [NT] row maximum of every thread's A row
    float *pre_ear_s = pre_rm_s + NT;              // [TM] row maxima of the tile's rows
    int *pre_eb_s = reinterpret_cast<int *>(pre_ear_s + TM);      // [BN]
    float *pre_bias_s = reinterpret_cast<float *>(pre_eb_s + BN); // [BN]

    // Thread coordinates.  Re-derived at the top of every tile from an OPAQUE copy of threadIdx.x (rethread()): everything
    // computed from them -- LDS addresses, window offsets, the epilogue's address arithmetic -- is then per-tile work the
    // compiler cannot hoist out of the persistent loop, where it would occupy registers across a k loop that has none to
    // spare (hoisted, 25-58 VGPRs spilled around it; re-derived, a few dozen integer instructions per 60 k-cycle tile).
    int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int wm = wave / (BN / WN), wn = wave % (BN / WN);
    // PERSISTENT workgroups, XCD-aware: workgroup b lives on XCD b & 7 and walks every (gridDim.x / 8)-th tile of that XCD's
    // contiguous range of row tiles (n fastest).  Before a tile's epilogue the workgroup requests the NEXT tile's first B
    // planes and A windows: the HBM round trip that used to open every tile (7-8 k cycles of 60 k) runs under the epilogue.
    const int tiles = g.tiles_m * g.tiles_n;
    const int cpx = tiles >> 3, rem = tiles & 7, xcd = blockIdx.x & 7;
    const int xcd_first = xcd * cpx + min(xcd, rem), xcd_count = cpx + (xcd < rem ? 1 : 0);
    const int slot_stride = max(1, (int)gridDim.x >> 3);
    int slot = blockIdx.x >> 3;
    if (slot >= xcd_count) return;                           // (workgroup-uniform)

    int arow = t / TPR, akc = t % TPR;                       // this thread's row / k chunk of the A tile
    // A LAST column tile with real columns in at most half of its BN (300 = 256 + 44; the gate's 600 = 2 x 256 + 88): the waves whose
    // columns are all padding run no MFMAs, and the 2 x 4 wave grid is laid out the other way round for that tile (wn = wave >> 1:
    // the live waves are waves 0..3 or 0..1, one per SIMD) -- the tile's matrix work halves instead of idling two SIMDs.  The k loop
    // exists twice (live / padding): a branch between the accumulators' uses inside it costs 12-170 spilled VGPRs.
    constexpr bool SKIPS = (BN / WN == 4 && TM == 128 && EPI == EPI_PLAIN && ONE);     // (the gate's and the fused layer's kernels have no register to spare)
    int alt = 0;                           // (of the tile in the k loop / its epilogue; workgroup-uniform)
    bool act = true;                       // (wave-uniform)
    auto rethread = [&]() {
        int t_o = threadIdx.x;
        asm volatile("" : "+v"(t_o));
        t = t_o; lane = t & 63;
        if constexpr (SKIPS) wave = __builtin_amdgcn_readfirstlane(t >> 6);        // (a scalar: so are wm, wn and what hangs on them)
        else wave = t >> 6;
        if (SKIPS && alt) {
            wm = wave & 1; wn = wave >> 1;
        } else {
            wm = wave / (BN / WN); wn = wave % (BN / WN);
        }
        arow = t / TPR; akc = t % TPR;
    };
    auto remap = [&](int n0_tile) {        // the wave grid of the tile that starts now
        if constexpr (SKIPS) {
            const int n_act = min(BN / WN, (g.n - n0_tile + WN - 1) / WN);
            alt = n_act <= 2 ? 1 : 0;
            const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            act = (alt ? (wave_u >> 1) : (wave_u % (BN / WN))) < n_act;
            rethread();
        }
    };
    // ---- per-tile state (set by setup(): the tile being computed, or -- from just before its epilogue on -- the next one)
    long m0 = 0, grow = 0;
    int n0 = 0, ea = 0;

    f32x16 acc[2][NJ], cor[ONE ? 1 : 2][ONE ? 1 : 2];

    // ONE branch-free load path for every tile of every panel (aligned or not, full or partial): a 16-byte window per
    // thread, its address clamped to the panel's last valid window (rows past m reuse row m-1, never stored; a window
    // that would run past the end of the LAST row is moved back and its elements shifted into place), columns past the
    // panel width masked to 0.  A branch around the loads would put an s_waitcnt vmcnt(0) at its join, every step.
    typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef __attribute__((address_space(1))) const float4u gf4;
    // byte addresses as integers; (static indices into the kernel arguments: a runtime index would make hipcc keep a
    // private copy of the struct)
#define LKG_PANEL(P)                                                                                                \
    const unsigned long pa##P = reinterpret_cast<unsigned long>((P < g.n_panels) ? g.a[P] : g.a[0]);              \
    const long ld##P = (P < g.n_panels) ? g.lda[P] : g.lda[0];                                                     \
    const int kw##P = (P < g.n_panels) ? g.ka[P] : g.ka[0];                                                        \
    unsigned long rowp##P = 0;                                                                                      \
    const unsigned long lastw##P = pa##P + 4ul * (unsigned long)((g.m - 1) * ld##P + kw##P - 4);
    LKG_PANEL(0)
    LKG_PANEL(1)
    LKG_PANEL(2)
#undef LKG_PANEL
    const int n_tiles = g.ktiles_total;

    typedef __attribute__((address_space(3))) void lds_void;
    const uint4 *bsrc = nullptr;
    // the registers that depend on the tile (pure integer arithmetic: re-derived rather than kept across an epilogue)
    auto tile_regs = [&](int slot_) {
        const int tile = xcd_first + slot_;
        const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
        m0 = (long)tm * TM;
        n0 = tn * BN;
        grow = min(m0 + arow, g.m - 1);                       // clamped: rows past m are computed and never stored
        rowp0 = pa0 + 4ul * (unsigned long)(grow * ld0 + akc * EPT);
        rowp1 = pa1 + 4ul * (unsigned long)(grow * ld1 + akc * EPT);
        if constexpr (EPI != EPI_ACTLN) rowp2 = pa2 + 4ul * (unsigned long)(grow * ld2 + akc * EPT);      // (the fused layer: <= 2 panels)
        bsrc = reinterpret_cast<const uint4 *>(g.bp) + (long)tn * g.ktiles_total * (2 * BPL / 8) + t;
    };
    // the tile's scalars from global memory: requested here (before the previous tile's epilogue) by LDS-DMA, one dword per
    // lane into the landing area; every thread reads back what its own wave requested (tile_open)
    auto tile_prefetch = [&]() {
        typedef __attribute__((address_space(1))) const unsigned gu1;
        auto dma4 = [&](const void *src, void *dst_wave) {
            __builtin_amdgcn_global_load_lds((gu1 *)src, (lds_void *)dst_wave, 4, 0, 0);
        };
        dma4(g.a_rowmax + grow, pre_rm_s + wave * 64);
        if (wave < TM / 64) dma4(g.a_rowmax + min(m0 + t, g.m - 1), pre_ear_s + wave * 64);       // (wave-uniform)
        if (wave < BN / 64) {
            dma4(g.eb + n0 + t, pre_eb_s + wave * 64);
            if (g.bias) {
                int idx;
                if constexpr (EPI == EPI_GATE) {   // stacked column s of the tile: group (s / 32) & 1, output column (s / 64) * 32 + s % 32
                    const int d = g.n / 2, oc = min((n0 >> 1) + (t >> 6) * 32 + (t & 31), d - 1);
                    idx = ((t >> 5) & 1) * d + oc;
                } else {
                    idx = min(n0 + t, g.n - 1);
                }
                dma4(g.bias + idx, pre_bias_s + wave * 64);
            }
        }
    };
    auto tile_open = [&]() {                       // (after the previous epilogue: its LDS scalars may be overwritten now)
        ea = scale_exponent(pre_rm_s[t]);
        if (t < TM) ea_s[t] = scale_exponent(pre_ear_s[t]);
        if (t < BN) {
            eb_s[t] = pre_eb_s[t];
            bias_s[t] = g.bias ? pre_bias_s[t] : 0.f;
        }
    };

    // ---- loads: LDS-DMA only (global_load_lds_dwordx4: 1 KB per wave-instruction, no VGPR destination).  B's planes are
    // ready-made LDS images, so a tile is a straight copy into the other buffer; A's raw f32 windows go into a RING-deep
    // ring of which every thread reads back exactly the 16-byte pieces it requested itself (no cross-thread hazard, the
    // ring only replaces the registers a deeper prefetch would need).  The waits are counted by hand (vmcnt is in issue
    // order): per step the loads are issued as [B tile, A windows], the top of a step needs everything but the newest A.
    constexpr int NA = EPT / 4;                    // A windows (LDS-DMA instructions) per thread per tile
    // the tile the next A fetch reads: (panel, tile inside the panel) walked incrementally -- a closed form
    // (gt >= kt1 ? ... : ...) became a lookup table in private memory.  The staging side walks the same sequence
    // RING - 1 tiles behind (wave-uniform scalars), the per-thread window shifts travel in a small bit FIFO.
    int f_gt = 0, f_tk = 0, f_panel = 0, f_nt = g.ktiles[0];
    unsigned long f_rowp = 0, f_lastw = lastw0;
    int s_gt = 0, s_tk = 0, s_panel = 0, s_nt = g.ktiles[0], s_kp = kw0;
    unsigned sh_fifo = 0;                          // 4 bits per tile in flight: 2 per window
    auto reset_walks = [&]() {                     // (a new tile: both walks start over at its first k tile)
        f_gt = f_tk = f_panel = 0; f_nt = g.ktiles[0]; f_rowp = rowp0; f_lastw = lastw0;
        s_gt = s_tk = s_panel = 0; s_nt = g.ktiles[0]; s_kp = kw0;
        sh_fifo = 0;
    };
    unsigned long a_ptr[NA];                       // the windows of the tile the next issue_a requests
    // (all scalar bookkeeping -- the two walks -- happens in plan_a / plan_stage, ahead of the step's one basic block of
    // barrier, DMA issue, MFMAs and split: a branch inside would keep the scheduler from mixing the split into the MFMAs)
    auto plan_a = [&](int fifo_pos) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const unsigned long want = f_rowp + 4ul * (unsigned long)(f_tk * TK + 4 * q);
            a_ptr[q] = want < f_lastw ? want : f_lastw;
            // 0 unless the window was moved back (1..3 elements; further back = past the panel: masked by k anyway)
            sh_fifo |= ((unsigned)((want - a_ptr[q]) >> 2) & 3u) << (4 * fifo_pos + 2 * q);
        }
        // advance (wave-uniform scalars only; past the last tile the state stays: duplicates are fetched)
        if (f_gt + 1 < n_tiles) {
            ++f_gt;
            if (++f_tk == f_nt) {
                f_tk = 0;
                if (f_panel == 0) {
                    f_rowp = rowp1; f_lastw = lastw1; f_nt = g.ktiles[1];
                } else {
                    f_rowp = rowp2; f_lastw = lastw2; f_nt = g.ktiles[2];
                }
                ++f_panel;
            }
        }
    };
    auto issue_a = [&](int slot) {
        float *dst = raw_s + slot * (TM * TK) + wave * 256;
#pragma unroll
        for (int q = 0; q < NA; ++q)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<gf4 *>(a_ptr[q]), (lds_void *)(dst + q * (NT * 4)), 16, 0, 0);
    };
    auto issue_b = [&](int tile_k, _Float16 *D) {
        typedef __attribute__((address_space(1))) const uint4 gu4;
        const uint4 *src = bsrc + (long)min(tile_k, n_tiles - 1) * (2 * BPL / 8);
        uint4 *d = reinterpret_cast<uint4 *>(D + 2 * APL) + wave * 64;
        constexpr int NB = (2 * BPL / 8) / NT;     // 16-byte DMAs per thread for the two planes: 2 (512 threads) or 4 (256)
#pragma unroll
        for (int q = 0; q < NB; ++q)
            __builtin_amdgcn_global_load_lds((gu4 *)(src + q * NT), (lds_void *)(d + q * NT), 16, 0, 0);
    };
    int st_k0 = 0, st_kp = 0;                      // the tile being staged: first k of this thread's windows, panel width
    unsigned st_sh = 0;
    auto plan_stage = [&]() {
        st_k0 = s_tk * TK + akc * EPT;
        st_kp = s_kp;
        st_sh = sh_fifo;
        sh_fifo >>= 4;
        if (s_gt + 1 < n_tiles) {                 // the same walk as plan_a's
            ++s_gt;
            if (++s_tk == s_nt) {
                s_tk = 0;
                if (s_panel == 0) {
                    s_kp = kw1; s_nt = g.ktiles[1];
                } else {
                    s_kp = kw2; s_nt = g.ktiles[2];
                }
                ++s_panel;
            }
        }
    };
    // The ring is read back with inline asm: a ds_read hipcc can see makes it drain EVERY LDS-DMA in flight first
    // (s_waitcnt vmcnt(0): it cannot tell which DMA wrote the bytes).  Its own counted lgkmcnt waits stay safe beside
    // an unseen LDS read (they can only wait for more than they need); the values are tied to ring_wait() below.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 rawv[NA];
    auto ring_read = [&](int slot) {
        const unsigned addr = (unsigned)(unsigned long)(lds_void *)(raw_s + slot * (TM * TK) + t * 4);
#pragma unroll
        for (int q = 0; q < NA; ++q)
            asm volatile("ds_read_b128 %0, %1" : "=v"(rawv[q]) : "v"(addr + q * (NT * 16)) : "memory");
    };
    auto ring_wait = [&]() {
        if constexpr (NA == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rawv[0]) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rawv[0]), "+v"(rawv[1]) :: "memory");
    };
    // One k step: the 12 MFMAs of the tile in S with the split of the NEXT tile (this thread's raw windows in rawv -> its
    // piece of the fp16 planes in D) cut into pieces that are pinned behind them: an MFMA occupies the matrix pipe for 32
    // cycles but the wave's issue port for 8 only, so 5-6 single-issue VALU instructions between two MFMAs are free
    // (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost') -- the compiler's own order puts the split behind the last
    // MFMA, where every instruction of it costs its full issue time.
#define LKG_PIN() __builtin_amdgcn_sched_barrier(0)
    // The step's first fragments (hi planes) are requested together with the ring read-back: ONE wait for LDS per step instead of
    // two in a row (ring, then fragments).
    f16x8 a0[2], b0[2];
    auto first_frags = [&](const _Float16 *S) {
        if constexpr (!PRE) {
            a0[0] = frag<BN>(S, wm * 64, lane);
            b0[0] = frag<BN>(S + 2 * APL, wn * WN, lane);
            b0[1] = frag<BN>(S + 2 * APL, wn * WN + 32, lane);
            a0[1] = frag<BN>(S, wm * 64 + 32, lane);
        }
    };
    // (the step's DMA requests were also tried INSIDE the MFMA block, behind its first two MFMAs: no measurable difference)
    auto step = [&](const _Float16 *S, _Float16 *D, bool do_stage) {
        // the other fragments are fetched where their registers become free (all eight up front would hold 32 registers at once)
        f16x8 a1[2], b1[2];
        auto fa = [&](int i, int pl) { return frag<BN>(S + pl * APL, wm * 64 + i * 32, lane); };
        auto fb = [&](int j, int pl) { return frag<BN>(S + 2 * APL + pl * BPL, wn * WN + j * 32, lane); };
        const _Float16 sc = (_Float16)(1.f / 2048.f);
        f16x8 ahs[2], bhs[2];
        float e[EPT];
        fp16x2 hi[EPT / 2], mid[EPT / 2];
        // ---- the pieces of the split (each a handful of VALU instructions per window)
        auto p_shift1 = [&]() {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const f32x4 v = rawv[q];
                const bool s1 = (st_sh >> (2 * q)) & 1;   // a barrel shifter of selects (no branch)
                e[4 * q] = s1 ? v[1] : v[0]; e[4 * q + 1] = s1 ? v[2] : v[1]; e[4 * q + 2] = s1 ? v[3] : v[2]; e[4 * q + 3] = s1 ? 0.f : v[3];
            }
        };
        auto p_shift2 = [&]() {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const bool s2 = (st_sh >> (2 * q)) & 2;
                const float w0 = e[4 * q], w1 = e[4 * q + 1], w2 = e[4 * q + 2], w3 = e[4 * q + 3];
                e[4 * q] = s2 ? w2 : w0; e[4 * q + 1] = s2 ? w3 : w1; e[4 * q + 2] = s2 ? 0.f : w2; e[4 * q + 3] = s2 ? 0.f : w3;
            }
        };
        auto p_mask = [&]() {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int kk = st_k0 + 4 * q;
#pragma unroll
                for (int u = 0; u < 4; ++u) e[4 * q + u] = kk + u < st_kp ? e[4 * q + u] : 0.f;
            }
        };
        auto p_hi = [&]() {
#pragma unroll
            for (int u = 0; u < EPT; u += 2) {
                e[u] = ldexpf(e[u], ea);
                e[u + 1] = ldexpf(e[u + 1], ea);
                hi[u / 2] = __builtin_amdgcn_cvt_pkrtz(e[u], e[u + 1]);
            }
        };
        auto p_res = [&]() {
            constexpr float rs = PRE ? 1.f : 2048.f;
#pragma unroll
            for (int u = 0; u < EPT; u += 2) {
                e[u] = (e[u] - (float)hi[u / 2][0]) * rs;
                e[u + 1] = (e[u + 1] - (float)hi[u / 2][1]) * rs;
            }
        };
        auto p_mid = [&]() {
#pragma unroll
            for (int u = 0; u < EPT; u += 2) mid[u / 2] = __builtin_amdgcn_cvt_pkrtz(e[u], e[u + 1]);
        };
        auto p_write = [&]() {
            _Float16 *pa = D + arow * TK + ((((akc * EPT) >> 3) ^ ((arow >> 4) & 1)) << 3) + ((akc * EPT) & 7);
            typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < EPT; q += 4) {
                const fp16x4 hv = {hi[q / 2][0], hi[q / 2][1], hi[q / 2 + 1][0], hi[q / 2 + 1][1]};
                const fp16x4 mv = {mid[q / 2][0], mid[q / 2][1], mid[q / 2 + 1][0], mid[q / 2 + 1][1]};
                *reinterpret_cast<fp16x4 *>(pa + q) = hv;
                *reinterpret_cast<fp16x4 *>(pa + APL + q) = mv;
            }
        };
#ifdef LKG_ABL_NO_MFMA      /* (ablation builds, tools/tall_ablation.sh: what is the k loop waiting for?) */
#define LKG_MFMA(C, A_, B_) asm volatile("" :: "v"(A_), "v"(B_))
#else
#define LKG_MFMA(C, A_, B_) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C, 0, 0, 0)
#endif
        LKG_PIN();
        // The wave's MFMA block runs at a raised priority: of the four waves on a SIMD (two per workgroup) the one that has its
        // fragments issues its MFMAs ahead of the others' address arithmetic and requests -- 2-3 % on every shape, measured in
        // both orders in one process (1 M x 300 x 300 1.255 -> 1.214 ms, gate 2.366 -> 2.327 ms).
#ifndef LKG_TALL_STEP_PRIO
#define LKG_TALL_STEP_PRIO 2
#endif
        __builtin_amdgcn_s_setprio(LKG_TALL_STEP_PRIO);
        if constexpr (PRE) {
            // 24 MFMAs: hi.hi, mid'.hi, hi.mid' over the wave's 2 x 4 blocks; a fragment is fetched where its registers
            // come free (peak: a hi 8 + b hi 16 + a mid' 8 + two b mid' 8), the split of the next tile rides behind them
            static_assert(!PRE || NJ == 4, "written out for 4 column blocks per wave");
            f16x8 ah[2], am[2], bh[4], bm[4];
            ah[0] = fa(0, 0); bh[0] = fb(0, 0); bh[1] = fb(1, 0); bh[2] = fb(2, 0); bh[3] = fb(3, 0); ah[1] = fa(1, 0);
            LKG_PIN();
            LKG_MFMA(acc[0][0], ah[0], bh[0]); am[0] = fa(0, 1); LKG_PIN();
            LKG_MFMA(acc[0][1], ah[0], bh[1]); am[1] = fa(1, 1); LKG_PIN();
            LKG_MFMA(acc[0][2], ah[0], bh[2]); if (do_stage) p_shift1(); LKG_PIN();
            LKG_MFMA(acc[0][3], ah[0], bh[3]); if (do_stage) p_shift2(); LKG_PIN();
            LKG_MFMA(acc[1][0], ah[1], bh[0]); if (do_stage) p_mask(); LKG_PIN();
            LKG_MFMA(acc[1][1], ah[1], bh[1]); LKG_PIN();
            LKG_MFMA(acc[1][2], ah[1], bh[2]); if (do_stage) p_hi(); LKG_PIN();
            LKG_MFMA(acc[1][3], ah[1], bh[3]); LKG_PIN();
            LKG_MFMA(acc[0][0], am[0], bh[0]); if (do_stage) p_res(); LKG_PIN();
            LKG_MFMA(acc[0][1], am[0], bh[1]); LKG_PIN();
            LKG_MFMA(acc[0][2], am[0], bh[2]); if (do_stage) p_mid(); LKG_PIN();
            LKG_MFMA(acc[0][3], am[0], bh[3]); LKG_PIN();
            LKG_MFMA(acc[1][0], am[1], bh[0]); bm[0] = fb(0, 1); if (do_stage) p_write(); LKG_PIN();
            LKG_MFMA(acc[1][1], am[1], bh[1]); bm[1] = fb(1, 1); LKG_PIN();
            LKG_MFMA(acc[1][2], am[1], bh[2]); bm[2] = fb(2, 1); LKG_PIN();
            LKG_MFMA(acc[1][3], am[1], bh[3]); bm[3] = fb(3, 1); LKG_PIN();
            LKG_MFMA(acc[0][0], ah[0], bm[0]); LKG_PIN();
            LKG_MFMA(acc[1][0], ah[1], bm[0]); LKG_PIN();
            LKG_MFMA(acc[0][1], ah[0], bm[1]); LKG_PIN();
            LKG_MFMA(acc[1][1], ah[1], bm[1]); LKG_PIN();
            LKG_MFMA(acc[0][2], ah[0], bm[2]); LKG_PIN();
            LKG_MFMA(acc[1][2], ah[1], bm[2]); LKG_PIN();
            LKG_MFMA(acc[0][3], ah[0], bm[3]); LKG_PIN();
            LKG_MFMA(acc[1][3], ah[1], bm[3]);
        } else if constexpr (ONE) {
            LKG_MFMA(acc[0][0], a0[0], b0[0]); ahs[0] = a0[0] * sc; b1[0] = fb(0, 1); LKG_PIN();
            LKG_MFMA(acc[0][1], a0[0], b0[1]); ahs[1] = a0[1] * sc; b1[1] = fb(1, 1); LKG_PIN();
            LKG_MFMA(acc[1][0], a0[1], b0[0]); bhs[0] = b0[0] * sc; LKG_PIN();
            LKG_MFMA(acc[1][1], a0[1], b0[1]); bhs[1] = b0[1] * sc; LKG_PIN();
            LKG_MFMA(acc[0][0], ahs[0], b1[0]); a1[0] = fa(0, 1); a1[1] = fa(1, 1); if (do_stage) p_shift1(); LKG_PIN();
            LKG_MFMA(acc[0][1], ahs[0], b1[1]); if (do_stage) p_shift2(); LKG_PIN();
            LKG_MFMA(acc[1][0], ahs[1], b1[0]); if (do_stage) p_mask(); LKG_PIN();
            LKG_MFMA(acc[1][1], ahs[1], b1[1]); if (do_stage) p_hi(); LKG_PIN();
            LKG_MFMA(acc[0][0], a1[0], bhs[0]); if (do_stage) p_res(); LKG_PIN();
            LKG_MFMA(acc[0][1], a1[0], bhs[1]); if (do_stage) p_mid(); LKG_PIN();
            LKG_MFMA(acc[1][0], a1[1], bhs[0]); if (do_stage) p_write(); LKG_PIN();
            LKG_MFMA(acc[1][1], a1[1], bhs[1]);
        } else {
            LKG_MFMA(acc[0][0], a0[0], b0[0]); b1[0] = fb(0, 1); LKG_PIN();
            LKG_MFMA(acc[0][1], a0[0], b0[1]); b1[1] = fb(1, 1); LKG_PIN();
            LKG_MFMA(acc[1][0], a0[1], b0[0]); a1[0] = fa(0, 1); LKG_PIN();
            LKG_MFMA(acc[1][1], a0[1], b0[1]); a1[1] = fa(1, 1); LKG_PIN();
            LKG_MFMA(cor[0][0], a0[0], b1[0]); if (do_stage) p_shift1(); LKG_PIN();
            LKG_MFMA(cor[0][1], a0[0], b1[1]); if (do_stage) p_shift2(); LKG_PIN();
            LKG_MFMA(cor[1][0], a0[1], b1[0]); if (do_stage) p_mask(); LKG_PIN();
            LKG_MFMA(cor[1][1], a0[1], b1[1]); if (do_stage) p_hi(); LKG_PIN();
            LKG_MFMA(cor[0][0], a1[0], b0[0]); if (do_stage) p_res(); LKG_PIN();
            LKG_MFMA(cor[0][1], a1[0], b0[1]); if (do_stage) p_mid(); LKG_PIN();
            LKG_MFMA(cor[1][0], a1[1], b0[0]); if (do_stage) p_write(); LKG_PIN();
            LKG_MFMA(cor[1][1], a1[1], b0[1]);
        }
        __builtin_amdgcn_s_setprio(0);
#undef LKG_MFMA
    };
#undef LKG_PIN
    auto stage_only = [&](_Float16 *D) {           // (prologue: the split of tile 0 alone)
        float e[EPT];
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const f32x4 v = rawv[q];
            const unsigned sh = st_sh >> (2 * q);
            const bool s1 = sh & 1, s2 = sh & 2;
            const float w0 = s1 ? v[1] : v[0], w1 = s1 ? v[2] : v[1], w2 = s1 ? v[3] : v[2], w3 = s1 ? 0.f : v[3];
            const float e0 = s2 ? w2 : w0, e1 = s2 ? w3 : w1, e2 = s2 ? 0.f : w2, e3 = s2 ? 0.f : w3;
            const int kk = st_k0 + 4 * q;
            e[4 * q] = kk < st_kp ? e0 : 0.f;
            e[4 * q + 1] = kk + 1 < st_kp ? e1 : 0.f;
            e[4 * q + 2] = kk + 2 < st_kp ? e2 : 0.f;
            e[4 * q + 3] = kk + 3 < st_kp ? e3 : 0.f;
        }
        _Float16 *pa = D + arow * TK + ((((akc * EPT) >> 3) ^ ((arow >> 4) & 1)) << 3) + ((akc * EPT) & 7);
#pragma unroll
        for (int q = 0; q < EPT; q += 4) {
            fp16x2 h0, m0_, h1, m1;
            split2<PRE>(ldexpf(e[q], ea), ldexpf(e[q + 1], ea), h0, m0_);
            split2<PRE>(ldexpf(e[q + 2], ea), ldexpf(e[q + 3], ea), h1, m1);
            typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
            const fp16x4 hv = {h0[0], h0[1], h1[0], h1[1]}, mv = {m0_[0], m0_[1], m1[0], m1[1]};
            *reinterpret_cast<fp16x4 *>(pa + q) = hv;
            *reinterpret_cast<fp16x4 *>(pa + APL + q) = mv;
        }
    };

#ifdef LKG_WS_STAMPS
    unsigned long long t_e_math = 0, t_e_put = 0, t_e_flush = 0;
#endif
    auto epilogue = [&](const long m0, const int n0) {
    // ---- epilogue (of the tile whose numbers are handed in: by then setup() may describe the next one).  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane
        // holds ONE column of 16 rows.  Written as it stands that is 16 dword stores per tile (two 128-byte runs per
        // instruction); instead every wave turns its tile through a private 4 KB of the (now free) staging LDS and stores
        // 16 bytes per lane, 8 rows x 128 B per instruction -- 4 stores per tile.
        float *ts = reinterpret_cast<float *>(smem) + wave * 1024;
        auto put = [&](const float(&v)[16]) {
    #pragma unroll
            for (int r = 0; r < 16; ++r) ts[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = v[r];
        };
        // rows row0 .. row0 + 31 (global), columns col0 .. col0 + 31 of `base` (n_cols wide): out = ts (+ beta * old)
        auto flush = [&](float *base, long ld, long row0, int col0, int n_cols, float beta) {
            const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
            const int col = col0 + 4 * (lane & 7);
            lkg_f32x4 tv[4];
            const unsigned ta = lds_addr_of(ts + (lane >> 3) * 32 + 4 * (lane & 7));
    #pragma unroll
            for (int q = 0; q < 4; ++q) lds_read16_issue(tv[q], ta + q * 1024);
            LKG_LDS_READS_DONE_4(tv[0], tv[1], tv[2], tv[3]);
    #pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long row = row0 + 8 * q + (lane >> 3);
                float4 v = make_float4(tv[q][0], tv[q][1], tv[q][2], tv[q][3]);
                if (row >= g.m) continue;
                float *dst = base + row * ld + col;
                if (vec && col + 3 < n_cols) {
                    if (beta != 0.f) {
                        const float4 o = *reinterpret_cast<const float4 *>(dst);
                        v.x = fmaf(beta, o.x, v.x); v.y = fmaf(beta, o.y, v.y); v.z = fmaf(beta, o.z, v.z); v.w = fmaf(beta, o.w, v.w);
                    }
                    // streaming (nontemporal) stores: C is far larger than the L2 and is not read again by this launch --
                    // 1 M x 256: K = 64 0.376 -> 0.309 ms, K = 256 0.706 -> 0.694 ms (LKG_EXTRA_HIPCC_FLAGS=-DLKG_TALL_PLAIN_STORE: A/B)
#ifdef LKG_TALL_PLAIN_STORE
                    *reinterpret_cast<float4 *>(dst) = v;
#else
                    typedef float nt4 __attribute__((ext_vector_type(4)));
                    const nt4 nv = {v.x, v.y, v.z, v.w};
#ifdef LKG_ABL_NO_STORE
                    if (g.m < 0)
#endif
                    __builtin_nontemporal_store(nv, reinterpret_cast<nt4 *>(dst));
#endif
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
    #pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (col + k < n_cols) dst[k] = beta != 0.f ? fmaf(beta, dst[k], e[k]) : e[k];
                }
            }
        };
        // the same for rows 16 h .. 16 h + 15 of the block (8 values per lane)
        auto put_half = [&](const float(&v)[8], int h) {
    #pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 8 * h + q;
                ts[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = v[q];
            }
        };
        auto flush_half = [&](float *base, long ld, long row0, int col0, int n_cols, int h) {
            const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
            const int col = col0 + 4 * (lane & 7);
            const unsigned ta = lds_addr_of(ts + (16 * h + (lane >> 3)) * 32 + 4 * (lane & 7));
    #pragma unroll
            for (int q = 2 * h; q < 2 * h + 2; ++q) {
                const long row = row0 + 8 * q + (lane >> 3);
                lkg_f32x4 tv;                              // (one at a time: the gate's epilogue has no 8 registers to spare)
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tv) : "v"(ta + (q - 2 * h) * 1024) : "memory");
                const float4 v = make_float4(tv[0], tv[1], tv[2], tv[3]);
                if (row >= g.m) continue;
                float *dst = base + row * ld + col;
                if (vec && col + 3 < n_cols) {
#ifdef LKG_TALL_PLAIN_STORE
                    *reinterpret_cast<float4 *>(dst) = v;
#else
                    typedef float nt4 __attribute__((ext_vector_type(4)));
                    const nt4 nv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(nv, reinterpret_cast<nt4 *>(dst));
#endif
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
    #pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (col + k < n_cols) dst[k] = e[k];
                }
            }
        };
        if constexpr (EPI == EPI_PLAIN) {
            float bias_v[NJ];
            int eb_v[NJ];
    #pragma unroll
            for (int j = 0; j < NJ; ++j) {
                eb_v[j] = eb_s[wn * WN + j * 32 + (lane & 31)];
                bias_v[j] = bias_s[wn * WN + j * 32 + (lane & 31)];
            }
    #pragma unroll
            for (int i = 0; i < 2; ++i)
    #pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col0 = n0 + wn * WN + j * 32;
                    if (col0 >= g.n) continue;                                // wave-uniform
                    const int lr0 = wm * 64 + i * 32 + 4 * (lane >> 5);       // row inside the tile
                    float out[16];
                    LKG_STAMP(a0_);
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        float v = acc[i][j][r];
                        if constexpr (!ONE) v = fmaf(cor[i][j][r], 1.f / 2048.f, v);
                        out[r] = g.alpha * ldexpf(v, -(ea_s[lr0 + dr] + eb_v[j])) + bias_v[j];
                    }
#ifdef LKG_WS_STAMPS
                    asm volatile("s_nop 0" :: "v"(out[15]), "v"(out[0]));
#endif
                    LKG_STAMP(a1_);
                    put(out);
                    LKG_STAMP(a2_);
                    flush(g.c, g.ldc, m0 + wm * 64 + i * 32, col0, g.n, g.beta);
                    LKG_STAMP(a3_);
                    LKG_STAMP_ADD(t_e_math, a0_, a1_); LKG_STAMP_ADD(t_e_put, a1_, a2_); LKG_STAMP_ADD(t_e_flush, a2_, a3_);
                    __builtin_amdgcn_sched_barrier(0);     // one 32x32 tile at a time: short live ranges next to 128 accumulators
                }
        } else if constexpr (EPI == EPI_ACTLN) {
            // K5 in the GEMM's epilogue (model.py:108-111, 161, 305): LeakyReLU -> LayerNorm -> dropout (+ the
            // L2-normalised copy).  The tile holds whole rows (n <= 256, one column tile), but in MFMA layout a row is spread
            // over the BN / WN waves of a row half -- so the tile goes through LDS in SLABS of 16 rows x 256 columns (inside the
            // area the other epilogues' transposes use): the waves that own a slab's rows write z = A B^T + bias into it row-major
            // (exactly what the plain epilogue would have stored), then ALL waves of the workgroup take four rows each and run
            // the row-wise kernel's own arithmetic on them -- one wave per row, 16 bytes per lane, the same lane <-> column map and
            // the same reduction tree as lkg_act_layernorm_fwd_f32: y, yn, mean and rstd are BIT-IDENTICAL to the unfused pair.
            // Registers: one 16-value block at a time, like the plain epilogue (three in-register designs spilled 80-230
            // VGPRs: hipcc kept per-row values, rewritten accumulators and their common subexpressions alive from pass to pass).
            // Stores: whole 1 KB rows per instruction.  Slabs of 16 rows (gamma / beta ride in LDS behind them): 16 workgroup
            // barriers per tile.
            static_assert(EPI != EPI_ACTLN || (ONE && BN == 256 && TM == 128), "the fused layer epilogue: one accumulator, 128 x 256 tiles");
            // The epilogue's own operands (19 dwords of kernel arguments) are read HERE, through an opaque pointer to the
            // kernel-argument segment: as fields of `g` they are loaded at kernel entry and sit in SGPRs through the k loop, the
            // scalar file overflows into VGPR lanes and the 128-VGPR budget with them (6 spilled VGPRs in the tile-opening code).
            // They are read ONCE, here, with scalar loads (the pointer keeps its constant address space): as a generic pointer
            // every use became a flat load + s_waitcnt vmcnt(0) in the row loop -- a wait for every store in flight.
            typedef __attribute__((address_space(4))) const TallArgs KArgs;
            KArgs *Lp = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(Lp));
            struct { float slope, ln_eps, norm_eps, drop_p; unsigned long long seed; const float *gamma, *beta_ln; float *yn; long ldyn; float *mean, *rstd; } Lv =
                {Lp->slope, Lp->ln_eps, Lp->norm_eps, Lp->drop_p, Lp->seed, Lp->gamma, Lp->beta_ln, Lp->yn, Lp->ldyn, Lp->mean, Lp->rstd};
            const auto *L = &Lv;
            constexpr int NWAVES = NT / 64, SROWS = 16, RPW = SROWS / NWAVES;  // rows of a slab per wave: 2 (8 waves)
            float *slab = reinterpret_cast<float *>(smem);                     // [16][BN], then gamma[BN], beta[BN]: 18 KB
            float *gb_s = slab + SROWS * BN;
            const int c4 = 4 * lane;                                           // this lane's four columns in the row phase
            if (wave < 2) {                                                    // gamma / beta of the layer: into LDS once per tile
                const float *src = wave == 0 ? L->gamma : L->beta_ln;            // (visible behind the first slab's barrier)
    #pragma unroll
                for (int k = 0; k < 4; ++k) gb_s[wave * BN + c4 + k] = c4 + k < g.n ? src[c4 + k] : 0.f;
            }
            const float inv_keep = L->drop_p > 0.f ? 1.f / (1.f - L->drop_p) : 1.f;
            const bool vec_y = g.c && (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.c) & 15) == 0);
            const bool vec_n = L->yn && (L->ldyn % 4 == 0) && ((reinterpret_cast<uintptr_t>(L->yn) & 15) == 0);
            auto slab_pass = [&](auto S) {
                // slab sl = rows 16 sl .. 16 sl + 15 of the tile: wave row sl / 4, block row (sl / 2) & 1, registers 8 h .. 8 h + 7
                constexpr int sl = decltype(S)::value, i = (sl >> 1) & 1, h = sl & 1;
                if (wm == (sl >> 2)) {                                         // (wave-uniform) this wave owns rows of the slab
    #pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int sc = wn * WN + j * 32 + (lane & 31);
                        const int eb_j = eb_s[sc];
                        const float bias_j = bias_s[sc];
    #pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int dr = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);           // row inside the slab
                            slab[dr * BN + sc] = g.alpha * ldexpf(acc[i][j][8 * h + q], -(ea_s[wm * 64 + i * 32 + 16 * h + dr] + eb_j)) + bias_j;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();
    #pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const int row = wave * RPW + rr;                           // row of the slab
                    const long grow = m0 + sl * SROWS + row;
                    lkg_f32x4 z4, gam4, bet4;
                    lds_read16_issue(z4, lds_addr_of(slab + row * BN + c4));
                    lds_read16_issue(gam4, lds_addr_of(gb_s + c4));
                    lds_read16_issue(bet4, lds_addr_of(gb_s + BN + c4));
                    LKG_LDS_READS_DONE_3(z4, gam4, bet4);
                    float a[4] = {z4[0], z4[1], z4[2], z4[3]};
                    const float gm[4] = {gam4[0], gam4[1], gam4[2], gam4[3]}, bt[4] = {bet4[0], bet4[1], bet4[2], bet4[3]};
                    float s_ = 0.f;
    #pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        a[k] = a[k] > 0.f ? a[k] : a[k] * L->slope;
                        s_ += c4 + k < g.n ? a[k] : 0.f;
                    }
                    const float mean = wave_sum(s_) / (float)g.n;
                    float q_ = 0.f;
    #pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float c = c4 + k < g.n ? a[k] - mean : 0.f;
                        q_ = fmaf(c, c, q_);
                    }
                    const float rstd = 1.f / sqrtf(wave_sum(q_) / (float)g.n + L->ln_eps);
                    const unsigned rkey = drop_row_key(L->seed, (unsigned long long)grow);
                    float nn = 0.f;
    #pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float o = (a[k] - mean) * rstd * gm[k] + bt[k];
                        if (L->drop_p > 0.f) o *= drop_scale(rkey, (unsigned)(c4 + k), L->drop_p, inv_keep);
                        a[k] = o;
                        nn += c4 + k < g.n ? o * o : 0.f;
                    }
                    const float inv = 1.f / fmaxf(sqrtf(wave_sum(nn)), L->norm_eps);
                    if (grow < g.m) {                                          // (wave-uniform)
                        if (lane == 0) {
                            L->mean[grow] = mean;
                            L->rstd[grow] = rstd;
                        }
                        if (g.c) {
                            float *dst = g.c + grow * g.ldc + c4;
                            if (vec_y && c4 + 3 < g.n) {
                                const lkg_f32x4 o4 = {a[0], a[1], a[2], a[3]};
                                __builtin_nontemporal_store(o4, reinterpret_cast<lkg_f32x4 *>(dst));
                            } else {
    #pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (c4 + k < g.n) dst[k] = a[k];
                            }
                        }
                        if (L->yn) {
                            float *dst = L->yn + grow * L->ldyn + c4;
                            if (vec_n && c4 + 3 < g.n) {
                                const lkg_f32x4 o4 = {a[0] * inv, a[1] * inv, a[2] * inv, a[3] * inv};
                                __builtin_nontemporal_store(o4, reinterpret_cast<lkg_f32x4 *>(dst));
                            } else {
    #pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (c4 + k < g.n) dst[k] = a[k] * inv;
                            }
                        }
                    }
                }
                __syncthreads();                                               // the next slab overwrites this one
            };
            slab_pass(std::integral_constant<int, 0>{});
            slab_pass(std::integral_constant<int, 1>{});
            slab_pass(std::integral_constant<int, 2>{});
            slab_pass(std::integral_constant<int, 3>{});
            slab_pass(std::integral_constant<int, 4>{});
            slab_pass(std::integral_constant<int, 5>{});
            slab_pass(std::integral_constant<int, 6>{});
            slab_pass(std::integral_constant<int, 7>{});
        } else {
            // gate: of every pair of tile column blocks, block 2 pr holds g and block 2 pr + 1 holds z of the SAME output columns
            const int d = g.n / 2;
    #pragma unroll
            for (int pr = 0; pr < NJ / 2; ++pr) {
            const int col0 = (n0 >> 1) + wn * (WN / 2) + pr * 32;
            const int col = col0 + (lane & 31);
            if (col0 < d) {                                                   // wave-uniform
                const int cc = min(col, d - 1);                               // lanes past d compute on a clamped column, never stored
                const int ebg = eb_s[wn * WN + pr * 64 + (lane & 31)], ebz = eb_s[wn * WN + pr * 64 + 32 + (lane & 31)];
                const float bg = bias_s[wn * WN + pr * 64 + (lane & 31)], bz = bias_s[wn * WN + pr * 64 + 32 + (lane & 31)];
    #pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int lr0 = wm * 64 + i * 32 + 4 * (lane >> 5);
                    const long row0 = m0 + wm * 64 + i * 32;
                    // 16 rows at a time (registers r = 8 h .. 8 h + 7 are rows 16 h .. 16 h + 15 of the 32 x 32 block): half the
                    // temporaries of a whole block at once, next to 64 accumulators
    #pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float xv[8];
    #pragma unroll
                        // x[rows of the tile, this lane's output column]: re-read (two 128-byte runs per instruction; +n d 4 bytes of
                        // traffic).  Round 2 kept these values in a 66 KB LDS stash filled while the x panel was staged: one workgroup
                        // per CU then, and the k loop of a lone workgroup leaves the matrix pipe 38 % busy -- two per CU: GateMul
                        // forward 1 M x (256+2+300) 2.73 -> 2.46 ms, forward + backward 8.5 -> 7.9 ms.
                        for (int q = 0; q < 8; ++q) {
                            const int r = 8 * h + q;
                            xv[q] = g.x[min(m0 + lr0 + (r & 3) + 8 * (r >> 2), g.m - 1) * g.ldx + cc];
                        }
                        // The x loads are in flight while tanh / sigmoid are evaluated; x is first touched BEHIND that arithmetic: the
                        // wait the compiler puts there is s_waitcnt vmcnt(0) (loads and stores share the counter and retire out of
                        // order with each other), which also waits for the previous half's stores -- by then a half's arithmetic old.
                        float ov[8], gv[8], zv[8];
    #pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int r = 8 * h + q;
                            const int dr = (r & 3) + 8 * (r >> 2);
                            const int e = ea_s[lr0 + dr];
                            float gs = acc[i][2 * pr][r], zs = acc[i][2 * pr + 1][r];
                            if constexpr (!ONE) {
                                gs = fmaf(cor[i][2 * pr][r], 1.f / 2048.f, gs);
                                zs = fmaf(cor[i][2 * pr + 1][r], 1.f / 2048.f, zs);
                            }
                            const float gp = ldexpf(gs, -(e + ebg)) + bg;
                            const float zp = ldexpf(zs, -(e + ebz)) + bz;
                            gv[q] = tanh_fast(gp);
                            zv[q] = sigmoid_fast(zp);
                        }
                        __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                        for (int q = 0; q < 8; ++q) ov[q] = fmaf(zv[q], gv[q] - xv[q], xv[q]);   // (1 - z) x + z g
                        put_half(ov, h);
                        flush_half(g.c, g.ldc, row0, col0, d, h);
                        if (g.g_out) {
                            put_half(gv, h);
                            flush_half(g.g_out, g.ldg, row0, col0, d, h);
                        }
                        if (g.z_out) {
                            put_half(zv, h);
                            flush_half(g.z_out, g.ldz, row0, col0, d, h);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
        }

    };

    // One barrier per step, no branch in the loop (the accumulators never meet a control-flow join).  Step t: wait for
    // this wave's B(t) and A(t+1) pieces + its plane writes, barrier (tile t complete in buffer t & 1, nobody still reads
    // the other one) -> LDS-DMA of B(t+1) into the other buffer and of A(t+3) into the ring slot tile t just left ->
    // fragment reads + 12 MFMAs of tile t, with the split of tile t+1 (ring -> planes of the other buffer) in their
    // shadow.  Past the last tile the staged / fetched tiles are duplicates of the last one (in bounds, never read).
#ifdef LKG_WS_STAMPS
    unsigned long long t_wait = 0, t_bar = 0, t_issue = 0, t_step = 0, t_epi = 0, t_pro = 0, n_steps = 0, t_tail = 0, t_after = 0;
#endif
#if defined(LKG_WS_STAMPS) || defined(LKG_CLOCK_STAMP)
    const unsigned long long all0_ = __builtin_amdgcn_s_memtime(), real0_ = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef LKG_WS_STAMPS
#define LKG_WAIT_BARRIER(N, BETWEEN)                                                                         \
    do {                                                                                                     \
        LKG_STAMP(w0_);                                                                                      \
        asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory");                                     \
        LKG_STAMP(w1_);                                                                                      \
        BETWEEN;                                                                                             \
        asm volatile("s_barrier" ::: "memory");                                                              \
        LKG_STAMP(w2_);                                                                                      \
        LKG_STAMP_ADD(t_wait, w0_, w1_); LKG_STAMP_ADD(t_bar, w1_, w2_);                                     \
    } while (0)
#else
#define LKG_WAIT_BARRIER(N, BETWEEN)                                                                         \
    do {                                                                                                     \
        asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory");                                     \
        BETWEEN;                                                                                             \
        asm volatile("s_barrier" ::: "memory");                                                              \
    } while (0)
#endif
    // A tile's FIRST k step reads buffer 1: the epilogue's transposes use the first 32 KB of the staging LDS (buffer 0 and the A
    // planes of buffer 1), so the next tile's first B planes -- requested before that epilogue -- land beyond them.
    auto request_first = [&](bool issue) {         // the loads that open a tile: B(0) -> buffer 1, A(0..2) -> the ring
        reset_walks();                             // (issue = false: the same walk, nothing requested -- re-derives the state)
        if (issue) issue_b(0, smem + BUF);
        plan_a(0); if (issue) issue_a(0);
        plan_a(1); if (issue) issue_a(1);
        plan_a(2); if (issue) issue_a(2);
    };
    tile_regs(slot);
    remap(n0);
    tile_prefetch();
    request_first(true);
    bool first_tile = true;
    while (true) {
        LKG_STAMP(p0_);
        if (!first_tile) {                         // (workgroup-uniform) re-derive what the epilogue was not asked to carry
            rethread();
            tile_regs(slot);
            remap(n0);
            request_first(false);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][j][r] = 0.f;
                    if constexpr (!ONE) cor[i][j][r] = 0.f;
                }
        if (first_tile) {                          // the oldest loads only: A(1), A(2) stay in flight
            if constexpr (NA == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            // requested before the previous epilogue, whose stores are younger: vmcnt counts loads and stores in one counter
            // and they retire out of order with each other, so only vmcnt(0) says that the loads are in
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        first_tile = false;
        tile_open();                               // (the scalars' DMAs are the oldest requests: in by now)
        plan_stage();
        ring_read(0);                              // (its own pieces: no barrier between the DMA and the read-back)
        ring_wait();
        stage_only(smem + BUF);
        LKG_STAMP(p1_);
        LKG_STAMP_ADD(t_pro, p0_, p1_);
        // (the k loop as a macro: a wave of padding columns -- LIVE = false -- runs its share of the split and nothing else; written
        //  out twice for the kernels that skip, and exactly as before for the others -- wrapped in a lambda it cost the gate's
        //  kernel 11 spilled VGPRs)
#ifdef LKG_ABL_NO_DMA
#define LKG_LOOP_ISSUE()
#else
#define LKG_LOOP_ISSUE() do { issue_b(gt + 1, nxt); issue_a(slot_f); } while (0)
#endif
#ifdef LKG_ABL_NO_STAGE
#define LKG_LOOP_STEP(LIVE) step(cur, nxt, false)
#else
#define LKG_LOOP_STEP(LIVE) do { if constexpr (LIVE) step(cur, nxt, true); else stage_only(nxt); } while (0)
#endif
#ifdef LKG_WS_STAMPS
#define LKG_LOOP_COUNT() do { asm volatile("s_nop 0" :: "v"(acc[1][NJ - 1][0])); ++n_steps; } while (0)
#else
#define LKG_LOOP_COUNT()
#endif
#define LKG_K_LOOP(LIVE)                                                                                                  \
        {                                                                                                                 \
            int slot_f = 0, slot_s = 1;      /* ring slots: tile gt + 3 goes where tile gt was; tile gt + 1 is staged */  \
            for (int gt = 0; gt < n_tiles; ++gt) {                                                                        \
                plan_a(2);                                                                                                \
                plan_stage();                                                                                             \
                _Float16 *cur = smem + ((gt + 1) & 1) * BUF, *nxt = smem + (gt & 1) * BUF;                                \
                /* the ring read-back of tile gt + 1 (this thread's OWN pieces: landed once its vmcnt wait is over) is    \
                   issued in front of the barrier: its LDS latency passes while the wave waits for the others */          \
                if constexpr (NA == 1) LKG_WAIT_BARRIER(1, ring_read(slot_s)); else LKG_WAIT_BARRIER(2, ring_read(slot_s)); \
                LKG_STAMP(k0_);                                                                                           \
                LKG_LOOP_ISSUE();                                                                                         \
                if constexpr (LIVE) first_frags(cur);   /* (in front of the requests: no faster, and a register more) */  \
                ring_wait();                                                                                              \
                LKG_STAMP(k1_);                                                                                           \
                LKG_LOOP_STEP(LIVE);                                                                                      \
                LKG_LOOP_COUNT();                                                                                         \
                LKG_STAMP(k2_);                                                                                           \
                LKG_STAMP_ADD(t_issue, k0_, k1_); LKG_STAMP_ADD(t_step, k1_, k2_);                                        \
                slot_f = slot_f == RING - 1 ? 0 : slot_f + 1;                                                             \
                slot_s = slot_s == RING - 1 ? 0 : slot_s + 1;                                                             \
            }                                                                                                             \
        }
        if constexpr (SKIPS) {
            if (act) LKG_K_LOOP(true) else LKG_K_LOOP(false)
        } else LKG_K_LOOP(true)
#undef LKG_K_LOOP
#undef LKG_LOOP_COUNT
#undef LKG_LOOP_STEP
#undef LKG_LOOP_ISSUE
        LKG_STAMP(q0_);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the duplicate tiles still in flight target the ring / buffers

        const long e_m0 = m0;                                  // (the epilogue's tile; tile_regs() moves on to the next one)
        const int e_n0 = n0;
        __syncthreads();                                       // every wave is done reading the staging buffers and the ring
        rethread();                                            // (what follows re-derives its coordinates: none live across the k loop)
        slot += slot_stride;
        const bool more = slot < xcd_count;                    // (workgroup-uniform)
        if (more) {
            tile_regs(slot);
            tile_prefetch();
            request_first(true);
        }
        LKG_STAMP(e0_);
        LKG_STAMP_ADD(t_tail, q0_, e0_);
        epilogue(e_m0, e_n0);
        LKG_STAMP(e1_);
        LKG_STAMP_ADD(t_epi, e0_, e1_);
        if (!more) break;
        __syncthreads();     // the transposes are done with buffer 1's A planes and the tile's LDS scalars: the next tile moves in
        LKG_STAMP(e2_);
        LKG_STAMP_ADD(t_after, e1_, e2_);
    }
#ifdef LKG_WS_STAMPS
    if (EPI == EPI_PLAIN && g.z_out && (threadIdx.x & 63) == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(g.z_out);
        atomicAdd(dbg + 16, t_wait); atomicAdd(dbg + 17, t_bar); atomicAdd(dbg + 18, t_issue); atomicAdd(dbg + 19, t_step);
        atomicAdd(dbg + 20, t_epi); atomicAdd(dbg + 21, t_pro); atomicAdd(dbg + 22, n_steps);
        atomicAdd(dbg + 23, t_e_math); atomicAdd(dbg + 24, t_e_put); atomicAdd(dbg + 25, t_e_flush);
        atomicAdd(dbg + 26, t_tail); atomicAdd(dbg + 27, t_after); atomicAdd(dbg + 28, __builtin_amdgcn_s_memtime() - all0_);
        atomicAdd(dbg + 29, 1ull);
        atomicAdd(dbg + 30, __builtin_amdgcn_s_memrealtime() - real0_);        // 100 MHz: the in-kernel clock = cycles / this * 100 MHz
    }
#elif defined(LKG_CLOCK_STAMP)      /* (the ONLY stamps of this build: two at kernel entry, two at its end -- MI355X_MICROARCH.md 'DVFS give-back' item 6) */
    if (EPI == EPI_PLAIN && g.z_out && (threadIdx.x & 63) == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(g.z_out);
        atomicAdd(dbg + 28, __builtin_amdgcn_s_memtime() - all0_);
        atomicAdd(dbg + 29, 1ull);
        atomicAdd(dbg + 30, __builtin_amdgcn_s_memrealtime() - real0_);
    }
#endif
#undef LKG_WAIT_BARRIER
}


// =====================================================================================================================
// Wave-specialised form (variant 4, "ws"): ONE 8-wave workgroup per CU, 128 x 256 tiles, two kinds of waves.
//
// Why.  `s_waitcnt vmcnt(N)` retires a wave's memory operations in ISSUE order, loads and LDS-DMAs alike.  In the forms
// above every wave requests both operands, so the wait for B's planes of the next step (an L2 hit, requested one step
// ago) also waits for every A window requested before them: whatever the depth of A's ring, a window has TWO steps to
// come back from HBM.  Measured (tools/tall_variants_micro.py, round 4): the k loop costs ~1800 cycles per tile and step
// on a CU whichever way its waves are arranged (8 waves of 64 x 64 or 4 of 64 x 128) -- the matrix pipe 43 % busy, 2.2 TB/s
// of A: a latency bound, bytes in flight = 2 steps x 8 KB per workgroup.  Here the two streams live in different waves:
//
//   waves 4-7 "loaders"   the A stream.  Every thread requests its own two 16-byte windows of the raw f32 tile (and its
//                         row's maximum) by LDS-DMA into a ring of WS_R slots, R - 1 items ahead -- across tile
//                         boundaries: the ring keeps filling while the compute waves store the previous tile --, reads
//                         its own pieces back, splits them (hi / prescaled mid, split2<true>) and writes its piece of
//                         the fp16 planes of the NEXT step.  They issue nothing but A requests, so their counted wait
//                         (vmcnt(3 (R - 2))) leaves R - 2 items in flight: 48 KB per CU with R = 8.
//   waves 0-3 "compute"   64 x 128 each (one per SIMD): B's ready-made planes by LDS-DMA into a ring of 3 buffers two steps
//                         ahead (their only requests besides the epilogue's stores), fragment reads, 24 MFMAs per step,
//                         the epilogue.  No VALU work of the split competes with their MFMA issue.
//
// One workgroup barrier per k step orders everything: before barrier g the loaders have written A's planes of step g and
// every compute wave has its pieces of B(g) in; after it the compute waves read planes g while the loaders overwrite the
// buffers of step g - 1.  Items (tile, k tile) form ONE stream over the tiles of the workgroup; a row tile's column tiles
// (the gate: 2) are consecutive items, so the second one reads A from the L2 its own CU has just filled.
constexpr int WS_R = 8, WS_NB = 3;
constexpr int WS_APL = TM * TK, WS_BPL = 256 * TK;                 // halves per plane
constexpr int WS_SLOT_BYTES = TM * TK * 4 + 1024;                  // raw A tile + one row maximum per loader thread
constexpr int WS_OFF_A = 0;                                        // 2 x [hi | mid']             16 KB
constexpr int WS_OFF_B = WS_OFF_A + 2 * 2 * WS_APL * 2;            // 3 x [hi | mid']             48 KB
constexpr int WS_OFF_RING = WS_OFF_B + WS_NB * 2 * WS_BPL * 2;     // R slots                     72 KB
constexpr int WS_OFF_TS = WS_OFF_RING + WS_R * WS_SLOT_BYTES;      // 4 x 4 KB epilogue transposes 16 KB
constexpr int WS_OFF_EA = WS_OFF_TS + 4 * 4096;                    // int[2][128] row exponents, by tile parity
constexpr int WS_OFF_ST = WS_OFF_EA + 2 * TM * 4;                  // float[3][2][128] row statistics (EPI_ACTLN)
constexpr int WS_OFF_COL = WS_OFF_ST + 3 * 2 * TM * 4;             // [4][256] per-column scalars (EPI_ACTLN: exponent, bias, gamma, beta)
constexpr int WS_LDS_BYTES = WS_OFF_COL + 4 * 256 * 4;
static_assert(WS_LDS_BYTES <= 160 * 1024, "one workgroup per CU: at most 160 KB of LDS");

struct WsWalk {            // position in the item stream: row tile (index into this workgroup's stripe), column tile, k tile
    int rt, tn, panel, tk, gt;
};

// Position in the item stream, advanced by one item.  (Plain functions with value / reference parameters instead of lambdas
// capturing by reference: with the loader written as lambdas hipcc kept 39 closures and their captured variables in scratch.)
struct WsShape {
    int n_rt, tiles_n, kt_total, kt0, kt1, kt2;
};
// (the shape BY VALUE: a select between fields of a struct behind a reference becomes a runtime index into a private copy)
__device__ __forceinline__ void ws_advance(WsWalk &w, const WsShape sh) {         // (past the last item: stays there -- duplicates)
    const int n_rt = sh.n_rt, tiles_n = sh.tiles_n, kt_total = sh.kt_total, kt0 = sh.kt0, kt1 = sh.kt1, kt2 = sh.kt2;
    if (w.rt == n_rt - 1 && w.tn == tiles_n - 1 && w.gt == kt_total - 1) return;
    if (w.gt + 1 < kt_total) {
        ++w.gt;
        int nt = kt0;
        if (w.panel == 1) nt = kt1;
        if (w.panel == 2) nt = kt2;
        if (++w.tk == nt) { w.tk = 0; ++w.panel; }
    } else {
        w.gt = 0; w.tk = 0; w.panel = 0;
        if (++w.tn == tiles_n) { w.tn = 0; ++w.rt; }
    }
}

// ======================================================================================= loaders: the A stream (waves 4-7)
// (a loader's step must stay under the compute waves' 768 MFMA cycles: everything per item is incremental -- row pointers per
//  row tile, a 64-byte step per k tile -- and the rare cases, windows moved back at the end of the LAST row and the partial
//  last k tile of a panel, sit behind wave-uniform branches)
template <int EPI>
__device__ __forceinline__ void ws_loader(const TallArgs &g, char *lds, const int lt, const int lw, const int n_rt,
                                          const int n_items, const int tile0, const int slot_stride) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef __attribute__((address_space(1))) const float4u gf4;
    typedef __attribute__((address_space(1))) const unsigned gu1;
    const WsShape shp{n_rt, g.tiles_n, g.ktiles_total, g.ktiles[0], g.ktiles[1], g.ktiles[2]};
    const int arow = lt >> 1, akc = lt & 1;                            // this thread's row of the tile / 8-float chunk of the k tile
    // (absent panels: the launcher leaves their fields zero; their addresses are computed and never used)
    const unsigned long pa0 = reinterpret_cast<unsigned long>(g.a[0]);
    const unsigned long pa1 = reinterpret_cast<unsigned long>(g.a[1]);
    const unsigned long pa2 = reinterpret_cast<unsigned long>(g.a[2]);
    const long ld0 = g.lda[0], ld1 = g.lda[1], ld2 = g.lda[2];
    const int kw0 = g.ka[0], kw1 = g.ka[1], kw2 = g.ka[2];
    // the last valid 16-byte window of every panel (of the LAST row: only windows past it are moved back)
    const unsigned long lastw0 = pa0 + 4ul * (unsigned long)((g.m - 1) * ld0 + kw0 - 4);
    const unsigned long lastw1 = pa1 + 4ul * (unsigned long)((g.m - 1) * ld1 + kw1 - 4);
    const unsigned long lastw2 = pa2 + 4ul * (unsigned long)((g.m - 1) * ld2 + kw2 - 4);
    unsigned long rowp0 = 0, rowp1 = 0, rowp2 = 0;                     // this thread's chunk of its row, per panel (per row tile)
    WsWalk fw{0, 0, 0, 0, 0}, sw{0, 0, 0, 0, 0}, cw{0, 0, 0, 0, 0};   // issue / stage walks; cw: the item the compute waves work on
    unsigned sh_fifo = 0;                                              // 4 bits per item in flight: how far its two windows were moved back
    int ea = 0;

    // item fw -> ring slot
#define LKG_WS_ISSUE(SLOT)                                                                                                 \
    do {                                                                                                                   \
        char *dst_ = lds + WS_OFF_RING + (SLOT) * WS_SLOT_BYTES;                                                           \
        if (fw.gt == 0 && fw.tn == 0) {          /* (wave-uniform) a new row tile: row pointers, row maximum */            \
            const long row_ = min((long)(tile0 + fw.rt * slot_stride) * TM + arow, g.m - 1);                               \
            rowp0 = pa0 + 4ul * (unsigned long)(row_ * ld0 + akc * 8);                                                     \
            rowp1 = pa1 + 4ul * (unsigned long)(row_ * ld1 + akc * 8);                                                     \
            rowp2 = pa2 + 4ul * (unsigned long)(row_ * ld2 + akc * 8);                                                     \
            __builtin_amdgcn_global_load_lds((gu1 *)(g.a_rowmax + row_), (lds_void *)(dst_ + 8192 + lw * 256), 4, 0, 0);   \
        }                                                                                                                  \
        const unsigned long rowp_ = fw.panel == 0 ? rowp0 : (fw.panel == 1 ? rowp1 : rowp2);                               \
        const unsigned long lastw_ = fw.panel == 0 ? lastw0 : (fw.panel == 1 ? lastw1 : lastw2);                           \
        const unsigned long want_ = rowp_ + (unsigned long)(fw.tk * (TK * 4));                                             \
        const unsigned long p0_ = want_ < lastw_ ? want_ : lastw_;                                                         \
        const unsigned long p1_ = want_ + 16 < lastw_ ? want_ + 16 : lastw_;                                               \
        sh_fifo |= (((unsigned)((want_ - p0_) >> 2) & 3u) | (((unsigned)((want_ + 16 - p1_) >> 2) & 3u) << 2))             \
                   << (4 * (WS_R - 2));                                                                                    \
        __builtin_amdgcn_global_load_lds(reinterpret_cast<gf4 *>(p0_), (lds_void *)(dst_ + lw * 1024), 16, 0, 0);          \
        __builtin_amdgcn_global_load_lds(reinterpret_cast<gf4 *>(p1_), (lds_void *)(dst_ + 4096 + lw * 1024), 16, 0, 0);   \
        ws_advance(fw, shp);                                                                                               \
    } while (0)

    // item sw: ring slot -> this thread's piece of the planes of buffer ABUF
#define LKG_WS_STAGE(SLOT, ABUF)                                                                                           \
    do {                                                                                                                   \
        const unsigned base_ = (unsigned)(unsigned long)(lds_void *)(lds + WS_OFF_RING + (SLOT) * WS_SLOT_BYTES);          \
        f32x4 v0, v1;                                                                                                      \
        /* (inline asm: a ds_read hipcc can see makes it wait for EVERY LDS-DMA in flight) */                              \
        asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(base_ + lt * 16) : "memory");                                  \
        asm volatile("ds_read_b128 %0, %1" : "=v"(v1) : "v"(base_ + 4096 + lt * 16) : "memory");                           \
        if (sw.gt == 0 && sw.tn == 0) {          /* (wave-uniform) a new row tile: this thread's row exponent */           \
            float rmv_;                                                                                                    \
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rmv_) : "v"(base_ + 8192 + lt * 4) : "memory"); \
            ea = scale_exponent(rmv_);                                                                                     \
        }                                                                                                                  \
        if (sw.gt == 0 && akc == 0)              /* the tile's row exponents for the compute waves' epilogue */            \
            reinterpret_cast<int *>(lds + WS_OFF_EA)[((sw.rt * g.tiles_n + sw.tn) & 1) * TM + arow] = ea;                   \
        const unsigned sh_ = sh_fifo & 15u;                                                                                \
        sh_fifo >>= 4;                                                                                                     \
        const int kw_ = sw.panel == 0 ? kw0 : (sw.panel == 1 ? kw1 : kw2);                                                 \
        const int k0_ = sw.tk * TK + akc * 8;                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1) :: "memory");                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        float e0 = v0[0], e1 = v0[1], e2 = v0[2], e3 = v0[3], e4 = v1[0], e5 = v1[1], e6 = v1[2], e7 = v1[3];              \
        if (__builtin_amdgcn_ballot_w64(sh_ != 0) != 0) {   /* (rare: the end of the panel's last row) windows moved back:   \
            element u of a window moved back by s elements sits at v[u + s]; past the loaded four: beyond the panel (0) */   \
            const int s0 = sh_ & 3, s1 = sh_ >> 2;                                                                         \
            e0 = s0 == 0 ? v0[0] : (s0 == 1 ? v0[1] : (s0 == 2 ? v0[2] : v0[3]));                                          \
            e1 = s0 == 0 ? v0[1] : (s0 == 1 ? v0[2] : (s0 == 2 ? v0[3] : 0.f));                                            \
            e2 = s0 == 0 ? v0[2] : (s0 == 1 ? v0[3] : 0.f);                                                                \
            e3 = s0 == 0 ? v0[3] : 0.f;                                                                                    \
            e4 = s1 == 0 ? v1[0] : (s1 == 1 ? v1[1] : (s1 == 2 ? v1[2] : v1[3]));                                          \
            e5 = s1 == 0 ? v1[1] : (s1 == 1 ? v1[2] : (s1 == 2 ? v1[3] : 0.f));                                            \
            e6 = s1 == 0 ? v1[2] : (s1 == 1 ? v1[3] : 0.f);                                                                \
            e7 = s1 == 0 ? v1[3] : 0.f;                                                                                    \
        }                                                                                                                  \
        if (sw.tk * TK + TK > kw_) {             /* (wave-uniform) the panel's partial last k tile: columns past it are 0 */ \
            e0 = k0_ + 0 < kw_ ? e0 : 0.f; e1 = k0_ + 1 < kw_ ? e1 : 0.f; e2 = k0_ + 2 < kw_ ? e2 : 0.f;                    \
            e3 = k0_ + 3 < kw_ ? e3 : 0.f; e4 = k0_ + 4 < kw_ ? e4 : 0.f; e5 = k0_ + 5 < kw_ ? e5 : 0.f;                    \
            e6 = k0_ + 6 < kw_ ? e6 : 0.f; e7 = k0_ + 7 < kw_ ? e7 : 0.f;                                                  \
        }                                                                                                                  \
        fp16x2 h0, h1, h2, h3, m0_, m1_, m2_, m3_;                                                                         \
        split2<true>(ldexpf(e0, ea), ldexpf(e1, ea), h0, m0_);                                                             \
        split2<true>(ldexpf(e2, ea), ldexpf(e3, ea), h1, m1_);                                                             \
        split2<true>(ldexpf(e4, ea), ldexpf(e5, ea), h2, m2_);                                                             \
        split2<true>(ldexpf(e6, ea), ldexpf(e7, ea), h3, m3_);                                                             \
        _Float16 *pa_ = reinterpret_cast<_Float16 *>(lds + WS_OFF_A) + (ABUF) * (2 * WS_APL) + arow * TK +                 \
                        ((akc ^ ((arow >> 4) & 1)) << 3);                                                                  \
        typedef __fp16 fp16x8 __attribute__((ext_vector_type(8)));                                                         \
        const fp16x8 hv_ = {h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};                                       \
        const fp16x8 mv_ = {m0_[0], m0_[1], m1_[0], m1_[1], m2_[0], m2_[1], m3_[0], m3_[1]};                               \
        *reinterpret_cast<fp16x8 *>(pa_) = hv_;                                                                            \
        *reinterpret_cast<fp16x8 *>(pa_ + WS_APL) = mv_;                                                                   \
        ws_advance(sw, shp);                                                                                               \
    } while (0)

    // 2 window requests per item (+ 1 row maximum per row tile): "all but the 12 youngest" leaves WS_R - 2 items in flight --
    // a row maximum among them only makes the wait ask for a little more than it needs
#define LKG_WS_WAIT_A() asm volatile("s_waitcnt vmcnt(12)" ::: "memory")
    static_assert(2 * (WS_R - 2) == 12, "the counted wait above");
#pragma unroll
    for (int i = 0; i < WS_R - 1; ++i) {
        sh_fifo >>= 4;                                                 // (the prologue fills the queue from the top: entry i ends at position i)
        LKG_WS_ISSUE(i);
    }
    LKG_WS_WAIT_A();
    LKG_WS_STAGE(0, 0);
    int slot_issue = WS_R - 1, slot_stage = 1;
#ifdef LKG_WS_STAMPS
    unsigned long long t_bar = 0, t_issue = 0, t_mem = 0, t_stage = 0;
#endif
    for (int it = 0; it < n_items; ++it) {
        LKG_STAMP(s0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this thread's plane writes of item `it`
        asm volatile("s_barrier" ::: "memory");                        // ---- barrier `it`
        LKG_STAMP(s1);
        LKG_WS_ISSUE(slot_issue);                                      // item it + R - 1 into the slot item it - 1 left
        LKG_STAMP(s2);
        LKG_WS_WAIT_A();                                               // item it + 1 is in
        LKG_STAMP(s3);
        LKG_WS_STAGE(slot_stage, (it + 1) & 1);
        LKG_STAMP(s4);
        LKG_STAMP_ADD(t_bar, s0, s1); LKG_STAMP_ADD(t_issue, s1, s2); LKG_STAMP_ADD(t_mem, s2, s3); LKG_STAMP_ADD(t_stage, s3, s4);
        if constexpr (EPI == EPI_ACTLN) {                              // the compute waves' six statistics exchanges of a tile end
            if (cw.gt == shp.kt_total - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int b6 = 0; b6 < 6; ++b6) asm volatile("s_barrier" ::: "memory");       // (three per row half)
            }
            ws_advance(cw, shp);
        }
        slot_issue = slot_issue == WS_R - 1 ? 0 : slot_issue + 1;
        slot_stage = slot_stage == WS_R - 1 ? 0 : slot_stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         // nothing may land in the LDS of a finished workgroup
#ifdef LKG_WS_STAMPS
    if (EPI == EPI_PLAIN && g.z_out && (lt & 63) == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(g.z_out);
        atomicAdd(dbg + 8, t_bar); atomicAdd(dbg + 9, t_issue); atomicAdd(dbg + 10, t_mem); atomicAdd(dbg + 11, t_stage);
        atomicAdd(dbg + 12, (unsigned long long)n_items);
    }
#endif
#undef LKG_WS_WAIT_A
#undef LKG_WS_ISSUE
#undef LKG_WS_STAGE
}


template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_tall_ws_kernel(TallArgs g) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    char *lds = reinterpret_cast<char *>(smem);
    typedef __attribute__((address_space(3))) void lds_void;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // ---- this workgroup's row tiles: workgroup b lives on XCD b & 7 and walks every (gridDim.x / 8)-th row tile of that
    // XCD's contiguous range (the same stripe as the forms above, over ROW tiles)
    const int cpx = g.tiles_m >> 3, rem = g.tiles_m & 7, xcd = blockIdx.x & 7;
    const int xcd_first = xcd * cpx + min(xcd, rem), xcd_count = cpx + (xcd < rem ? 1 : 0);
    const int slot_stride = max(1, (int)gridDim.x >> 3), slot0 = blockIdx.x >> 3;
    if (slot0 >= xcd_count) return;                                    // (workgroup-uniform)
    const int n_rt = (xcd_count - slot0 + slot_stride - 1) / slot_stride;
    const int kt_total = g.ktiles_total;
    const int n_items = n_rt * g.tiles_n * kt_total;
    const WsShape shp{n_rt, g.tiles_n, kt_total, g.ktiles[0], g.ktiles[1], g.ktiles[2]};
    auto advance = [&](WsWalk &w) __attribute__((always_inline)) { ws_advance(w, shp); };
    auto row0_of = [&](int rt) __attribute__((always_inline)) { return (long)(xcd_first + slot0 + rt * slot_stride) * TM; };
#define LKG_WS_BARRIER() asm volatile("s_barrier" ::: "memory")

    if (wave >= 4) {
        ws_loader<EPI>(g, lds, t - 256, wave - 4, n_rt, n_items, xcd_first + slot0, slot_stride);
        return;
    }

    // =========================================================================================== compute: B stream + MFMA
    const int wm_k = wave >> 1, wn_k = wave & 1;                      // (the k loop's copies; the epilogue re-derives its own)
    typedef __attribute__((address_space(1))) const uint4 gu4;
    WsWalk bw{0, 0, 0, 0, 0}, cw{0, 0, 0, 0, 0};
    // item bw: 16 KB of ready-made planes, 4 requests per wave -- issued one by one between the step's MFMAs (a request
    // holds the wave's issue port for ~60-100 cycles: four in a row in front of the MFMAs would idle the matrix pipe)
    const uint4 *b_src = nullptr;
    uint4 *b_dst = nullptr;
    auto plan_b = [&](int buf) __attribute__((always_inline)) {
        b_src = reinterpret_cast<const uint4 *>(g.bp) + ((long)bw.tn * kt_total + bw.gt) * (2 * WS_BPL / 8) + t;
        b_dst = reinterpret_cast<uint4 *>(lds + WS_OFF_B + buf * (2 * WS_BPL * 2)) + wave * 64;
        advance(bw);
    };
    auto issue_b_piece = [&](int q) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((gu4 *)(b_src + q * 256), (lds_void *)(b_dst + q * 256), 16, 0, 0);
    };
    auto issue_b = [&](int buf) __attribute__((always_inline)) {
        plan_b(buf);
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_b_piece(q);
    };
    f32x16 acc[2][4];
    // The epilogue's coordinates, re-derived per tile from an OPAQUE copy of threadIdx.x: everything computed from them --
    // LDS addresses of 32 rows, column indices, the transposes' addresses -- is then per-tile work that hipcc cannot hoist
    // out of the persistent loop, where it would sit in registers (and spill: 102 VGPRs with the LayerNorm epilogue) across
    // a k loop that has none to spare.
    int lane_e = lane, wm = wm_k, wn = wn_k;
    float *ts = reinterpret_cast<float *>(lds + WS_OFF_TS) + wave * 1024;
    auto rethread = [&]() __attribute__((always_inline)) {
        int t_o = threadIdx.x;
        asm volatile("" : "+v"(t_o));
        lane_e = t_o & 63;
        wm = (t_o >> 6) >> 1;
        wn = (t_o >> 6) & 1;
        ts = reinterpret_cast<float *>(lds + WS_OFF_TS) + (t_o >> 6) * 1024;
    };
    // one 32 x 32 block through the wave's private 4 KB: every lane then stores 16 bytes, 8 rows x 128 B per instruction
    auto put = [&](const float(&v)[16]) {
        const int lane = lane_e;
#pragma unroll
        for (int r = 0; r < 16; ++r) ts[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = v[r];
    };
    auto flush = [&](float *base, long ld, long row0, int col0, int n_cols, float beta) __attribute__((always_inline)) {
        const int lane = lane_e;
        const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
        const int col = col0 + 4 * (lane & 7);
        lkg_f32x4 tv[4];
        const unsigned ta = lds_addr_of(ts + (lane >> 3) * 32 + 4 * (lane & 7));
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_read16_issue(tv[q], ta + q * 1024);
        LKG_LDS_READS_DONE_4(tv[0], tv[1], tv[2], tv[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long row = row0 + 8 * q + (lane >> 3);
            float4 v = make_float4(tv[q][0], tv[q][1], tv[q][2], tv[q][3]);
            if (row >= g.m) continue;
            float *dst = base + row * ld + col;
            if (vec && col + 3 < n_cols) {
                if (beta != 0.f) {
                    const float4 o = *reinterpret_cast<const float4 *>(dst);
                    v.x = fmaf(beta, o.x, v.x); v.y = fmaf(beta, o.y, v.y); v.z = fmaf(beta, o.z, v.z); v.w = fmaf(beta, o.w, v.w);
                }
                typedef float nt4 __attribute__((ext_vector_type(4)));
                const nt4 nv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(nv, reinterpret_cast<nt4 *>(dst));
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (col + k < n_cols) dst[k] = beta != 0.f ? fmaf(beta, dst[k], e[k]) : e[k];
            }
        }
    };
    if constexpr (EPI == EPI_ACTLN) {
        // one column tile: the per-column scalars are the same for every tile -- into LDS once (visible behind barrier 0),
        // read in the epilogue where they are needed instead of riding in 16 registers through the tile's last k step
        float *cs = reinterpret_cast<float *>(lds + WS_OFF_COL);
        const int c_ = min(t, g.n - 1);
        reinterpret_cast<int *>(cs)[t] = g.eb[t];
        cs[256 + t] = g.bias ? g.bias[c_] : 0.f;
        cs[512 + t] = g.gamma[c_];
        cs[768 + t] = g.beta_ln[c_];
    }
    issue_b(0);
    issue_b(1);
    int abuf = 0, bbuf = 0, bnext = 2;
    bool stores_pending = false;
#ifdef LKG_WS_STAMPS
    unsigned long long t_mem = 0, t_bar = 0, t_step = 0, t_epi = 0;
#endif
    // The accumulators are never zeroed: a tile's first eight MFMAs take the constant 0 as their C operand.  (Zeroed under
    // "first k tile" at the loop head, the old accumulators stay live through the epilogue on the path hipcc cannot rule out --
    // and an epilogue that rewrites them, the LayerNorm one, spilled 100 VGPRs.)
    for (int it = 0; it < n_items; ++it) {
        LKG_STAMP(c0);
        // this wave's pieces of B(it) are in: everything but the 4 requests of B(it + 1) -- after an epilogue only vmcnt(0)
        // says so (its stores are younger and retire out of order with the loads)
        if (stores_pending) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        stores_pending = false;
        LKG_STAMP(c1);
        LKG_WS_BARRIER();                                              // ---- barrier `it`
        LKG_STAMP(c2);
        const bool last_k = cw.gt == kt_total - 1;                     // (wave-uniform)
        // the epilogue's per-column scalars: requested under the tile's last 24 MFMAs
        const int n0 = cw.tn * 256;
        int eb_v[4];
        float bias_v[4];
        if (last_k) rethread();
        if (last_k && EPI != EPI_ACTLN) {
            const int lane = lane_e;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sc = n0 + wn * 128 + j * 32 + (lane & 31);   // stacked column of the tile
                eb_v[j] = g.eb[sc];
                if constexpr (EPI == EPI_GATE) {     // stacked column s of the tile: group (s / 32) & 1, output column (s / 64) * 32 + s % 32
                    const int s_ = sc - n0, d = g.n / 2, oc = min((n0 >> 1) + (s_ >> 6) * 32 + (s_ & 31), d - 1);
                    bias_v[j] = g.bias ? g.bias[((s_ >> 5) & 1) * d + oc] : 0.f;
                } else {
                    bias_v[j] = g.bias ? g.bias[min(sc, g.n - 1)] : 0.f;
                }
            }
        }
        plan_b(bnext);                                                 // item it + 2 into the buffer B(it - 1) left: requested below
        {
            const _Float16 *Ap = reinterpret_cast<const _Float16 *>(lds + WS_OFF_A) + abuf * (2 * WS_APL);
            const _Float16 *Bp = reinterpret_cast<const _Float16 *>(lds + WS_OFF_B) + bbuf * (2 * WS_BPL);
            auto fa = [&](int i, int pl) __attribute__((always_inline)) { return frag<256>(Ap + pl * WS_APL, wm_k * 64 + i * 32, lane); };
            auto fb = [&](int j, int pl) __attribute__((always_inline)) { return frag<256>(Bp + pl * WS_BPL, wn_k * 128 + j * 32, lane); };
#define LKG_MFMA(C, A_, B_) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C, 0, 0, 0)
#define LKG_MFMA0(C, A_, B_) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, zero16, 0, 0, 0)
#define LKG_PIN() __builtin_amdgcn_sched_barrier(0)
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f16x8 ah[2], am[2], bh[4], bm[4];
            ah[0] = fa(0, 0); bh[0] = fb(0, 0); bh[1] = fb(1, 0); bh[2] = fb(2, 0); bh[3] = fb(3, 0); ah[1] = fa(1, 0);
            LKG_PIN();
            if (cw.gt == 0) {                                           // (wave-uniform) a tile's first step: C = 0
                LKG_MFMA0(acc[0][0], ah[0], bh[0]); am[0] = fa(0, 1); LKG_PIN();
                LKG_MFMA0(acc[0][1], ah[0], bh[1]); am[1] = fa(1, 1); LKG_PIN();
                LKG_MFMA0(acc[0][2], ah[0], bh[2]); LKG_PIN();
                LKG_MFMA0(acc[0][3], ah[0], bh[3]); LKG_PIN();
                LKG_MFMA0(acc[1][0], ah[1], bh[0]); LKG_PIN();
                LKG_MFMA0(acc[1][1], ah[1], bh[1]); LKG_PIN();
                LKG_MFMA0(acc[1][2], ah[1], bh[2]); LKG_PIN();
                LKG_MFMA0(acc[1][3], ah[1], bh[3]); LKG_PIN();
            } else {
                LKG_MFMA(acc[0][0], ah[0], bh[0]); am[0] = fa(0, 1); LKG_PIN();
                LKG_MFMA(acc[0][1], ah[0], bh[1]); am[1] = fa(1, 1); LKG_PIN();
                LKG_MFMA(acc[0][2], ah[0], bh[2]); LKG_PIN();
                LKG_MFMA(acc[0][3], ah[0], bh[3]); LKG_PIN();
                LKG_MFMA(acc[1][0], ah[1], bh[0]); LKG_PIN();
                LKG_MFMA(acc[1][1], ah[1], bh[1]); LKG_PIN();
                LKG_MFMA(acc[1][2], ah[1], bh[2]); LKG_PIN();
                LKG_MFMA(acc[1][3], ah[1], bh[3]); LKG_PIN();
            }
            LKG_MFMA(acc[0][0], am[0], bh[0]); bm[0] = fb(0, 1); LKG_PIN();
            LKG_MFMA(acc[0][1], am[0], bh[1]); bm[1] = fb(1, 1); LKG_PIN();
            LKG_MFMA(acc[0][2], am[0], bh[2]); issue_b_piece(0); LKG_PIN();
            LKG_MFMA(acc[0][3], am[0], bh[3]); LKG_PIN();
            LKG_MFMA(acc[1][0], am[1], bh[0]); bm[2] = fb(2, 1); LKG_PIN();
            LKG_MFMA(acc[1][1], am[1], bh[1]); bm[3] = fb(3, 1); LKG_PIN();
            LKG_MFMA(acc[1][2], am[1], bh[2]); issue_b_piece(1); LKG_PIN();
            LKG_MFMA(acc[1][3], am[1], bh[3]); LKG_PIN();
            LKG_MFMA(acc[0][0], ah[0], bm[0]); LKG_PIN();
            LKG_MFMA(acc[1][0], ah[1], bm[0]); issue_b_piece(2); LKG_PIN();
            LKG_MFMA(acc[0][1], ah[0], bm[1]); LKG_PIN();
            LKG_MFMA(acc[1][1], ah[1], bm[1]); LKG_PIN();
            LKG_MFMA(acc[0][2], ah[0], bm[2]); issue_b_piece(3); LKG_PIN();
            LKG_MFMA(acc[1][2], ah[1], bm[2]); LKG_PIN();
            LKG_MFMA(acc[0][3], ah[0], bm[3]); LKG_PIN();
            LKG_MFMA(acc[1][3], ah[1], bm[3]);
#undef LKG_MFMA
#undef LKG_MFMA0
#undef LKG_PIN
        }
        abuf ^= 1;
        bbuf = bbuf == WS_NB - 1 ? 0 : bbuf + 1;
        bnext = bnext == WS_NB - 1 ? 0 : bnext + 1;
#ifdef LKG_WS_STAMPS
        asm volatile("s_nop 0" :: "v"(acc[1][3][0]));                  // (the step's last MFMA has issued)
#endif
        LKG_STAMP(c3);
        LKG_STAMP_ADD(t_mem, c0, c1); LKG_STAMP_ADD(t_bar, c1, c2); LKG_STAMP_ADD(t_step, c2, c3);
        if (last_k) {
            // ------------------------------------------------------------------------------------------------ epilogue
            const int lane = lane_e;                                   // (the opaque copy: see rethread())
            const long m0 = row0_of(cw.rt);
            const int *ea_s = reinterpret_cast<const int *>(lds + WS_OFF_EA) + ((cw.rt * g.tiles_n + cw.tn) & 1) * TM;
            if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col0 = n0 + wn * 128 + j * 32;
                        if (col0 >= g.n) continue;                                // wave-uniform
                        const int lr0 = wm * 64 + i * 32 + 4 * (lane >> 5);
                        float out[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            out[r] = g.alpha * ldexpf(acc[i][j][r], -(ea_s[lr0 + (r & 3) + 8 * (r >> 2)] + eb_v[j])) + bias_v[j];
                        put(out);
                        flush(g.c, g.ldc, m0 + wm * 64 + i * 32, col0, g.n, g.beta);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else if constexpr (EPI == EPI_GATE) {
                // of every pair of tile column blocks, block 2 pr holds g and block 2 pr + 1 holds z of the SAME output columns
                const int d = g.n / 2;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int col0 = (n0 >> 1) + wn * 64 + pr * 32;
                    if (col0 >= d) continue;                                      // wave-uniform
                    const int cc = min(col0 + (lane & 31), d - 1);                // lanes past d compute on a clamped column, never stored
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int lr0 = wm * 64 + i * 32 + 4 * (lane >> 5);
                        const long row0 = m0 + wm * 64 + i * 32;
                        float xv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r)                              // x[rows of the block, this lane's output column]
                            xv[r] = g.x[min(m0 + lr0 + (r & 3) + 8 * (r >> 2), g.m - 1) * g.ldx + cc];
                        float ov[16], gv[16], zv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int e = ea_s[lr0 + (r & 3) + 8 * (r >> 2)];
                            const float gp = ldexpf(acc[i][2 * pr][r], -(e + eb_v[2 * pr])) + bias_v[2 * pr];
                            const float zp = ldexpf(acc[i][2 * pr + 1][r], -(e + eb_v[2 * pr + 1])) + bias_v[2 * pr + 1];
                            gv[r] = tanh_fast(gp);
                            zv[r] = sigmoid_fast(zp);
                            ov[r] = fmaf(zv[r], gv[r] - xv[r], xv[r]);            // (1 - z) x + z g
                        }
                        put(ov);
                        flush(g.c, g.ldc, row0, col0, d, 0.f);
                        if (g.g_out) {
                            put(gv);
                            flush(g.g_out, g.ldg, row0, col0, d, 0.f);
                        }
                        if (g.z_out) {
                            put(zv);
                            flush(g.z_out, g.ldz, row0, col0, d, 0.f);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // LeakyReLU -> LayerNorm -> dropout (+ the L2-normalised copy): the tile holds whole rows (n <= 256), a row's
                // 256 columns live in the two waves wn = 0 / 1 of a row half -- three sums per row cross them through LDS
                float *st = reinterpret_cast<float *>(lds + WS_OFF_ST);            // [3][2][TM]
                const float *cs = reinterpret_cast<const float *>(lds + WS_OFF_COL);
                float gam_v[4], bet_v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sc = wn * 128 + j * 32 + (lane & 31);
                    eb_v[j] = reinterpret_cast<const int *>(cs)[sc];
                    bias_v[j] = cs[256 + sc];
                    gam_v[j] = cs[512 + sc];
                    bet_v[j] = cs[768 + sc];
                }
                const float inv_n = 1.f / (float)g.n;
                auto row_of = [&](int i, int r) __attribute__((always_inline)) { return wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); };
                // a row's partial sum over this wave's 128 columns -> LDS (one lane per row writes); after the barrier either
                // wave reads both halves back where it needs the total (nothing per-row lives in registers between the passes)
                auto share = [&](float v, int which, int i, int r) __attribute__((always_inline)) {
                    v = group_sum<32>(v);
                    if ((lane & 31) == r) st[(which * 2 + wn) * TM + row_of(i, r)] = v;
                };
                auto total = [&](int which, int i, int r) __attribute__((always_inline)) {
                    return st[(which * 2) * TM + row_of(i, r)] + st[(which * 2 + 1) * TM + row_of(i, r)];
                };
                auto sync = [&]() __attribute__((always_inline)) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    LKG_WS_BARRIER();
                };
                // One ROW HALF of the wave's tile at a time (i = 0, 1: 32 rows x 128 columns, four accumulators): all three sums,
                // then the stores -- the rewritten values of one half live beside the untouched accumulators of the other, not
                // beside a second copy of all eight (which spilled 100 VGPRs).  Six barriers per tile; the loaders match them.
                const float inv_keep = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
                auto half = [&](auto I) __attribute__((always_inline)) {            // (a generic lambda called with 0 and 1: `i` stays a constant -- the body is too
                    constexpr int i = decltype(I)::value;   //  large for hipcc to unroll as a loop, and a runtime i indexes scratch)
                    float row_a[16], row_b[16];                                    // per-row values of the half (sums, then mean / rstd)
#pragma unroll
                    for (int r = 0; r < 16; ++r) row_a[r] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool live = wn * 128 + j * 32 + (lane & 31) < g.n;
                        f32x16 blk = acc[i][j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int e = ea_s[wm * 64 + i * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2)];
                            float v = ldexpf(blk[r], -(e + eb_v[j])) + bias_v[j];
                            v = v > 0.f ? v : v * g.slope;
                            v = live ? v : 0.f;
                            blk[r] = v;
                            row_a[r] += v;
                        }
                        acc[i][j] = blk;
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) share(row_a[r], 0, i, r);
                    sync();                                                        // row sums -> means
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        row_a[r] = total(0, i, r) * inv_n;
                        row_b[r] = 0.f;
                        const long grow = m0 + row_of(i, r);
                        if (wn == 0 && (lane & 31) == r && grow < g.m) g.mean[grow] = row_a[r];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                      // the CENTRED values go back into the accumulators: the means
                        const bool live = wn * 128 + j * 32 + (lane & 31) < g.n;       // are not needed (in registers) after this pass
                        f32x16 blk = acc[i][j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float c = live ? blk[r] - row_a[r] : 0.f;
                            blk[r] = c;
                            row_b[r] = fmaf(c, c, row_b[r]);
                        }
                        acc[i][j] = blk;
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) share(row_b[r], 1, i, r);
                    sync();                                                        // centred squares -> 1 / std
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        row_b[r] = 1.f / sqrtf(total(1, i, r) * inv_n + g.ln_eps);
                        const long grow = m0 + row_of(i, r);
                        if (wn == 0 && (lane & 31) == r && grow < g.m) g.rstd[grow] = row_b[r];
                        row_a[r] = 0.f;                                 // (from here on: the row's sum of squares of y)
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = wn * 128 + j * 32 + (lane & 31);
                        f32x16 blk = acc[i][j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float o = blk[r] * row_b[r] * gam_v[j] + bet_v[j];
                            if (g.drop_p > 0.f)       // (the row key is recomputed per element: 16 keys would not fit beside the rest)
                                o *= drop_scale(drop_row_key(g.seed, (unsigned long long)(m0 + row_of(i, r))), (unsigned)col, g.drop_p, inv_keep);
                            o = col < g.n ? o : 0.f;
                            blk[r] = o;
                            row_a[r] = fmaf(o, o, row_a[r]);
                        }
                        acc[i][j] = blk;
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) share(row_a[r], 2, i, r);
                    sync();                                                        // |y|^2 -> the normalised copy's scale
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col0 = wn * 128 + j * 32;
                        if (col0 >= g.n) continue;                                // wave-uniform
                        float out[16];
                        if (g.c) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) out[r] = acc[i][j][r];
                            put(out);
                            flush(g.c, g.ldc, m0 + wm * 64 + i * 32, col0, g.n, 0.f);
                        }
                        if (g.yn) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) out[r] = acc[i][j][r] * (1.f / fmaxf(sqrtf(total(2, i, r)), g.norm_eps));
                            put(out);
                            flush(g.yn, g.ldyn, m0 + wm * 64 + i * 32, col0, g.n, 0.f);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                half(std::integral_constant<int, 0>{});
                half(std::integral_constant<int, 1>{});
            }
            stores_pending = true;
            LKG_STAMP(c4);
            LKG_STAMP_ADD(t_epi, c3, c4);
        }
        advance(cw);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef LKG_WS_STAMPS
    if (EPI == EPI_PLAIN && g.z_out && lane == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(g.z_out);
        atomicAdd(dbg + 0, t_mem); atomicAdd(dbg + 1, t_bar); atomicAdd(dbg + 2, t_step); atomicAdd(dbg + 3, t_epi);
        atomicAdd(dbg + 4, (unsigned long long)n_items);
    }
#endif
#undef LKG_WS_BARRIER
}

// out[i] = max_j |x[i, j]|   (accumulate: max with the value already there).  One wave per row.
__global__ __launch_bounds__(256) void row_absmax_kernel(long n, int d, const float *__restrict__ x, long ldx,
                                                         float *__restrict__ out, int accumulate, int vec) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float m = 0.f;
    if (vec) {
        const float4 *p = reinterpret_cast<const float4 *>(x + i * ldx);
        for (int c = lane; c < d / 4; c += 64) {
            const float4 v = p[c];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (int c = lane; c < d; c += 64) m = fmaxf(m, fabsf(x[i * ldx + c]));
    }
    m = wave_max(m);
    if (lane == 0) out[i] = accumulate ? fmaxf(out[i], m) : m;
}

inline int total_ktiles(int n_panels, const int32_t *ka) {
    int tt = 0;
    for (int p = 0; p < n_panels; ++p) tt += (ka[p] + TK - 1) / TK;
    return tt;
}
// tile width and number of column tiles: the gate's stacked columns are interleaved in blocks of 32 (g, z, g, z ...), so
// its stacked extent is 64 per 32 output columns
// variant: 0 "256x2" (two accumulators, 128 x 256 tiles), 1 "128x1", 2 "256x1" (one accumulator, 8 waves of 64 x 64),
// 3 "256x1w" (one accumulator, prescaled mids, 4 waves of 64 x 128), 4 "ws" (wave-specialised: one 8-wave workgroup per CU,
// loader waves for the A stream, compute waves of 64 x 128 for B + MFMA; the only form of the act + LayerNorm epilogue;
// outputs of at most 128 columns without that epilogue stay on "128x1"), 5 "256r" (256-row tiles: 8 waves of 64 x 128, one
// workgroup per CU -- B's planes cross the CU's vector memory pipe once per 256 rows).  Chosen PER CALL: bits 8-15 of the `epilogue`
// argument hold variant + 1 (0 = the library's default), so a test or a tool runs any variant next to any other in one
// process and nothing in the environment selects code (LKG_TALL_VARIANT is gone).
inline int variant_of(int epilogue_arg) {
    const int v = (epilogue_arg >> 8) & 0xff;
    return v == 0 ? LKG_TALL_DEFAULT_VARIANT : v - 1;
}
inline void geometry(int n, int epilogue_arg, int &bn, int &tiles_n) {
    const int v = variant_of(epilogue_arg), epilogue = epilogue_arg & 0xff;
    if (epilogue == EPI_GATE) {
        bn = v == 1 ? 128 : 256;
        tiles_n = ((n / 2 + 31) / 32 * 64 + bn - 1) / bn;
    } else {
        bn = ((n <= 128 && epilogue != EPI_ACTLN) || v == 1) ? 128 : 256;
        tiles_n = (n + bn - 1) / bn;
    }
}

struct LnExtra {             // the act + LayerNorm epilogue's operands (lkg_linear_act_layernorm_fwd_f32)
    float slope, ln_eps, norm_eps, drop_p;
    unsigned long long seed;
    const float *gamma, *beta;
    float *yn;
    int64_t ldyn;
    float *mean, *rstd;
};

}  // namespace

extern "C" int lkg_row_absmax_f32(int64_t n, int32_t d, const float *x, int64_t ldx, float *out, int32_t accumulate,
                                  void *stream) {
    LKG_REQUIRE(n >= 0 && d > 0 && ldx >= d, "lkg_row_absmax_f32: bad sizes");
    if (n == 0) return LKG_OK;
    LKG_REQUIRE(x && out, "lkg_row_absmax_f32: null pointer");
    const int vec = (d % 4 == 0 && ldx % 4 == 0 && lkg_aligned16(x)) ? 1 : 0;
    hipLaunchKernelGGL(row_absmax_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (long)n, d,
                       x, (long)ldx, out, accumulate, vec);
    LKG_CHECK_LAUNCH("lkg_row_absmax_f32");
    return LKG_OK;
}

extern "C" int64_t lkg_gemm_tall_workspace(int32_t n, int32_t n_panels, const int32_t *ka, int32_t epilogue) {
    if (n <= 0 || n_panels <= 0 || n_panels > MAX_PANELS || !ka) return -1;
    int bn, tiles_n;
    geometry(n, epilogue, bn, tiles_n);
    return (long)tiles_n * total_ktiles(n_panels, ka) * (2L * bn * TK) * 2 + (long)tiles_n * bn * 4 + 256;
}

static int tall_call(int64_t m, int32_t n, int32_t n_panels, const float *const *a, const int64_t *lda,
                     const int32_t *ka, const float *a_rowmax, int32_t n_groups, const float *const *b,
                     const int64_t *ldb, int32_t trans_b, float alpha, float beta, float *c, int64_t ldc,
                     const float *bias, int32_t epilogue, const float *gate_x, int64_t ld_x, float *gate_g,
                     int64_t ld_g, float *gate_z, int64_t ld_z, void *workspace, int64_t workspace_bytes,
                     void *stream, const LnExtra *ln) {
    const int32_t epilogue_arg = epilogue;          // bits 0-7: the epilogue, bits 8-15: variant + 1 (0 = default)
    epilogue = epilogue_arg & 0xff;
    LKG_REQUIRE((epilogue_arg >> 16) == 0 && ((epilogue_arg >> 8) & 0xff) <= 6, "lkg_gemm_tall_f32: unknown variant in the "
                "epilogue argument (0x%x)", epilogue_arg);
    LKG_REQUIRE(m >= 0 && n > 0 && n_panels >= 1 && n_panels <= MAX_PANELS, "lkg_gemm_tall_f32: bad sizes");
    LKG_REQUIRE(epilogue == EPI_PLAIN || epilogue == EPI_GATE || (epilogue == EPI_ACTLN && ln),
                "lkg_gemm_tall_f32: unknown epilogue %d", epilogue);
    LKG_REQUIRE(epilogue != EPI_ACTLN || (n_panels <= 2 || ((epilogue_arg >> 8) & 0xff) == 5),
                "lkg_linear_act_layernorm_fwd_f32: at most two K-panels");
    LKG_REQUIRE(epilogue != EPI_ACTLN || (n <= 256 && alpha == 1.f && beta == 0.f && ln->gamma && ln->beta && ln->mean &&
                                          ln->rstd && (c || ln->yn) && (!ln->yn || ln->ldyn >= n) && ln->drop_p >= 0.f &&
                                          ln->drop_p < 1.f),
                "lkg_linear_act_layernorm_fwd_f32: at most 256 output columns, gamma / beta / mean / rstd, an output, 0 <= p < 1");
    LKG_REQUIRE(n_groups == (epilogue == EPI_GATE ? 2 : 1), "lkg_gemm_tall_f32: the gate stacks 2 weight groups, a plain "
                "product 1");
    if (m == 0) return LKG_OK;
    LKG_REQUIRE(a && lda && ka && a_rowmax && b && ldb && (c || epilogue == EPI_ACTLN) && workspace, "lkg_gemm_tall_f32: null pointer");
    const int d_out = epilogue == EPI_GATE ? n / 2 : n;
    LKG_REQUIRE(epilogue != EPI_GATE || (n % 2 == 0 && gate_x && ld_x >= d_out && beta == 0.f && alpha == 1.f),
                "lkg_gemm_tall_f32: gate epilogue needs x, an even stacked width, alpha = 1, beta = 0");
    LKG_REQUIRE(epilogue != EPI_GATE || (gate_x == a[0] && ld_x == lda[0] && ka[0] == d_out),
                "lkg_gemm_tall_f32: the gate blends its FIRST K-panel (gate_x must be a[0], %d wide)", d_out);
    LKG_REQUIRE(!c || ldc >= d_out, "lkg_gemm_tall_f32: ldc %lld smaller than the output width %d", (long long)ldc, d_out);
    const int64_t need = lkg_gemm_tall_workspace(n, n_panels, ka, epilogue_arg);
    LKG_REQUIRE(workspace_bytes >= need, "lkg_gemm_tall_f32: workspace of %lld bytes is smaller than the %lld required",
                (long long)workspace_bytes, (long long)need);
    hipStream_t s = (hipStream_t)stream;
    int bn, tiles_n;
    geometry(n, epilogue_arg, bn, tiles_n);
    const int variant = variant_of(epilogue_arg);
    const bool ws = variant == 4 && bn == 256;                                // wave-specialised
    const bool tall256 = variant == 5 && bn == 256 && !ws;                    // 256-row tiles, 8 waves of 64 x 128
    const bool wide = (variant == 3 && bn == 256) || ws || tall256;           // prescaled mid planes
    const int tm_rows = tall256 ? 256 : TM;
    TallArgs g{};
    BDesc bd{};
    g.m = m; g.n = n; g.n_panels = n_panels;
    bd.n_groups = n_groups; bd.n_panels = n_panels; bd.rows_per_group = d_out; bd.trans_b = trans_b;
    bd.interleave = epilogue == EPI_GATE ? 1 : 0;
    for (int p = 0; p < n_panels; ++p) {
        LKG_REQUIRE(ka[p] > 0 && a[p] && lda[p] >= ka[p], "lkg_gemm_tall_f32: bad A panel %d", p);
        g.a[p] = a[p]; g.lda[p] = lda[p]; g.ka[p] = ka[p]; g.ktiles[p] = (ka[p] + TK - 1) / TK;
        bd.k[p] = ka[p];
        for (int gr = 0; gr < n_groups; ++gr) {
            const float *bp = b[gr * n_panels + p];
            const int64_t ld = ldb[gr * n_panels + p];
            LKG_REQUIRE(bp && ld >= (trans_b ? ka[p] : d_out), "lkg_gemm_tall_f32: bad B block (group %d, panel %d)", gr, p);
            bd.ptr[gr][p] = bp; bd.ld[gr][p] = ld;
        }
    }
    g.ktiles_total = total_ktiles(n_panels, ka);
    g.tiles_m = (int)((m + tm_rows - 1) / tm_rows);
    g.tiles_n = tiles_n;
    LKG_REQUIRE((long)g.tiles_m * g.tiles_n < INT32_MAX, "lkg_gemm_tall_f32: too many tiles");
    _Float16 *planes = reinterpret_cast<_Float16 *>(workspace);
    int *eb = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) +
                                      (long)g.tiles_n * g.ktiles_total * (2L * bn * TK) * 2);
    const int n_stacked = g.tiles_n * bn;
    hipLaunchKernelGGL(b_exponent_kernel, dim3((n_stacked + 3) / 4), dim3(256), 0, s, bd, n_stacked, eb);
    if (bn == 256)
        hipLaunchKernelGGL((b_planes_kernel<256>), dim3(g.tiles_n * g.ktiles_total), dim3(256), 0, s, bd, g.ktiles_total, eb, planes,
                           wide ? 1 : 0);
    else
        hipLaunchKernelGGL((b_planes_kernel<128>), dim3(g.tiles_n * g.ktiles_total), dim3(128), 0, s, bd, g.ktiles_total, eb, planes, 0);
    g.a_rowmax = a_rowmax; g.bp = planes; g.eb = eb; g.alpha = alpha; g.beta = beta; g.c = c; g.ldc = ldc; g.bias = bias;
    g.x = gate_x; g.ldx = ld_x; g.g_out = gate_g; g.ldg = ld_g; g.z_out = gate_z; g.ldz = ld_z;
    const long n_tiles_mn = (long)g.tiles_m * g.tiles_n;
    auto lds_bytes = [](int bn_, int nt_, int tm_) {
        return 2 * (2 * tm_ * TK + 2 * bn_ * TK) * 2 + 3 * tm_ * TK * 4 + tm_ * 4 + 2 * bn_ * 4 +
               (nt_ + tm_ + 2 * bn_) * 4;       // + the landing area of the next tile's scalars
    };
    const int lds = lds_bytes(bn, tall256 ? 512 : (wide ? 256 : 2 * bn), tm_rows);
    const bool one = variant != 0;
#define LKG_TALL_GO(BN_, EPI_, ONE_) LKG_TALL_GO_T(BN_, EPI_, ONE_, 64, 128)
#define LKG_TALL_GO_W(BN_, EPI_, ONE_, WN_) LKG_TALL_GO_T(BN_, EPI_, ONE_, WN_, 128)
#define LKG_TALL_GO_T(BN_, EPI_, ONE_, WN_, TM_)                                                                                   \
    do {                                                                                                               \
        static bool raised = false;                                                                                    \
        static int resident = 0;      /* workgroups of this kernel the device holds at once (occupancy x CUs) */       \
        if (!raised && lds > 48 * 1024) {                                                                              \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tall_kernel<BN_, EPI_, ONE_, WN_, TM_>),                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {                  \
                lkg_set_error("lkg_gemm_tall_f32: cannot raise the dynamic LDS limit");                                \
                return LKG_ERR_HIP;                                                                                    \
            }                                                                                                          \
            raised = true;                                                                                             \
        }                                                                                                              \
        if (!resident) {                                                                                               \
            int dev_ = 0, per_cu_ = 0, cus_ = 0;                                                                       \
            if (hipGetDevice(&dev_) != hipSuccess ||                                                                   \
                hipDeviceGetAttribute(&cus_, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess ||             \
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_, gemm_tall_kernel<BN_, EPI_, ONE_, WN_, TM_>,              \
                                                             (TM_ / 64) * (BN_ / WN_) * 64, lds) != hipSuccess ||               \
                per_cu_ < 1 || cus_ < 1) {                                                                             \
                lkg_set_error("lkg_gemm_tall_f32: cannot size the persistent grid");                                   \
                return LKG_ERR_HIP;                                                                                    \
            }                                                                                                          \
            resident = std::max(8, per_cu_ * cus_ / 8 * 8);                                                            \
        }                                                                                                              \
        /* persistent workgroups: one stripe of tiles per resident workgroup, a multiple of 8 (one per XCD);           \
           LKG_TALL_ONE_TILE=1 (A/B switch): one workgroup per tile, as before round 3 */                              \
        /* several column tiles per row tile (the gate: 2): one workgroup per tile -- the workgroups of a row tile then \
           start together and the second one finds the A windows in the L2 (1.54 x the algorithmic traffic by the fabric \
           counters); persistent workgroups drift apart and every row tile's inputs cross the fabric twice (1.98 x) */   \
        const bool one_tile_ = g.tiles_n > 1;                                                                          \
        const dim3 grid((unsigned)std::min<long>((n_tiles_mn + 7) / 8 * 8, one_tile_ ? (1L << 30) : (long)resident));   \
        hipLaunchKernelGGL((gemm_tall_kernel<BN_, EPI_, ONE_, WN_, TM_>), grid, dim3((TM_ / 64) * (BN_ / WN_) * 64), lds, s, g);     \
    } while (0)
    if (ln) {
        g.slope = ln->slope; g.ln_eps = ln->ln_eps; g.norm_eps = ln->norm_eps; g.drop_p = ln->drop_p; g.seed = ln->seed;
        g.gamma = ln->gamma; g.beta_ln = ln->beta; g.yn = ln->yn; g.ldyn = ln->ldyn; g.mean = ln->mean; g.rstd = ln->rstd;
    }
    if (ws) {
#define LKG_TALL_GO_WS(EPI_)                                                                                           \
    do {                                                                                                               \
        static bool raised = false;                                                                                    \
        static int cus = 0;                                                                                            \
        if (!raised) {                                                                                                 \
            int dev_ = 0, per_cu_ = 0;                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tall_ws_kernel<EPI_>),                         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) != hipSuccess ||         \
                hipGetDevice(&dev_) != hipSuccess ||                                                                   \
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess ||              \
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_, gemm_tall_ws_kernel<EPI_>, 512, WS_LDS_BYTES)   \
                    != hipSuccess || per_cu_ < 1 || cus < 8) {                                                         \
                lkg_set_error("lkg_gemm_tall_f32: cannot set up the wave-specialised kernel (LDS limit / occupancy)"); \
                return LKG_ERR_HIP;                                                                                    \
            }                                                                                                          \
            raised = true;                                                                                             \
        }                                                                                                              \
        /* persistent: one workgroup per CU (a multiple of 8: one stripe of row tiles per workgroup and XCD) */         \
        const dim3 grid((unsigned)std::min<long>(((long)g.tiles_m + 7) / 8 * 8, (long)(cus / 8 * 8)));                 \
        hipLaunchKernelGGL((gemm_tall_ws_kernel<EPI_>), grid, dim3(512), WS_LDS_BYTES, s, g);                          \
    } while (0)
        if (epilogue == EPI_GATE) LKG_TALL_GO_WS(EPI_GATE);
        else if (epilogue == EPI_ACTLN) LKG_TALL_GO_WS(EPI_ACTLN);
        else LKG_TALL_GO_WS(EPI_PLAIN);
#undef LKG_TALL_GO_WS
    } else if (tall256) {
        if (epilogue == EPI_GATE) LKG_TALL_GO_T(256, EPI_GATE, true, 128, 256);
        else LKG_TALL_GO_T(256, EPI_PLAIN, true, 128, 256);
    } else if (wide) {
        if (epilogue == EPI_GATE) LKG_TALL_GO_W(256, EPI_GATE, true, 128);
        else LKG_TALL_GO_W(256, EPI_PLAIN, true, 128);
    } else if (epilogue == EPI_ACTLN) {
        LKG_TALL_GO(256, EPI_ACTLN, true);       // (8 waves of 64 x 64, two workgroups per CU)
    } else if (epilogue == EPI_GATE) {
        if (bn == 256 && !one) LKG_TALL_GO(256, EPI_GATE, false);
        else if (bn == 256) LKG_TALL_GO(256, EPI_GATE, true);
        else LKG_TALL_GO(128, EPI_GATE, true);
    } else if (bn == 256) {
        if (!one) LKG_TALL_GO(256, EPI_PLAIN, false);
        else LKG_TALL_GO(256, EPI_PLAIN, true);
    } else {
        if (!one) LKG_TALL_GO(128, EPI_PLAIN, false);
        else LKG_TALL_GO(128, EPI_PLAIN, true);
    }
#undef LKG_TALL_GO
#undef LKG_TALL_GO_W
#undef LKG_TALL_GO_T
    LKG_CHECK_LAUNCH("lkg_gemm_tall_f32");
    return LKG_OK;
}

extern "C" int lkg_gemm_tall_f32(int64_t m, int32_t n, int32_t n_panels, const float *const *a, const int64_t *lda,
                                 const int32_t *ka, const float *a_rowmax, int32_t n_groups, const float *const *b,
                                 const int64_t *ldb, int32_t trans_b, float alpha, float beta, float *c, int64_t ldc,
                                 const float *bias, int32_t epilogue, const float *gate_x, int64_t ld_x, float *gate_g,
                                 int64_t ld_g, float *gate_z, int64_t ld_z, void *workspace, int64_t workspace_bytes,
                                 void *stream) {
    LKG_REQUIRE((epilogue & 0xff) != EPI_ACTLN, "lkg_gemm_tall_f32: the act + LayerNorm epilogue has its own entry point "
                "(lkg_linear_act_layernorm_fwd_f32)");
    return tall_call(m, n, n_panels, a, lda, ka, a_rowmax, n_groups, b, ldb, trans_b, alpha, beta, c, ldc, bias, epilogue,
                     gate_x, ld_x, gate_g, ld_g, gate_z, ld_z, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int64_t lkg_linear_act_layernorm_workspace(int32_t n, int32_t n_panels, const int32_t *ka) {
    return lkg_gemm_tall_workspace(n, n_panels, ka, EPI_ACTLN);
}

extern "C" int lkg_linear_act_layernorm_fwd_f32(int64_t m, int32_t n, int32_t n_panels, const float *const *a,
                                                const int64_t *lda, const int32_t *ka, const float *a_rowmax,
                                                const float *const *w, const int64_t *ldw, const float *bias, float slope,
                                                const float *gamma, const float *beta, float eps, float *y, int64_t ldy,
                                                float *yn, int64_t ldyn, float norm_eps, float *save_mean,
                                                float *save_rstd, float drop_p, uint64_t seed, void *workspace,
                                                int64_t workspace_bytes, void *stream) {
    const LnExtra ln{slope, eps, norm_eps, drop_p, (unsigned long long)seed, gamma, beta, yn, ldyn, save_mean, save_rstd};
    return tall_call(m, n, n_panels, a, lda, ka, a_rowmax, 1, w, ldw, 1, 1.f, 0.f, y, ldy, bias, EPI_ACTLN, nullptr, 0,
                     nullptr, 0, nullptr, 0, workspace, workspace_bytes, stream, &ln);
}

// lkg_preload(): HIP loads a translation unit's code object on the first use of one of its kernels; asking for a kernel's
// attributes is such a use (no launch).
int lkg_internal_preload_gemm_tall() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&b_exponent_kernel)) == hipSuccess ? 0 : 1;
}
