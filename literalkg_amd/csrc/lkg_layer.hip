// A NARROW aggregation layer's dense backward in one launch (SURVEY.md 2.1, K5 + the Linear around it: model.py:108-111 --
// y = Dropout(LayerNorm(LeakyReLU(x W^T + b))), yn = y / |y| -- for the reference's default conv_dim of 32, eight layers of it:
// argument_pretraining.py:54-58).
//
// Unfused, the backward of such a layer over all N rows is four launches: the row-wise LayerNorm / LeakyReLU / dropout backward
// (reads z, y, g_y, g_yn, writes g_z), the data gradient g_x = g_z W (reads g_z, writes g_x), the weight gradient
// g_W = g_z^T x (reads g_z and x again) and the bias gradient's column sum (reads g_z a third time): 461 us at 1 M rows
// (profiles/r04_default_step_kernels.txt).  Here a wave takes 32 rows at a time: the row-wise arithmetic -- the SAME
// arithmetic, lane for lane, as act_ln_bwd_narrow_kernel<8> in lkg_rowwise.hip -- leaves g_z of the 32 rows in the wave's
// own LDS tile next to the rows of x, and both products run on the matrix cores in f32 (v_mfma_f32_32x32x2_f32: exact f32
// products, f32 sums): g_x's 32 x 32 tile from 16 MFMAs against W held in 16 registers, g_W accumulated over ALL of the
// wave's tiles in 16 more (16 MFMAs per tile), folded across waves in LDS; every workgroup leaves its 1120 partial sums
// (g_W, g_b, g_gamma, g_beta) in a workspace row and a second, tiny launch adds the rows up in a fixed order (1024 workgroups'
// atomics on the same 4.4 KB took longer than the rows' whole pass: contended f32 atomics run at ~0.1 TB/s).
// g_z never reaches memory.  HBM-bound: 4 n d (z, y, g_y, g_yn, x read; g_x written) = 768 MB at 1 M rows.
#include "lkg_common.h"

#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int D = 32;            // columns in and out
constexpr int LPR = 8;           // lanes per row in the row-wise phase (16 bytes each)
constexpr int RPP = 64 / LPR;    // rows per pass
constexpr int TILE = 32;         // rows per wave tile
constexpr int PITCH = 36;        // floats per LDS row: 16-byte aligned, rows 4 banks apart
constexpr int WAVES = 4;
constexpr int N_SUMS = D * D + 3 * D;   // what a workgroup leaves behind: g_W | g_gamma | g_beta | g_bias
constexpr int MAX_BLOCKS = 256 * 4;     // 36 KB of LDS per workgroup: four per CU

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void narrow_layer_bwd_kernel(
    long n, const float *__restrict__ x, long ldx, const float *__restrict__ w, long ldw, const float *__restrict__ z, long ldz,
    float slope, const float *__restrict__ gamma, const float *__restrict__ y, long ldy, const float *__restrict__ save_mean,
    const float *__restrict__ save_rstd, const float *__restrict__ g_y, long ldgy, const float *__restrict__ g_yn, long ldgyn,
    float norm_eps, float drop_p, unsigned long long seed, const unsigned char *__restrict__ gyn_rows, float *__restrict__ g_x,
    long ldgx, float *__restrict__ partials) {
    // g_z of the tile, DE-INTERLEAVED per row: position (o & 1) * 16 + (o >> 1) holds column o -- the MFMA's A operand of the data
    // gradient is "columns 2 t + k of row m" for lane (m, k): 16 consecutive floats of the row here, four 16-byte reads
    __shared__ __attribute__((aligned(16))) float gz_s[WAVES][TILE][PITCH], x_s[WAVES][TILE][PITCH];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sl = lane % LPR, sub = lane / LPR;
    const int e = sl * 4;
    const int mm = lane & 31, kk = lane >> 5;                 // this lane in an MFMA operand: row / column mm, k index kk
    const long wave = (long)blockIdx.x * WAVES + wv, nwaves = (long)gridDim.x * WAVES;
    const long n_tiles = (n + TILE - 1) / TILE;
    float gam[4];
    {
        const float4 t = *reinterpret_cast<const float4 *>(gamma + e);
        gam[0] = t.x; gam[1] = t.y; gam[2] = t.z; gam[3] = t.w;
    }
    // B operand of the data gradient g_x[m][i] = sum_o g_z[m][o] W[o][i]: B[k][i] = W[2 t + k][i], loop-invariant
    float wb[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) wb[t] = w[(long)(2 * t + kk) * ldw + mm];
    f32x16 gw_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) gw_acc[r] = 0.f;
    float acc_g[4] = {0.f, 0.f, 0.f, 0.f}, acc_b[4] = {0.f, 0.f, 0.f, 0.f}, acc_bias[4] = {0.f, 0.f, 0.f, 0.f};
    float(*gz_t)[PITCH] = gz_s[wv];
    float(*x_t)[PITCH] = x_s[wv];

    for (long tile = wave; tile < n_tiles; tile += nwaves) {
        const long row0 = tile * TILE;
        // ---- the row-wise phase: 4 passes of 8 rows, 8 lanes per row (act_ln_bwd_narrow_kernel<8>'s arithmetic).  Not unrolled:
        // four passes' loads at once are 80 registers -- 212 VGPRs, two waves per SIMD; one pass at a time fits 128, four waves.
#pragma unroll 1
        for (int p = 0; p < TILE / RPP; ++p) {
            const int tr = p * RPP + sub;                          // row inside the tile
            const long row_raw = row0 + tr;
            const bool valid = row_raw < n;
            const long row = valid ? row_raw : n - 1;
            const bool has_gyn = g_yn && (!gyn_rows || gyn_rows[row]);
            const bool any = valid && (g_y || has_gyn);            // does a gradient reach this row at all?
            auto ld4 = [&](const float *base, long ld, bool on, float (&v)[4]) {
                v[0] = v[1] = v[2] = v[3] = 0.f;
                if (on) {                                          // (a masked global load; rows it skips are not fetched)
                    const float4 t = *reinterpret_cast<const float4 *>(base + row * ld + e);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                }
            };
            float zz[4], G[4], xv[4];
            ld4(z, ldz, any, zz);
            ld4(g_y, ldgy, any && g_y != nullptr, G);
            ld4(x, ldx, any, xv);                                  // (rows no gradient reaches add nothing to g_W: zeros)
            {
                float yy[4], gn[4];
                ld4(y, ldy, any && has_gyn, yy);
                ld4(g_yn, ldgyn, any && has_gyn, gn);
                float n2 = 0.f, dt = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    n2 = fmaf(yy[k], yy[k], n2);
                    dt = fmaf(yy[k], gn[k], dt);
                }
                const float nrm = sqrtf(group_sum<LPR>(n2));
                dt = group_sum<LPR>(dt);
                if (nrm > norm_eps) {                 // (rows without g_yn: yy = gn = 0, nrm = 0: the else branch adds zeros)
                    const float inv = 1.f / nrm;
                    const float proj = dt * inv * inv * inv;
#pragma unroll
                    for (int k = 0; k < 4; ++k) G[k] += gn[k] * inv - yy[k] * proj;
                } else {
                    const float inv = 1.f / norm_eps;
#pragma unroll
                    for (int k = 0; k < 4; ++k) G[k] += gn[k] * inv;
                }
            }
            if (drop_p > 0.f) {
                const float inv_keep = 1.f / (1.f - drop_p);
                const unsigned rkey = drop_row_key(seed, (unsigned long long)row);
#pragma unroll
                for (int k = 0; k < 4; ++k) G[k] *= drop_scale(rkey, (unsigned)(e + k), drop_p, inv_keep);
            }
            const float mean = save_mean[row], rstd = save_rstd[row];
            float s1 = 0.f, s2 = 0.f, xh[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = zz[k];
                const float a = t > 0.f ? t : t * slope;
                xh[k] = any ? (a - mean) * rstd : 0.f;
                const float dx = G[k] * gam[k];
                s1 += dx;
                s2 = fmaf(dx, xh[k], s2);
                acc_g[k] = fmaf(G[k], xh[k], acc_g[k]);
                acc_b[k] += G[k];
            }
            s1 = group_sum<LPR>(s1) / (float)D;
            s2 = group_sum<LPR>(s2) / (float)D;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dx = G[k] * gam[k];
                const float da = rstd * (dx - s1 - xh[k] * s2);
                zz[k] = any ? da * (zz[k] > 0.f ? 1.f : slope) : 0.f;
                acc_bias[k] += zz[k];
            }
            // columns e .. e + 3 = 4 sl .. 4 sl + 3: even ones to positions 2 sl, 2 sl + 1, odd ones to 16 + 2 sl, 16 + 2 sl + 1
            *reinterpret_cast<float2 *>(&gz_t[tr][2 * sl]) = make_float2(zz[0], zz[2]);
            *reinterpret_cast<float2 *>(&gz_t[tr][16 + 2 * sl]) = make_float2(zz[1], zz[3]);
            *reinterpret_cast<float4 *>(&x_t[tr][e]) = make_float4(xv[0], xv[1], xv[2], xv[3]);
        }
        // ---- g_x tile = g_z tile . W on the matrix cores (f32 operands): A[m][k] = g_z[m][2 t + k] = gz_t[m][16 k + t]
        {
            float ga[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&gz_t[mm][16 * kk + 4 * q]);
                ga[4 * q] = t4.x; ga[4 * q + 1] = t4.y; ga[4 * q + 2] = t4.z; ga[4 * q + 3] = t4.w;
            }
            f32x16 gx;
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) gx = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[t], wb[t], gx, 0, 0, 0);
            // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): two 128-byte runs per store
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row < n) g_x[row * ldgx + mm] = gx[r];
            }
        }
        // ---- g_W += g_z tile^T . x tile: A[o][k] = g_z[2 t + k][o], B[k][i] = x[2 t + k][i]
        {
            const int opos = (mm & 1) * 16 + (mm >> 1);
#pragma unroll
            for (int t = 0; t < 16; ++t)
                gw_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gz_t[2 * t + kk][opos], x_t[2 * t + kk][mm], gw_acc, 0, 0, 0);
        }
    }

    // ---- the sums every wave carries: g_W (C/D layout), g_gamma / g_beta / g_bias (columns e .. e+3 in lane sl of every
    // sub-group): fold the sub-groups by shuffles, the waves in LDS (the tiles' own memory, free now), then this workgroup's row
    // of the workspace: [g_W 1024 | g_gamma 32 | g_beta 32 | g_bias 32]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc_g[k] = lane_xor_add<32>(lane_xor_add<16>(lane_xor_add<8>(acc_g[k])));
        acc_b[k] = lane_xor_add<32>(lane_xor_add<16>(lane_xor_add<8>(acc_b[k])));
        acc_bias[k] = lane_xor_add<32>(lane_xor_add<16>(lane_xor_add<8>(acc_bias[k])));
    }
    __syncthreads();                                          // every wave is done with its tiles
    float *red_w = &gz_s[0][0][0];                            // [3][16][64] floats = 3072 <= WAVES * TILE * PITCH = 4608
    float *red_v = &x_s[0][0][0];                             // [3][3][32]
    if (wv > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red_w[((wv - 1) * 16 + r) * 64 + lane] = gw_acc[r];
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                red_v[((wv - 1) * 3 + 0) * D + e + k] = acc_g[k];
                red_v[((wv - 1) * 3 + 1) * D + e + k] = acc_b[k];
                red_v[((wv - 1) * 3 + 2) * D + e + k] = acc_bias[k];
            }
        }
    }
    __syncthreads();
    if (wv == 0) {
        float *mine = partials + (long)blockIdx.x * N_SUMS;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = gw_acc[r] + red_w[(0 * 16 + r) * 64 + lane] + red_w[(1 * 16 + r) * 64 + lane] + red_w[(2 * 16 + r) * 64 + lane];
            const int o = (r & 3) + 8 * (r >> 2) + 4 * kk;
            mine[o * D + mm] = v;
        }
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mine[D * D + e + k] = acc_g[k] + red_v[(0 * 3 + 0) * D + e + k] + red_v[(1 * 3 + 0) * D + e + k] + red_v[(2 * 3 + 0) * D + e + k];
                mine[D * D + D + e + k] = acc_b[k] + red_v[(0 * 3 + 1) * D + e + k] + red_v[(1 * 3 + 1) * D + e + k] + red_v[(2 * 3 + 1) * D + e + k];
                mine[D * D + 2 * D + e + k] = acc_bias[k] + red_v[(0 * 3 + 2) * D + e + k] + red_v[(1 * 3 + 2) * D + e + k] + red_v[(2 * 3 + 2) * D + e + k];
            }
        }
    }
}

// out[c] = sum over the rows of partials[rows][N_SUMS], in a fixed order: 32 columns per workgroup, 32 slices of the rows per column
__global__ __launch_bounds__(1024) void narrow_layer_sums_kernel(const float *__restrict__ partials, int rows, float *__restrict__ g_w,
                                                                 float *__restrict__ g_gamma, float *__restrict__ g_beta,
                                                                 float *__restrict__ g_bias) {
    __shared__ float red[32][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), slice = threadIdx.x >> 5;
    float v = 0.f;
    for (int r = slice; r < rows; r += 32) v += partials[(long)r * N_SUMS + c];
    red[slice][threadIdx.x & 31] = v;
    __syncthreads();
    if (slice == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) t += red[q][threadIdx.x & 31];
        if (c < D * D) g_w[c] = t;
        else if (c < D * D + D) g_gamma[c - D * D] = t;
        else if (c < D * D + 2 * D) g_beta[c - D * D - D] = t;
        else if (g_bias) g_bias[c - D * D - 2 * D] = t;
    }
}

}  // namespace

// 1 when lkg_narrow_layer_bwd_f32 takes this layer: 32 columns in and out, many rows, 16-byte aligned rows everywhere
extern "C" int lkg_narrow_layer_bwd_ok(int64_t n, int32_t d_in, int32_t d_out, const float *x, int64_t ldx, const float *z, int64_t ldz,
                                       const float *y, int64_t ldy, const float *g_y, int64_t ldgy, const float *g_yn, int64_t ldgyn) {
    auto rows_ok = [](const float *p, int64_t ld) { return !p || (ld % 4 == 0 && lkg_aligned16(p)); };
    return n >= 4096 && d_in == D && d_out == D && x && z && rows_ok(x, ldx) && rows_ok(z, ldz) && rows_ok(y, ldy) && rows_ok(g_y, ldgy) &&
           rows_ok(g_yn, ldgyn) && (!g_yn || y);
}

// floats of workspace lkg_narrow_layer_bwd_f32 needs for n rows (every workgroup's partial sums)
extern "C" int64_t lkg_narrow_layer_bwd_workspace(int64_t n) {
    const int64_t tiles = (n + TILE - 1) / TILE;
    return std::max<int64_t>(1, std::min<int64_t>((tiles + WAVES - 1) / WAVES, MAX_BLOCKS)) * N_SUMS;
}

// The dense backward of Dropout(LayerNorm(LeakyReLU(x W^T + b))) with its normalised copy, 32 columns in and out
// (model.py:108-111, 161, 305; the unfused pieces: lkg_act_layernorm_bwd_f32, lkg_gemm_skinny_f32, lkg_gemm_smallm_f32,
// lkg_colsum_f32).  g_w [32 x 32, contiguous], g_bias (may be null), g_gamma, g_beta are OVERWRITTEN with the sums (added up in
// a fixed order: the same bits from run to run).
extern "C" int lkg_narrow_layer_bwd_f32(int64_t n, int32_t d_in, int32_t d_out, const float *x, int64_t ldx, const float *w, int64_t ldw,
                                        const float *z, int64_t ldz, float slope, const float *gamma, const float *y, int64_t ldy,
                                        const float *save_mean, const float *save_rstd, const float *g_y, int64_t ldgy,
                                        const float *g_yn, int64_t ldgyn, float norm_eps, float drop_p, uint64_t seed,
                                        const uint8_t *g_yn_rows, float *g_x, int64_t ldgx, float *g_w, float *g_bias,
                                        float *g_gamma, float *g_beta, float *workspace, int64_t workspace_floats, void *stream) {
    LKG_REQUIRE(n > 0 && x && w && z && gamma && save_mean && save_rstd && g_x && g_w && g_gamma && g_beta && (g_y || g_yn) && workspace,
                "lkg_narrow_layer_bwd_f32: null pointer");
    LKG_REQUIRE(ldx >= d_in && ldw >= d_in && ldz >= d_out && ldgx >= d_in && (!y || ldy >= d_out) && (!g_y || ldgy >= d_out) &&
                (!g_yn || ldgyn >= d_out), "lkg_narrow_layer_bwd_f32: a row stride is shorter than its row");
    LKG_REQUIRE(lkg_narrow_layer_bwd_ok(n, d_in, d_out, x, ldx, z, ldz, y, ldy, g_y, ldgy, g_yn, ldgyn) && ldgx % 4 == 0 && lkg_aligned16(g_x) &&
                lkg_aligned16(gamma), "lkg_narrow_layer_bwd_f32: needs 32 columns in and out, n >= 4096, 16-byte aligned rows and y "
                "wherever g_yn is given (lkg_narrow_layer_bwd_ok)");
    LKG_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "lkg_narrow_layer_bwd_f32: drop_p outside [0, 1)");
    LKG_REQUIRE(workspace_floats >= lkg_narrow_layer_bwd_workspace(n), "lkg_narrow_layer_bwd_f32: workspace of %lld floats, %lld needed "
                "(lkg_narrow_layer_bwd_workspace)", (long long)workspace_floats, (long long)lkg_narrow_layer_bwd_workspace(n));
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(lkg_narrow_layer_bwd_workspace(n) / N_SUMS);
    hipLaunchKernelGGL(narrow_layer_bwd_kernel, dim3(blocks), dim3(256), 0, s, (long)n, x, (long)ldx, w, (long)ldw, z, (long)ldz, slope,
                       gamma, y, (long)ldy, save_mean, save_rstd, g_y, (long)ldgy, g_yn, (long)ldgyn, norm_eps, drop_p,
                       (unsigned long long)seed, g_yn_rows, g_x, (long)ldgx, workspace);
    LKG_CHECK_LAUNCH("lkg_narrow_layer_bwd_f32");
    hipLaunchKernelGGL(narrow_layer_sums_kernel, dim3(N_SUMS / 32), dim3(1024), 0, s, workspace, (int)blocks, g_w, g_gamma, g_beta, g_bias);
    LKG_CHECK_LAUNCH("lkg_narrow_layer_bwd_f32 (sums)");
    return LKG_OK;
}

int lkg_internal_preload_layer() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&narrow_layer_bwd_kernel)) == hipSuccess ? 0 : 1;
}
