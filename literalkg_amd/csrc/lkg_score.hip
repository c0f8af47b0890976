// K8: triple scoring + loss (model.py:364-428 TransR form on projected rows,
// model_bce.py:329-368 TransE form on table rows) and its backward.
//
// HBM/L2-bound row gathers: one wave per triple, each lane strides over the row in 16-byte chunks,
// six running sums are finished with one xor-shuffle tree each.
#include <algorithm>

#include "lkg_common.h"

namespace {

__device__ __forceinline__ void six_reduce(float &a, float &b, float &c, float &d, float &e, float &f) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    d = wave_sum(d);
    e = wave_sum(e);
    f = wave_sum(f);
}

template <bool VEC>
__device__ __forceinline__ void score_row(const float *eh, const float *ep, const float *en, const float *er, int dim,
                                          int lane, float &pos, float &neg, float &reg) {
    float sp = 0.f, sn = 0.f, qh = 0.f, qr = 0.f, qp = 0.f, qn = 0.f;
    if constexpr (VEC) {
        const float4 *h4 = reinterpret_cast<const float4 *>(eh), *p4 = reinterpret_cast<const float4 *>(ep),
                     *n4 = reinterpret_cast<const float4 *>(en), *r4 = reinterpret_cast<const float4 *>(er);
        for (int c = lane; c < dim / 4; c += 64) {
            const float4 a = h4[c], b = p4[c], g = n4[c], r = r4[c];
#define LKG_ACC(F)                                    \
    {                                                 \
        const float u = a.F + r.F - b.F;              \
        const float w = a.F + r.F - g.F;              \
        sp = fmaf(u, u, sp);                          \
        sn = fmaf(w, w, sn);                          \
        qh = fmaf(a.F, a.F, qh);                      \
        qr = fmaf(r.F, r.F, qr);                      \
        qp = fmaf(b.F, b.F, qp);                      \
        qn = fmaf(g.F, g.F, qn);                      \
    }
            LKG_ACC(x) LKG_ACC(y) LKG_ACC(z) LKG_ACC(w)
#undef LKG_ACC
        }
    } else {
        for (int c = lane; c < dim; c += 64) {
            const float a = eh[c], b = ep[c], g = en[c], r = er[c];
            const float u = a + r - b, w = a + r - g;
            sp = fmaf(u, u, sp);
            sn = fmaf(w, w, sn);
            qh = fmaf(a, a, qh);
            qr = fmaf(r, r, qr);
            qp = fmaf(b, b, qp);
            qn = fmaf(g, g, qn);
        }
    }
    six_reduce(sp, sn, qh, qr, qp, qn);
    pos = sp;
    neg = sn;
    reg = 0.5f * (qh + qr + qp + qn);
}

template <bool VEC, bool DENSE>
__global__ __launch_bounds__(256) void score_fwd_kernel(long batch, int dim, const float *__restrict__ a,
                                                         const float *__restrict__ b, const float *__restrict__ c,
                                                         long ld, const float *__restrict__ relemb, long ld_rel,
                                                         const long *__restrict__ h, const long *__restrict__ r,
                                                         const long *__restrict__ pt, const long *__restrict__ nt,
                                                         float *__restrict__ pos, float *__restrict__ neg,
                                                         float *__restrict__ reg, float *__restrict__ rank, int rpg) {
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= batch) return;
    const float *eh, *ep, *en;
    long tr = t;
    if constexpr (DENSE) {   // rpg consecutive rows share one projected head / positive tail / relation (a12 layout)
        tr = t / rpg;
        eh = a + tr * ld;
        ep = b + tr * ld;
        en = c + t * ld;
    } else {
        eh = a + h[t] * ld;
        ep = a + pt[t] * ld;
        en = a + nt[t] * ld;
    }
    const float *er = relemb + r[tr] * ld_rel;
    float ps, ng, rg;
    score_row<VEC>(eh, ep, en, er, dim, lane, ps, ng, rg);
    if (lane == 0) {
        pos[t] = ps;
        neg[t] = ng;
        reg[t] = rg;
        rank[t] = neg_logsigmoid(ng - ps);
    }
}

// loss = mean(rank) + lambda * mean(reg).  One block, fixed summation order (deterministic).
__global__ __launch_bounds__(1024) void loss_reduce_kernel(long batch, const float *__restrict__ rank,
                                                            const float *__restrict__ reg, float lambda,
                                                            float *__restrict__ out) {
    __shared__ float sa[16], sb[16];
    float a = 0.f, b = 0.f;
    for (long i = threadIdx.x; i < batch; i += blockDim.x) {
        a += rank[i];
        b += reg[i];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sa[w] = a;
        sb[w] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
            ta += sa[i];
            tb += sb[i];
        }
        out[0] = ta / (float)batch + lambda * (tb / (float)batch);
    }
}

// d loss / d pos_b = g * sigmoid(pos_b - neg_b) / B ;  d loss / d neg_b = -(same)
// u = h + r - p, w = h + r - n
//   g_h += 2u*dp + 2w*dn + (lambda/B) h     g_r  += 2u*dp + 2w*dn + (lambda/B) r
//   g_p += -2u*dp + (lambda/B) p            g_n  += -2w*dn + (lambda/B) n
template <bool DENSE>
__global__ __launch_bounds__(256) void score_bwd_kernel(long batch, int dim, const float *__restrict__ a,
                                                         const float *__restrict__ b, const float *__restrict__ c,
                                                         long ld, const float *__restrict__ relemb, long ld_rel,
                                                         const long *__restrict__ h, const long *__restrict__ r,
                                                         const long *__restrict__ pt, const long *__restrict__ nt,
                                                         const float *__restrict__ pos, const float *__restrict__ neg,
                                                         float lambda, const float *__restrict__ g_loss,
                                                         float *__restrict__ ga, float *__restrict__ gb,
                                                         float *__restrict__ gc, long ldg, float *__restrict__ g_rel,
                                                         long ld_grel) {
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= batch) return;
    const float g = g_loss[0];
    const float inv_b = 1.f / (float)batch;
    const float dp = g * sigmoidf_(pos[t] - neg[t]) * inv_b;
    const float dn = -dp;
    const float lr = g * lambda * inv_b;
    const float *eh, *ep, *en;
    float *gh, *gp, *gn;
    if constexpr (DENSE) {
        eh = a + t * ld;
        ep = b + t * ld;
        en = c + t * ld;
        gh = ga + t * ldg;
        gp = gb + t * ldg;
        gn = gc + t * ldg;
    } else {
        eh = a + h[t] * ld;
        ep = a + pt[t] * ld;
        en = a + nt[t] * ld;
        gh = ga + h[t] * ldg;
        gp = ga + pt[t] * ldg;
        gn = ga + nt[t] * ldg;
    }
    const float *er = relemb + r[t] * ld_rel;
    float *gr = g_rel + r[t] * ld_grel;
    for (int k = lane; k < dim; k += 64) {
        const float vh = eh[k], vp = ep[k], vn = en[k], vr = er[k];
        const float u = vh + vr - vp, w = vh + vr - vn;
        const float common = 2.f * (u * dp + w * dn);
        if constexpr (DENSE) {   // each (t, k) is written by exactly one lane: plain stores
            gh[k] = common + lr * vh;
            gp[k] = -2.f * u * dp + lr * vp;
            gn[k] = -2.f * w * dn + lr * vn;
        } else {
            atomicAdd(gh + k, common + lr * vh);
            atomicAdd(gp + k, -2.f * u * dp + lr * vp);
            atomicAdd(gn + k, -2.f * w * dn + lr * vn);
        }
        atomicAdd(gr + k, common + lr * vr);
    }
}

// Backward on projected rows when RPG consecutive rows share (h, r, t+) (the a12 layout of generate_kg_batch,
// dataloader.py:318-330): ph / pp / r / g_ph / g_pp hold ONE row per group.  One workgroup per group; its four waves
// take interleaved rows, a lane owns JT columns of a 64*JT-column tile and keeps the group's sums in registers
//   S = sum_k 2 (u_k dp_k + w_k dn_k)      P = sum_k -2 u_k dp_k
// (g_ph = S + K lr ph, g_pp = P + K lr pp, g_rel[r] += S + K lr r: one atomic per group and column instead of one
// per row); g_pn is written row by row.  The waves meet in LDS in a fixed order: deterministic apart from g_rel.
template <int JT>
__global__ __launch_bounds__(256) void score_bwd_grouped_kernel(long n_groups, int rpg, int dim,
                                                                 const float *__restrict__ ph,
                                                                 const float *__restrict__ pp,
                                                                 const float *__restrict__ pn, long ld,
                                                                 const float *__restrict__ relemb, long ld_rel,
                                                                 const long *__restrict__ r,
                                                                 const float *__restrict__ pos,
                                                                 const float *__restrict__ neg, float lambda,
                                                                 const float *__restrict__ g_loss,
                                                                 float *__restrict__ g_ph, float *__restrict__ g_pp,
                                                                 float *__restrict__ g_pn, long ldg,
                                                                 float *__restrict__ g_rel, long ld_grel) {
    __shared__ float red[2][4][JT][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long grp = blockIdx.x;
    const long batch = n_groups * rpg;
    const float g = g_loss[0];
    const float inv_b = 1.f / (float)batch;
    const float lr = g * lambda * inv_b;
    const float *eh = ph + grp * ld, *ep = pp + grp * ld, *er = relemb + r[grp] * ld_rel;
    for (int c0 = 0; c0 < dim; c0 += 64 * JT) {
        float vh[JT], vp[JT], vr[JT], S[JT], P[JT];
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int c = c0 + lane + 64 * j;
            const bool ok = c < dim;
            vh[j] = ok ? eh[c] : 0.f;
            vp[j] = ok ? ep[c] : 0.f;
            vr[j] = ok ? er[c] : 0.f;
            S[j] = P[j] = 0.f;
        }
        for (int k = w; k < rpg; k += 4) {
            const long t = grp * rpg + k;
            const float dp = g * sigmoidf_(pos[t] - neg[t]) * inv_b, dn = -dp;
            const float *en = pn + t * ld;
            float *gn = g_pn + t * ldg;
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int c = c0 + lane + 64 * j;
                if (c < dim) {
                    const float vn = en[c];
                    const float u = vh[j] + vr[j] - vp[j], q = vh[j] + vr[j] - vn;
                    S[j] += 2.f * (u * dp + q * dn);
                    P[j] += -2.f * u * dp;
                    gn[c] = -2.f * q * dn + lr * vn;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            red[0][w][j][lane] = S[j];
            red[1][w][j][lane] = P[j];
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int c = c0 + lane + 64 * j;
                if (c < dim) {
                    const float s = (red[0][0][j][lane] + red[0][1][j][lane]) + (red[0][2][j][lane] + red[0][3][j][lane]);
                    const float p = (red[1][0][j][lane] + red[1][1][j][lane]) + (red[1][2][j][lane] + red[1][3][j][lane]);
                    const float klr = lr * (float)rpg;
                    g_ph[grp * ldg + c] = s + klr * vh[j];
                    g_pp[grp * ldg + c] = p + klr * vp[j];
                    atomicAdd(g_rel + r[grp] * ld_grel + c, s + klr * vr[j]);
                }
            }
        }
        __syncthreads();
    }
}

// f1 fine-tuning head: dot-product BPR (model.py:316-348)
__global__ __launch_bounds__(256) void dot_fwd_kernel(long batch, int dim, const float *__restrict__ emb, long ld,
                                                       const long *__restrict__ h, const long *__restrict__ pt,
                                                       const long *__restrict__ nt, float *__restrict__ pos,
                                                       float *__restrict__ neg, float *__restrict__ reg,
                                                       float *__restrict__ rank) {
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= batch) return;
    const float *eh = emb + h[t] * ld, *ep = emb + pt[t] * ld, *en = emb + nt[t] * ld;
    float sp = 0.f, sn = 0.f, qh = 0.f, qp = 0.f, qn = 0.f, zero = 0.f;
    for (int k = lane; k < dim; k += 64) {
        const float a = eh[k], b = ep[k], c = en[k];
        sp = fmaf(a, b, sp);
        sn = fmaf(a, c, sn);
        qh = fmaf(a, a, qh);
        qp = fmaf(b, b, qp);
        qn = fmaf(c, c, qn);
    }
    six_reduce(sp, sn, qh, qp, qn, zero);
    if (lane == 0) {
        pos[t] = sp;
        neg[t] = sn;
        reg[t] = 0.5f * (qh + qp + qn);
        rank[t] = neg_logsigmoid(sp - sn);
    }
}

// d loss/d pos_b = -g * sigmoid(neg_b - pos_b) / B, d loss/d neg_b = +(same)
__global__ __launch_bounds__(256) void dot_bwd_kernel(long batch, int dim, const float *__restrict__ emb, long ld,
                                                       const long *__restrict__ h, const long *__restrict__ pt,
                                                       const long *__restrict__ nt, const float *__restrict__ pos,
                                                       const float *__restrict__ neg, float lambda,
                                                       const float *__restrict__ g_loss, float *__restrict__ ge,
                                                       long ldg) {
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= batch) return;
    const float g = g_loss[0], inv_b = 1.f / (float)batch;
    const float dn = g * sigmoidf_(neg[t] - pos[t]) * inv_b, dp = -dn, lr = g * lambda * inv_b;
    const float *eh = emb + h[t] * ld, *ep = emb + pt[t] * ld, *en = emb + nt[t] * ld;
    float *gh = ge + h[t] * ldg, *gp = ge + pt[t] * ldg, *gn = ge + nt[t] * ldg;
    for (int k = lane; k < dim; k += 64) {
        const float a = eh[k], b = ep[k], c = en[k];
        atomicAdd(gh + k, dp * b + dn * c + lr * a);
        atomicAdd(gp + k, dp * a + lr * b);
        atomicAdd(gn + k, dn * a + lr * c);
    }
}

inline bool vec_ok(int dim, int64_t ld, int64_t ld_rel, const void *p0, const void *p1) {
    return dim % 4 == 0 && ld % 4 == 0 && ld_rel % 4 == 0 && lkg_aligned16(p0) && lkg_aligned16(p1);
}

}  // namespace

extern "C" int lkg_transe_score_fwd_f32(int64_t batch, int32_t dim, const float *emb, int64_t ld_emb,
                                        const float *relemb, int64_t ld_rel, const int64_t *h, const int64_t *r,
                                        const int64_t *pos_t, const int64_t *neg_t, float *pos, float *neg,
                                        float *reg, float *rank, void *stream) {
    LKG_REQUIRE(batch >= 0 && dim > 0 && ld_emb >= dim && ld_rel >= dim, "lkg_transe_score_fwd_f32: bad sizes");
    if (batch == 0) return LKG_OK;
    LKG_REQUIRE(emb && relemb && h && r && pos_t && neg_t && pos && neg && reg && rank,
                "lkg_transe_score_fwd_f32: null pointer");
    const dim3 grid((unsigned)((batch + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (vec_ok(dim, ld_emb, ld_rel, emb, relemb))
        hipLaunchKernelGGL((score_fwd_kernel<true, false>), grid, block, 0, s, (long)batch, dim, emb, nullptr, nullptr,
                           (long)ld_emb, relemb, (long)ld_rel, (const long *)h, (const long *)r, (const long *)pos_t,
                           (const long *)neg_t, pos, neg, reg, rank, 1);
    else
        hipLaunchKernelGGL((score_fwd_kernel<false, false>), grid, block, 0, s, (long)batch, dim, emb, nullptr,
                           nullptr, (long)ld_emb, relemb, (long)ld_rel, (const long *)h, (const long *)r,
                           (const long *)pos_t, (const long *)neg_t, pos, neg, reg, rank, 1);
    LKG_CHECK_LAUNCH("lkg_transe_score_fwd_f32");
    return LKG_OK;
}

extern "C" int lkg_dense_score_fwd_f32(int64_t batch, int32_t rows_per_group, int32_t dim, const float *ph,
                                       const float *pp, const float *pn, int64_t ld, const float *relemb,
                                       int64_t ld_rel, const int64_t *r, float *pos, float *neg, float *reg,
                                       float *rank, void *stream) {
    LKG_REQUIRE(batch >= 0 && dim > 0 && ld >= dim && ld_rel >= dim, "lkg_dense_score_fwd_f32: bad sizes");
    LKG_REQUIRE(rows_per_group >= 1 && batch % rows_per_group == 0,
                "lkg_dense_score_fwd_f32: batch %lld is not a whole number of groups of %d rows", (long long)batch,
                rows_per_group);
    if (batch == 0) return LKG_OK;
    LKG_REQUIRE(ph && pp && pn && relemb && r && pos && neg && reg && rank, "lkg_dense_score_fwd_f32: null pointer");
    const dim3 grid((unsigned)((batch + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool v = vec_ok(dim, ld, ld_rel, ph, relemb) && lkg_aligned16(pp) && lkg_aligned16(pn);
    if (v)
        hipLaunchKernelGGL((score_fwd_kernel<true, true>), grid, block, 0, s, (long)batch, dim, ph, pp, pn, (long)ld,
                           relemb, (long)ld_rel, nullptr, (const long *)r, nullptr, nullptr, pos, neg, reg, rank,
                           rows_per_group);
    else
        hipLaunchKernelGGL((score_fwd_kernel<false, true>), grid, block, 0, s, (long)batch, dim, ph, pp, pn, (long)ld,
                           relemb, (long)ld_rel, nullptr, (const long *)r, nullptr, nullptr, pos, neg, reg, rank,
                           rows_per_group);
    LKG_CHECK_LAUNCH("lkg_dense_score_fwd_f32");
    return LKG_OK;
}

extern "C" int lkg_loss_reduce_f32(int64_t batch, const float *rank, const float *reg, float lambda, float *loss_out,
                                   void *stream) {
    LKG_REQUIRE(batch > 0, "lkg_loss_reduce_f32: empty batch (mean over zero triples is undefined)");
    LKG_REQUIRE(rank && reg && loss_out, "lkg_loss_reduce_f32: null pointer");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (long)batch, rank, reg, lambda,
                       loss_out);
    LKG_CHECK_LAUNCH("lkg_loss_reduce_f32");
    return LKG_OK;
}

extern "C" int lkg_transe_score_bwd_f32(int64_t batch, int32_t dim, const float *emb, int64_t ld_emb,
                                        const float *relemb, int64_t ld_rel, const int64_t *h, const int64_t *r,
                                        const int64_t *pos_t, const int64_t *neg_t, const float *pos, const float *neg,
                                        float lambda, const float *g_loss, float *g_emb, int64_t ld_gemb, float *g_rel,
                                        int64_t ld_grel, void *stream) {
    LKG_REQUIRE(batch >= 0 && dim > 0 && ld_emb >= dim && ld_rel >= dim && ld_gemb >= dim && ld_grel >= dim,
                "lkg_transe_score_bwd_f32: bad sizes");
    if (batch == 0) return LKG_OK;
    LKG_REQUIRE(emb && relemb && h && r && pos_t && neg_t && pos && neg && g_loss && g_emb && g_rel,
                "lkg_transe_score_bwd_f32: null pointer");
    hipLaunchKernelGGL((score_bwd_kernel<false>), dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (long)batch, dim, emb, nullptr, nullptr, (long)ld_emb, relemb, (long)ld_rel, (const long *)h,
                       (const long *)r, (const long *)pos_t, (const long *)neg_t, pos, neg, lambda, g_loss, g_emb,
                       nullptr, nullptr, (long)ld_gemb, g_rel, (long)ld_grel);
    LKG_CHECK_LAUNCH("lkg_transe_score_bwd_f32");
    return LKG_OK;
}

extern "C" int lkg_dense_score_bwd_f32(int64_t batch, int32_t rows_per_group, int32_t dim, const float *ph,
                                       const float *pp, const float *pn, int64_t ld, const float *relemb,
                                       int64_t ld_rel, const int64_t *r, const float *pos, const float *neg,
                                       float lambda, const float *g_loss, float *g_ph, float *g_pp, float *g_pn,
                                       int64_t ldg, float *g_rel, int64_t ld_grel, void *stream) {
    LKG_REQUIRE(batch >= 0 && dim > 0 && ld >= dim && ld_rel >= dim && ldg >= dim && ld_grel >= dim,
                "lkg_dense_score_bwd_f32: bad sizes");
    LKG_REQUIRE(rows_per_group >= 1 && batch % rows_per_group == 0,
                "lkg_dense_score_bwd_f32: batch %lld is not a whole number of groups of %d rows", (long long)batch,
                rows_per_group);
    if (batch == 0) return LKG_OK;
    LKG_REQUIRE(ph && pp && pn && relemb && r && pos && neg && g_loss && g_ph && g_pp && g_pn && g_rel,
                "lkg_dense_score_bwd_f32: null pointer");
    if (rows_per_group > 1) {
        const long n_groups = batch / rows_per_group;
        if (dim <= 256)
            hipLaunchKernelGGL((score_bwd_grouped_kernel<4>), dim3((unsigned)n_groups), dim3(256), 0,
                               (hipStream_t)stream, n_groups, rows_per_group, dim, ph, pp, pn, (long)ld, relemb,
                               (long)ld_rel, (const long *)r, pos, neg, lambda, g_loss, g_ph, g_pp, g_pn, (long)ldg,
                               g_rel, (long)ld_grel);
        else
            hipLaunchKernelGGL((score_bwd_grouped_kernel<8>), dim3((unsigned)n_groups), dim3(256), 0,
                               (hipStream_t)stream, n_groups, rows_per_group, dim, ph, pp, pn, (long)ld, relemb,
                               (long)ld_rel, (const long *)r, pos, neg, lambda, g_loss, g_ph, g_pp, g_pn, (long)ldg,
                               g_rel, (long)ld_grel);
        LKG_CHECK_LAUNCH("lkg_dense_score_bwd_f32");
        return LKG_OK;
    }
    hipLaunchKernelGGL((score_bwd_kernel<true>), dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (long)batch, dim, ph, pp, pn, (long)ld, relemb, (long)ld_rel, nullptr, (const long *)r, nullptr,
                       nullptr, pos, neg, lambda, g_loss, g_ph, g_pp, g_pn, (long)ldg, g_rel, (long)ld_grel);
    LKG_CHECK_LAUNCH("lkg_dense_score_bwd_f32");
    return LKG_OK;
}

extern "C" int lkg_dot_score_fwd_f32(int64_t batch, int32_t dim, const float *emb, int64_t ld_emb, const int64_t *h,
                                     const int64_t *pos_t, const int64_t *neg_t, float *pos, float *neg, float *reg,
                                     float *rank, void *stream) {
    LKG_REQUIRE(batch >= 0 && dim > 0 && ld_emb >= dim, "lkg_dot_score_fwd_f32: bad sizes");
    if (batch == 0) return LKG_OK;
    LKG_REQUIRE(emb && h && pos_t && neg_t && pos && neg && reg && rank, "lkg_dot_score_fwd_f32: null pointer");
    hipLaunchKernelGGL(dot_fwd_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (long)batch, dim, emb, (long)ld_emb, (const long *)h, (const long *)pos_t, (const long *)neg_t,
                       pos, neg, reg, rank);
    LKG_CHECK_LAUNCH("lkg_dot_score_fwd_f32");
    return LKG_OK;
}

extern "C" int lkg_dot_score_bwd_f32(int64_t batch, int32_t dim, const float *emb, int64_t ld_emb, const int64_t *h,
                                     const int64_t *pos_t, const int64_t *neg_t, const float *pos, const float *neg,
                                     float lambda, const float *g_loss, float *g_emb, int64_t ld_gemb, void *stream) {
    LKG_REQUIRE(batch >= 0 && dim > 0 && ld_emb >= dim && ld_gemb >= dim, "lkg_dot_score_bwd_f32: bad sizes");
    if (batch == 0) return LKG_OK;
    LKG_REQUIRE(emb && h && pos_t && neg_t && pos && neg && g_loss && g_emb, "lkg_dot_score_bwd_f32: null pointer");
    hipLaunchKernelGGL(dot_bwd_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (long)batch, dim, emb, (long)ld_emb, (const long *)h, (const long *)pos_t, (const long *)neg_t,
                       pos, neg, lambda, g_loss, g_emb, (long)ld_gemb);
    LKG_CHECK_LAUNCH("lkg_dot_score_bwd_f32");
    return LKG_OK;
}

// lkg_preload(): HIP loads a translation unit's code object on the first use of one of its kernels; asking for a kernel's
// attributes is such a use (no launch).
int lkg_internal_preload_score() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&loss_reduce_kernel)) == hipSuccess ? 0 : 1;
}
