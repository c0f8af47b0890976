// K1 + K2 fused: attention refresh (model.py:430-471).
//   logit of stored entry j of head row h:  sum over its raw edges e of
//        sum_d ent[t_j,d] * tanh(ent[h,d] + relemb[rel[e],d])
//   (several raw edges per entry only when the same (h,t) pair occurs under several relations:
//    coalesce() sums them in logit space before the softmax)
//   value = softmax of the logits over the stored entries of row h  (torch.sparse.softmax dim=1).
//
// One wave per head row: the head's embedding chunk stays in registers for the whole row, tails are
// gathered as whole 16-byte-per-lane rows like in lkg_spmm.hip, the per-edge dot product is finished
// with xor-shuffles inside the LPE-lane sub-group.  The logits are parked in val_out, then the same
// wave runs the row softmax over them (coalesced, 64 entries per step).
#include <algorithm>

#include "lkg_common.h"

namespace {

template <typename V>
struct dot_ops;
template <>
struct dot_ops<float4> {
    static __device__ __forceinline__ float tanh_dot(const float4 &t, const float4 &h, const float4 &r) {
        return t.x * tanhf(h.x + r.x) + t.y * tanhf(h.y + r.y) + t.z * tanhf(h.z + r.z) + t.w * tanhf(h.w + r.w);
    }
    static __device__ __forceinline__ float4 zero() { return f4_zero(); }
};
template <>
struct dot_ops<float> {
    static __device__ __forceinline__ float tanh_dot(const float &t, const float &h, const float &r) {
        return t * tanhf(h + r);
    }
    static __device__ __forceinline__ float zero() { return 0.f; }
};

template <typename V, int LPE, int CPL, int U, bool DUPS>
__global__ __launch_bounds__(256) void edge_softmax_kernel(int n_rows, long row_offset, int nchunk,
                                                            const int *__restrict__ rowptr,
                                                            const int *__restrict__ col,
                                                            const int *__restrict__ eptr,
                                                            const int *__restrict__ rel,
                                                            const float *__restrict__ ent, long ld_ent,
                                                            const float *__restrict__ relemb, long ld_rel,
                                                            float *__restrict__ val_out,
                                                            float *__restrict__ logits_out) {
    using ops = dot_ops<V>;
    constexpr int EPW = 64 / LPE;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int start = __builtin_amdgcn_readfirstlane(rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
    if (start >= end) return;
    const int sub = lane / LPE;
    const int sl = lane % LPE;

    // head embedding chunk(s) of this lane
    V hv[CPL];
    {
        const V *hs = reinterpret_cast<const V *>(ent + (row_offset + row) * ld_ent);
#pragma unroll
        for (int i = 0; i < CPL; ++i) hv[i] = hs[min(sl + i * LPE, nchunk - 1)];
    }
    // lanes whose chunk index falls outside the row contribute nothing
    bool live[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) live[i] = (sl + i * LPE) < nchunk;

    for (int base = start; base < end; base += 64) {
        const int cnt = min(64, end - base);
        const int last = cnt - 1;
        const int jl = base + min(lane, last);
        const int c = col[jl];
        int e0 = jl, e1 = jl + 1;
        if constexpr (DUPS) {
            e0 = eptr[jl];
            e1 = eptr[jl + 1];
        }
        const int r0 = rel[e0];
        for (int k = 0; k < cnt; k += EPW * U) {
            float part[U];
            int cc[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idc = min(k + u * EPW + sub, last);
                cc[u] = __shfl(c, idc, 64);
                rr[u] = __shfl(r0, idc, 64);
            }
            V tv[U][CPL], rv[U][CPL];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const V *ts = reinterpret_cast<const V *>(ent + (long)cc[u] * ld_ent);
                const V *rs = reinterpret_cast<const V *>(relemb + (long)rr[u] * ld_rel);
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    const int chunk = min(sl + i * LPE, nchunk - 1);
                    tv[u][i] = ts[chunk];
                    rv[u][i] = rs[chunk];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float p = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; ++i) p += live[i] ? ops::tanh_dot(tv[u][i], hv[i], rv[u][i]) : 0.f;
                part[u] = p;
            }
            if constexpr (DUPS) {
                // rare: further raw edges of the same (h,t) entry, other relations
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idc = min(k + u * EPW + sub, last);
                    const int f0 = __shfl(e0, idc, 64), f1 = __shfl(e1, idc, 64);
                    for (int e = f0 + 1; e < f1; ++e) {
                        const V *rs = reinterpret_cast<const V *>(relemb + (long)rel[e] * ld_rel);
#pragma unroll
                        for (int i = 0; i < CPL; ++i)
                            part[u] += live[i] ? ops::tanh_dot(tv[u][i], hv[i], rs[min(sl + i * LPE, nchunk - 1)]) : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float tot = group_sum<LPE>(part[u]);
                const int idx = k + u * EPW + sub;
                if (sl == 0 && idx < cnt) {
                    val_out[base + idx] = tot;
                    if (logits_out) logits_out[base + idx] = tot;
                }
            }
        }
    }
    // the logits were written by other lanes of this wave: make them visible before re-reading
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    float m = -INFINITY;
    for (int j = start + lane; j < end; j += 64) m = fmaxf(m, val_out[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = start + lane; j < end; j += 64) s += expf(val_out[j] - m);
    s = wave_sum(s);
    for (int j = start + lane; j < end; j += 64) val_out[j] = expf(val_out[j] - m) / s;
}

template <typename V, int LPE, int CPL, int U>
int launch(bool dups, int64_t n_rows, int64_t row_offset, int nchunk, const int *rowptr, const int *col,
           const int *eptr, const int *rel, const float *ent, int64_t ld_ent, const float *relemb, int64_t ld_rel,
           float *val_out, float *logits_out, hipStream_t s) {
    const int64_t blocks = (n_rows + 3) / 4;
    if (dups)
        hipLaunchKernelGGL((edge_softmax_kernel<V, LPE, CPL, U, true>), dim3((unsigned)blocks), dim3(256), 0, s,
                           (int)n_rows, (long)row_offset, nchunk, rowptr, col, eptr, rel, ent, (long)ld_ent, relemb,
                           (long)ld_rel, val_out, logits_out);
    else
        hipLaunchKernelGGL((edge_softmax_kernel<V, LPE, CPL, U, false>), dim3((unsigned)blocks), dim3(256), 0, s,
                           (int)n_rows, (long)row_offset, nchunk, rowptr, col, eptr, rel, ent, (long)ld_ent, relemb,
                           (long)ld_rel, val_out, logits_out);
    LKG_CHECK_LAUNCH("lkg_edge_softmax_f32");
    return LKG_OK;
}

template <typename V>
int dispatch(bool dups, int64_t n_rows, int64_t row_offset, int nchunk, const int *rowptr, const int *col,
             const int *eptr, const int *rel, const float *ent, int64_t ld_ent, const float *relemb,
             int64_t ld_rel, float *val_out, float *logits_out, hipStream_t s) {
#define LKG_GO(LPE, CPL, U)                                                                                 \
    return launch<V, LPE, CPL, U>(dups, n_rows, row_offset, nchunk, rowptr, col, eptr, rel, ent, ld_ent, relemb, \
                                  ld_rel, val_out, logits_out, s)
    if (nchunk <= 8) LKG_GO(8, 1, 4);
    if (nchunk <= 16) LKG_GO(16, 1, 4);
    if (nchunk <= 32) LKG_GO(32, 1, 4);
    if (nchunk <= 64) LKG_GO(64, 1, 4);
    if (nchunk <= 128) LKG_GO(64, 2, 2);
    if (nchunk <= 192) LKG_GO(64, 3, 2);
    if (nchunk <= 256) LKG_GO(64, 4, 1);
#undef LKG_GO
    lkg_set_error("lkg_edge_softmax_f32: embedding width of %d chunks exceeds the supported 256", nchunk);
    return LKG_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int lkg_edge_softmax_f32(int64_t n_rows, int64_t row_offset, int32_t d, const int32_t *rowptr,
                                    const int32_t *col, const int32_t *eptr, const int32_t *rel, const float *ent,
                                    int64_t ld_ent, const float *relemb, int64_t ld_rel, float *val_out,
                                    float *logits_out, void *stream) {
    LKG_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX && row_offset >= 0, "lkg_edge_softmax_f32: bad row range");
    LKG_REQUIRE(d > 0 && ld_ent >= d && ld_rel >= d, "lkg_edge_softmax_f32: bad d/strides (d=%d)", d);
    if (n_rows == 0) return LKG_OK;
    LKG_REQUIRE(rowptr && col && rel && ent && relemb && val_out, "lkg_edge_softmax_f32: null pointer");
    const bool vec = (d % 4 == 0) && (ld_ent % 4 == 0) && (ld_rel % 4 == 0) && lkg_aligned16(ent) && lkg_aligned16(relemb);
    hipStream_t s = (hipStream_t)stream;
    return vec ? dispatch<float4>(eptr != nullptr, n_rows, row_offset, d / 4, rowptr, col, eptr, rel, ent, ld_ent,
                                  relemb, ld_rel, val_out, logits_out, s)
               : dispatch<float>(eptr != nullptr, n_rows, row_offset, d, rowptr, col, eptr, rel, ent, ld_ent, relemb,
                                 ld_rel, val_out, logits_out, s);
}
