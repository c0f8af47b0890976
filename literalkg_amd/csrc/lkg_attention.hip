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
        const float a = h.x + r.x, b = h.y + r.y, c = h.z + r.z, e = h.w + r.w;
        // embeddings are small (xavier init: |x| ~ 1e-2): when the whole wave is inside the series' range the
        // exp / rcp branch of tanh_fast is skipped -- same values, about half the VALU work
        const float big = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(e)));
        if (__builtin_amdgcn_ballot_w64(big >= 0.25f) == 0)
            return t.x * tanh_series(a) + t.y * tanh_series(b) + t.z * tanh_series(c) + t.w * tanh_series(e);
        return t.x * tanh_fast(a) + t.y * tanh_fast(b) + t.z * tanh_fast(c) + t.w * tanh_fast(e);
    }
    static __device__ __forceinline__ float4 zero() { return f4_zero(); }
};
template <>
struct dot_ops<float> {
    static __device__ __forceinline__ float tanh_dot(const float &t, const float &h, const float &r) {
        return t * tanh_fast(h + r);
    }
    static __device__ __forceinline__ float zero() { return 0.f; }
};

// Pre-pass for the (rare) stored entries whose (h,t) pair occurs under several relations: one wave per such entry
// writes the logit terms of its 2nd, 3rd ... raw edge to val_out[entry]; the main kernel adds the first edge's
// term on top.  Keeping this out of the main kernel keeps the hot loop at the no-duplicate register count.
template <typename V>
__global__ __launch_bounds__(256) void extra_relations_kernel(int n_dup, long row_offset, int nchunk,
                                                               const int *__restrict__ dup_entries,
                                                               const int *__restrict__ dup_rows,
                                                               const int *__restrict__ col,
                                                               const int *__restrict__ eptr,
                                                               const int *__restrict__ rel,
                                                               const float *__restrict__ ent, long ld_ent,
                                                               const float *__restrict__ relemb, long ld_rel,
                                                               float *__restrict__ val_out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n_dup) return;
    const int j = dup_entries[i];
    const V *hs = reinterpret_cast<const V *>(ent + (row_offset + dup_rows[i]) * ld_ent);
    const V *ts = reinterpret_cast<const V *>(ent + (long)col[j] * ld_ent);
    float extra = 0.f;
    for (int e = eptr[j] + 1; e < eptr[j + 1]; ++e) {
        const V *rs = reinterpret_cast<const V *>(relemb + (long)rel[e] * ld_rel);
        for (int ch = lane; ch < nchunk; ch += 64) extra += dot_ops<V>::tanh_dot(ts[ch], hs[ch], rs[ch]);
    }
    extra = wave_sum(extra);
    if (lane == 0) val_out[j] = extra;
}

// One launch, two kinds of workgroup (256 threads = 4 waves), as in lkg_spmm.hip:
//   blocks [0, n_long)   : one LONG head row (> long_thresh entries) per workgroup; the four waves take interleaved
//                          64-entry chunks and meet in LDS for the softmax statistics;
//   blocks [n_long, ...) : four ordinary rows, one wave each.
template <typename V, int LPE, int CPL, int U, bool DUPS, bool RLDS>
__global__ __launch_bounds__(256) void edge_softmax_kernel(
    int n_rows, long row_offset, int nchunk, const int *__restrict__ rowptr, const int *__restrict__ col,
    const int *__restrict__ eptr, const int *__restrict__ rel, const int *__restrict__ rel_first,
    const float *__restrict__ ent, long ld_ent, const float *__restrict__ relemb, long ld_rel,
    float *__restrict__ val_out, float *__restrict__ logits_out, const int *__restrict__ long_rows, int n_long,
    int long_thresh, int n_rel, int rows_per_wave) {
    using ops = dot_ops<V>;
    constexpr int EPW = 64 / LPE;
    __shared__ float red[4];
    extern __shared__ __align__(16) unsigned char rel_lds_bytes[];
    V *rel_lds = reinterpret_cast<V *>(rel_lds_bytes);
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const bool team = (int)blockIdx.x < n_long;      // workgroup-uniform
    if constexpr (RLDS) {
        // the relation table (n_rel x nchunk chunks) into LDS once per workgroup: every entry adds one of its rows to the
        // head's, and fetched per entry from global memory those rows cost the texture path as much as the tails do
        for (int i = threadIdx.x; i < n_rel * nchunk; i += 256) {
            const int rr = i / nchunk;
            rel_lds[i] = reinterpret_cast<const V *>(relemb + (long)rr * ld_rel)[i - rr * nchunk];
        }
        __syncthreads();
    }
    const int sub = lane / LPE;
    const int sl = lane % LPE;
    const int wave_i = team ? w : 0, n_waves = team ? 4 : 1;
    // lanes whose chunk index falls outside the row contribute nothing
    bool live[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) live[i] = (sl + i * LPE) < nchunk;

    // team: ONE long row; otherwise rows_per_wave rows per wave, the four waves interleaved over 4 * rows_per_wave
    // consecutive rows (1 without the LDS table: as many workgroups as row quadruples)
    const int first = team ? 0 : ((int)blockIdx.x - n_long) * 4 * rows_per_wave + w;
    for (int it = 0; it < (team ? 1 : rows_per_wave); ++it) {
    int row;
    if (team) {
        row = long_rows[blockIdx.x];
    } else {
        row = first + 4 * it;
        if (row >= n_rows) break;
    }
    const int start = __builtin_amdgcn_readfirstlane(rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
    if (start >= end) continue;
    if (!team && n_long > 0 && end - start > long_thresh) continue;

    // head embedding chunk(s) of this lane
    V hv[CPL];
    {
        const V *hs = reinterpret_cast<const V *>(ent + (row_offset + row) * ld_ent);
#pragma unroll
        for (int i = 0; i < CPL; ++i) hv[i] = hs[min(sl + i * LPE, nchunk - 1)];
    }
    // rows of at most 64 entries (nearly all of them) never leave the registers: lane i keeps the logit of
    // entry i and the softmax statistics are two wave reductions; longer rows park their logits in val_out
    const bool in_regs = !team && end - start <= 64;
    float mylogit = -INFINITY;

    for (int base = start + 64 * wave_i; base < end; base += 64 * n_waves) {
        const int cnt = min(64, end - base);
        const int last = cnt - 1;
        const int jl = base + min(lane, last);
        const int c = col[jl];
        // relation of the entry's FIRST raw edge: entry-indexed (rel_first) so that it does not wait for eptr
        const int r0 = DUPS ? rel_first[jl] : rel[jl];
        // terms of the entry's further raw edges, left in val_out by extra_relations_kernel (0 for most entries)
        const float pre = DUPS ? val_out[jl] : 0.f;
        for (int k = 0; k < cnt; k += EPW * U) {
            float part[U];
            int cc[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idc = min(k + u * EPW + sub, last);
                if constexpr (LPE == 64) {   // one entry per wave step: the index is wave-uniform, scalar broadcast
                    cc[u] = __builtin_amdgcn_readlane(c, idc);
                    rr[u] = __builtin_amdgcn_readlane(r0, idc);
                } else {
                    cc[u] = __shfl(c, idc, 64);
                    rr[u] = __shfl(r0, idc, 64);
                }
            }
            V tv[U][CPL], rv[U][CPL];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const V *ts = reinterpret_cast<const V *>(ent + (long)cc[u] * ld_ent);
                const V *rs = RLDS ? rel_lds + rr[u] * nchunk : reinterpret_cast<const V *>(relemb + (long)rr[u] * ld_rel);
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    const int chunk = min(sl + i * LPE, nchunk - 1);
                    tv[u][i] = ts[chunk];
                    if constexpr (!RLDS) rv[u][i] = rs[chunk];
                }
                if constexpr (RLDS) {        // (after the tails are requested: the LDS reads wait on lgkmcnt, not on them)
#pragma unroll
                    for (int i = 0; i < CPL; ++i) rv[u][i] = rs[min(sl + i * LPE, nchunk - 1)];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float p = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    // all lanes evaluate (tanh_dot holds a wave-wide ballot: under a lane predicate it would turn into
                    // an exec-masked branch per entry); lanes past the row end hold clamped copies and are dropped here
                    const float q = ops::tanh_dot(tv[u][i], hv[i], rv[u][i]);
                    p += live[i] ? q : 0.f;
                }
                part[u] = p;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = k + u * EPW + sub;
                float tot = group_sum<LPE>(part[u]);
                if constexpr (DUPS) {
                    if constexpr (LPE == 64)
                        tot += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pre), min(idx, last)));
                    else
                        tot += __shfl(pre, min(idx, last), 64);
                }
                if (in_regs) {
                    // entry e = k + u*EPW + s was summed by sub-group s: lane e fetches it from that group's lane 0
                    const int rel_e = lane - (k + u * EPW);
                    const float mine = (EPW == 1) ? tot : __shfl(tot, (rel_e & (EPW - 1)) * LPE, 64);
                    if (rel_e >= 0 && rel_e < EPW) mylogit = mine;
                    if (logits_out && sl == 0 && idx < cnt) logits_out[base + idx] = tot;
                } else if (sl == 0 && idx < cnt) {
                    val_out[base + idx] = tot;
                    if (logits_out) logits_out[base + idx] = tot;
                }
            }
        }
    }
    if (in_regs) {
        const bool has = lane < end - start;
        const float m = wave_max(has ? mylogit : -INFINITY);
        const float e = has ? expf(mylogit - m) : 0.f;
        const float s = wave_sum(e);
        if (has) val_out[start + lane] = e / s;
        continue;
    }
    // the logits were written by other lanes (team: other waves) of this workgroup: make them visible
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    if (team) __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const int tid = team ? (int)threadIdx.x : lane;
    const int ts = team ? 256 : 64;
    float m = -INFINITY;
    for (int j = start + tid; j < end; j += ts) m = fmaxf(m, val_out[j]);
    m = wave_max(m);
    if (team) {
        if (lane == 0) red[w] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
    }
    float s = 0.f;
    for (int j = start + tid; j < end; j += ts) s += expf(val_out[j] - m);
    s = wave_sum(s);
    if (team) {
        if (lane == 0) red[w] = s;
        __syncthreads();
        s = (red[0] + red[1]) + (red[2] + red[3]);
    }
    for (int j = start + tid; j < end; j += ts) val_out[j] = expf(val_out[j] - m) / s;
    }   // rows of this wave
}

struct EsArgs {
    int64_t n_rows, row_offset;
    int nchunk;
    const int *rowptr, *col, *eptr, *rel, *rel_first, *dup_entries, *dup_rows;
    int n_dup;
    int64_t entry_lo, entry_hi;
    const float *ent;
    int64_t ld_ent;
    const float *relemb;
    int64_t ld_rel;
    float *val_out, *logits_out;
    const int *long_rows;
    int n_long, long_thresh;
    int n_rel;
    size_t chunk_bytes;
};

// The relation table is staged in LDS when it fits REL_LDS_BYTES: up to there the 160 KB of a CU hold the tables of as many
// workgroups as the registers allow waves (8 per SIMD), and the refresh of the 10 M-edge graph at D = 256 with 16 relations
// goes from 2.02 to 1.85 ms; a 32 KB table (5 workgroups per CU) already LOSES to the rows from global memory (2.37 vs
// 2.10 ms), a 64 KB one takes 3.8 ms.  Each workgroup then works through ROWS_PER_WAVE_LDS rows per wave, so that the fill
// is read once per ~600 entries.
constexpr size_t REL_LDS_BYTES = 16 * 1024;
constexpr int ROWS_PER_WAVE_LDS = 16;

template <typename V, int LPE, int CPL, int U, bool DUPS, bool RLDS>
int launch3(const EsArgs &a, hipStream_t s) {
    const int rpw = RLDS ? ROWS_PER_WAVE_LDS : 1;
    const size_t lds = RLDS ? (size_t)a.n_rel * a.nchunk * a.chunk_bytes : 0;
    const int64_t blocks = (a.n_rows + 4 * rpw - 1) / (4 * rpw) + a.n_long;
    hipLaunchKernelGGL((edge_softmax_kernel<V, LPE, CPL, U, DUPS, RLDS>), dim3((unsigned)blocks), dim3(256), lds, s,
                       (int)a.n_rows, (long)a.row_offset, a.nchunk, a.rowptr, a.col, a.eptr, a.rel, a.rel_first, a.ent,
                       (long)a.ld_ent, a.relemb, (long)a.ld_rel, a.val_out, a.logits_out, a.long_rows, a.n_long,
                       a.long_thresh, a.n_rel, rpw);
    LKG_CHECK_LAUNCH("lkg_edge_softmax_f32");
    return LKG_OK;
}

template <typename V, int LPE, int CPL, int U, int U_LDS, bool DUPS>
int launch2(const EsArgs &a, hipStream_t s) {
    if constexpr (DUPS) {
        // only THIS call's entries: a row-range refresh into an existing value array leaves the other rows alone
        if (hipMemsetAsync(a.val_out + a.entry_lo, 0, sizeof(float) * (a.entry_hi - a.entry_lo), s) != hipSuccess) {
            lkg_set_error("lkg_edge_softmax_f32: hipMemsetAsync failed");
            return LKG_ERR_HIP;
        }
        if (a.n_dup > 0)
            hipLaunchKernelGGL((extra_relations_kernel<V>), dim3((unsigned)((a.n_dup + 3) / 4)), dim3(256), 0, s,
                               a.n_dup, (long)a.row_offset, a.nchunk, a.dup_entries, a.dup_rows, a.col, a.eptr, a.rel,
                               a.ent, (long)a.ld_ent, a.relemb, (long)a.ld_rel, a.val_out);
    }
    if (a.n_rel > 0 && (size_t)a.n_rel * a.nchunk * a.chunk_bytes <= REL_LDS_BYTES)
        return launch3<V, LPE, CPL, U_LDS, DUPS, true>(a, s);
    return launch3<V, LPE, CPL, U, DUPS, false>(a, s);
}

template <typename V, int LPE, int CPL, int U, int U_LDS>
int launch(const EsArgs &a, hipStream_t s) {
    return a.eptr ? launch2<V, LPE, CPL, U, U_LDS, true>(a, s) : launch2<V, LPE, CPL, U, U_LDS, false>(a, s);
}

template <typename V>
int dispatch(const EsArgs &a, hipStream_t s) {
    const int nchunk = a.nchunk;
    // U = 2 entries in flight per sub-group: measured best on one box (D=256 zipf: U=1 2.65 ms, U=2 2.29, U=4 2.63 --
    // at U=4 the 91 VGPRs of the tanh temporaries cut the occupancy to 5 waves per SIMD).  U_LDS: the same with the
    // relation rows out of LDS (D=256, 16 relations: 2, 3 and 4 within 1 % of each other at 1.85 ms, 6 slower)
    if (nchunk <= 8) return launch<V, 8, 1, 2, 2>(a, s);
    if (nchunk <= 16) return launch<V, 16, 1, 2, 2>(a, s);
    if (nchunk <= 32) return launch<V, 32, 1, 2, 2>(a, s);
    if (nchunk <= 64) return launch<V, 64, 1, 2, 2>(a, s);   // (after the VALU reductions: U=3 the same, U=4 slower)
    if (nchunk <= 128) return launch<V, 64, 2, 2, 2>(a, s);
    if (nchunk <= 192) return launch<V, 64, 3, 2, 2>(a, s);
    if (nchunk <= 256) return launch<V, 64, 4, 1, 1>(a, s);
    lkg_set_error("lkg_edge_softmax_f32: embedding width of %d chunks exceeds the supported 256", nchunk);
    return LKG_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int lkg_edge_softmax_f32(int64_t n_rows, int64_t row_offset, int32_t d, const int32_t *rowptr,
                                    const int32_t *col, const int32_t *eptr, const int32_t *rel,
                                    const int32_t *rel_first, const int32_t *dup_entries, const int32_t *dup_rows,
                                    int32_t n_dup, int64_t entry_lo, int64_t entry_hi, const float *ent,
                                    int64_t ld_ent,
                                    const float *relemb, int64_t ld_rel, float *val_out, float *logits_out,
                                    const int32_t *long_rows, int32_t n_long, int32_t long_thresh, int32_t n_rel,
                                    void *stream) {
    LKG_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX && row_offset >= 0, "lkg_edge_softmax_f32: bad row range");
    LKG_REQUIRE(n_rel >= 0, "lkg_edge_softmax_f32: negative relation count");
    LKG_REQUIRE(d > 0 && ld_ent >= d && ld_rel >= d, "lkg_edge_softmax_f32: bad d/strides (d=%d)", d);
    LKG_REQUIRE(n_long >= 0 && (n_long == 0 || (long_rows && long_thresh >= 64)),
                "lkg_edge_softmax_f32: long-row list needs a pointer and a threshold >= 64");
    if (n_rows == 0) return LKG_OK;
    LKG_REQUIRE(rowptr && col && rel && ent && relemb && val_out, "lkg_edge_softmax_f32: null pointer");
    LKG_REQUIRE(!eptr || (rel_first && entry_lo >= 0 && entry_hi >= entry_lo && n_dup >= 0 &&
                          (n_dup == 0 || (dup_entries && dup_rows))),
                "lkg_edge_softmax_f32: eptr needs rel_first, the entry range and the duplicate-entry list");
    const bool vec = (d % 4 == 0) && (ld_ent % 4 == 0) && (ld_rel % 4 == 0) && lkg_aligned16(ent) && lkg_aligned16(relemb);
    EsArgs a{n_rows, row_offset, vec ? d / 4 : d, rowptr, col, eptr, rel, rel_first, dup_entries, dup_rows, n_dup,
             entry_lo, entry_hi, ent, ld_ent, relemb, ld_rel,
             val_out, logits_out, long_rows, n_long, long_thresh, n_rel, vec ? sizeof(float4) : sizeof(float)};
    hipStream_t s = (hipStream_t)stream;
    return vec ? dispatch<float4>(a, s) : dispatch<float>(a, s);
}

// lkg_preload(): HIP loads a translation unit's code object on the first use of one of its kernels; asking for a kernel's
// attributes is such a use (no launch).
int lkg_internal_preload_attention() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&extra_relations_kernel<float4>)) == hipSuccess ? 0 : 1;
}
