// Device-side KG structure build: what lkg_csr_build / lkg_csr_transpose (lkg_graph_host.cpp) do on the host, as HIP
// kernels, so that the first update_att of an edge list (model.py:444-471: per-relation torch.where / cat / stack /
// sparse_coo_tensor / coalesce) costs milliseconds instead of a host sort.  Same outputs bit for bit:
//   triples stable-sorted by (head, tail) (ties keep input order), duplicate (head, tail) pairs merged into one
//   stored entry, rowptr / col / eptr / rel / order; CSC with heads ascending inside every tail.
//
// The sort is a hand-written LSD radix sort (8-bit digits) over the 64-bit key head * N + tail with the input position
// as payload: ceil(log2(N^2) / 8) stable passes (5 at N = 1 M, 6 at 5 M).  Per pass:
//   rs_hist     one 256-bin histogram per 2048-key tile (LDS atomics), written digit-major [256][tiles]
//   scan        exclusive scan of that array = first output position of every (digit, tile)
//   rs_scatter  the tile again: every key's rank among EARLIER keys of the tile with the same digit -- inside a wave by
//               8 ballots (one per digit bit), across the 32 (round, wave) segments of the tile by a per-digit prefix
//               in LDS -- then one scattered store per key.  No atomics on the output: deterministic and stable.
// HBM-bound integer work: 8 B + 12 B read and 12 B written per key and pass.
#include <algorithm>

#include "lkg_common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int RS_THREADS = 256, RS_ROUNDS = 8, RS_TILE = RS_THREADS * RS_ROUNDS, RS_SEGS = RS_ROUNDS * 4;
constexpr int SC_THREADS = 256, SC_ITEMS = 8, SC_TILE = SC_THREADS * SC_ITEMS;

inline long ceil_div(long a, long b) { return (a + b - 1) / b; }
inline long align_up(long a, long b) { return ceil_div(a, b) * b; }

// ---------------------------------------------------------------------------------------------- exclusive scan (int32)
// three launches: tile totals -> scan of the totals (one workgroup) -> tiles rescanned with their offset.
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds /*[SC_THREADS/64 + 1]*/, int &total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
        const int s = lds[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SC_THREADS) void scan_totals_kernel(const int *__restrict__ in, long n, int *__restrict__ sums) {
    __shared__ int lds[8];
    const long base = (long)blockIdx.x * SC_TILE + (long)threadIdx.x * SC_ITEMS;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; ++j) s += (base + j < n) ? in[base + j] : 0;
    int tot;
    block_exclusive_scan(s, lds, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// one workgroup: sums[i] <- exclusive prefix, total_out (nullable) <- grand total (+ total_add, nullable int64 out)
__global__ __launch_bounds__(1024) void scan_top_kernel(int *__restrict__ sums, long n, long *__restrict__ total_out) {
    __shared__ int lds[20];
    long carry = 0;
    for (long c0 = 0; c0 < n; c0 += 1024) {
        const long i = c0 + threadIdx.x;
        const int v = i < n ? sums[i] : 0;
        int tot;
        const int ex = block_exclusive_scan(v, lds, tot);
        if (i < n) sums[i] = (int)(carry + ex);
        carry += tot;
    }
    if (total_out && threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SC_THREADS) void scan_apply_kernel(const int *__restrict__ in, long n,
                                                                const int *__restrict__ sums, int *__restrict__ out) {
    __shared__ int lds[8];
    const long base = (long)blockIdx.x * SC_TILE + (long)threadIdx.x * SC_ITEMS;
    int v[SC_ITEMS], s = 0;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0;
        s += v[j];
    }
    int tot;
    int run = block_exclusive_scan(s, lds, tot) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SC_ITEMS; ++j) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
}

// out may alias in; sums: workspace of ceil(n / SC_TILE) ints; total_out: nullable device int64
int exclusive_scan(const int *in, int *out, long n, int *sums, long *total_out, hipStream_t s) {
    if (n == 0) return LKG_OK;
    const long nb = ceil_div(n, SC_TILE);
    hipLaunchKernelGGL(scan_totals_kernel, dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, n, sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, s, sums, nb, total_out);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, n, sums, out);
    LKG_CHECK_LAUNCH("lkg_csr_build_device (scan)");
    return LKG_OK;
}

// ---------------------------------------------------------------------------------------------- LSD radix sort pass
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const u64 *__restrict__ keys, long n, int shift,
                                                             int *__restrict__ hist, int n_tiles) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const long i = base + r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255)], 1);
    }
    __syncthreads();
    hist[(long)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// vals_in == nullptr: the payload is the input position (first pass)
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const u64 *__restrict__ keys_in,
                                                                const u32 *__restrict__ vals_in,
                                                                u64 *__restrict__ keys_out, u32 *__restrict__ vals_out,
                                                                long n, int shift, const int *__restrict__ offsets,
                                                                int n_tiles) {
    __shared__ int seg[RS_SEGS][256];     // per (round, wave) segment: keys per digit, then their first output position
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < RS_SEGS * 256; i += RS_THREADS) (&seg[0][0])[i] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * RS_TILE;
    u64 key[RS_ROUNDS];
    int rank[RS_ROUNDS];
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const long i = base + r * RS_THREADS + tid;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0ull;
        const int d = (int)((key[r] >> shift) & 255);
        u64 peers = __ballot(valid);          // lanes of this wave holding the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const u64 m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        rank[r] = __popcll(peers & lt);
        if (valid && rank[r] == 0) seg[r * 4 + wave][d] = __popcll(peers);
    }
    __syncthreads();
    {   // thread = digit: running first position over the tile's segments in element order
        int run = offsets[(long)tid * n_tiles + blockIdx.x];
#pragma unroll 4
        for (int s = 0; s < RS_SEGS; ++s) {
            const int c = seg[s][tid];
            seg[s][tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const long i = base + r * RS_THREADS + tid;
        if (i < n) {
            const int d = (int)((key[r] >> shift) & 255);
            const long pos = (long)seg[r * 4 + wave][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = vals_in ? vals_in[i] : (u32)i;
        }
    }
}

struct SortWs {
    u64 *k0, *k1;
    u32 *v0, *v1;
    int *hist;     // [256][n_tiles]
    int *sums;     // scan workspace
};

long sort_ws_bytes(long n) {
    const long n_tiles = ceil_div(std::max<long>(n, 1), RS_TILE);
    return align_up(2 * 8 * n, 256) + align_up(2 * 4 * n, 256) + align_up(4 * 256 * n_tiles, 256) +
           align_up(4 * ceil_div(256 * n_tiles, SC_TILE) + 4 * ceil_div(std::max<long>(n, 1), SC_TILE) + 64, 256);
}

SortWs carve(char *ws, long n) {
    const long n_tiles = ceil_div(std::max<long>(n, 1), RS_TILE);
    SortWs w;
    w.k0 = (u64 *)ws;
    w.k1 = w.k0 + n;
    ws += align_up(2 * 8 * n, 256);
    w.v0 = (u32 *)ws;
    w.v1 = w.v0 + n;
    ws += align_up(2 * 4 * n, 256);
    w.hist = (int *)ws;
    ws += align_up(4 * 256 * n_tiles, 256);
    w.sums = (int *)ws;
    return w;
}

// sorts w.k0 (payload = position) on key bits [0, bits); returns in *kout / *vout the buffers holding the result
int radix_sort(const SortWs &w, long n, int bits, u64 **kout, u32 **vout, hipStream_t s) {
    const int n_tiles = (int)ceil_div(n, RS_TILE);
    u64 *kin = w.k0, *ko = w.k1;
    u32 *vin = nullptr, *vo = w.v0;          // first pass writes v0 from positions
    u32 *spare = w.v1;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(n_tiles), dim3(RS_THREADS), 0, s, kin, n, shift, w.hist, n_tiles);
        int rc = exclusive_scan(w.hist, w.hist, 256L * n_tiles, w.sums, nullptr, s);
        if (rc != LKG_OK) return rc;
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(n_tiles), dim3(RS_THREADS), 0, s, kin, vin, ko, vo, n, shift,
                           w.hist, n_tiles);
        std::swap(kin, ko);
        u32 *done = vo;
        vo = vin ? vin : spare;
        vin = done;
    }
    LKG_CHECK_LAUNCH("lkg_csr_build_device (sort)");
    *kout = kin;
    *vout = vin;
    return LKG_OK;
}

int key_bits(u64 max_key) {
    int b = 1;
    while (b < 64 && (max_key >> b)) ++b;
    return b;
}

// ---------------------------------------------------------------------------------------------- structure kernels
__global__ void make_keys_kernel(long n_edges, long n_ent, const long *__restrict__ h, const long *__restrict__ t,
                                 const long *__restrict__ r, u64 *__restrict__ keys, long *__restrict__ n_bad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n_edges) {
        const long hh = h[i], tt = t[i];
        bad = (u64)hh >= (u64)n_ent || (u64)tt >= (u64)n_ent || (r && (r[i] < 0 || r[i] > 0x7fffffffL));
        keys[i] = bad ? 0ull : (u64)hh * (u64)n_ent + (u64)tt;
    }
    const u64 m = __ballot(bad);
    if (m && (threadIdx.x & 63) == 0) atomicAdd((u64 *)n_bad, (u64)__popcll(m));
}

// flag[k] = 1 where sorted key k starts a new (head, tail) pair; rel / order in sorted order
__global__ void mark_entries_kernel(long n_edges, const u64 *__restrict__ keys, const u32 *__restrict__ pos,
                                    const long *__restrict__ r, int *__restrict__ flag, int *__restrict__ rel,
                                    int *__restrict__ order) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    flag[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
    const u32 e = pos[k];
    rel[k] = r ? (int)r[e] : 0;
    order[k] = (int)e;
}

// entry id of sorted key k = exclusive scan of the flags; the flagged positions write col / eptr
__global__ void emit_entries_kernel(long n_edges, long n_ent, const u64 *__restrict__ keys, const int *__restrict__ flag,
                                    const int *__restrict__ ent, int *__restrict__ col, int *__restrict__ eptr,
                                    const long *__restrict__ nnz) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) eptr[*nnz] = (int)n_edges;
    if (k >= n_edges || !flag[k]) return;
    const int j = ent[k];
    col[j] = (int)(keys[k] % (u64)n_ent);
    eptr[j] = (int)k;
}

// ptr[i] = scan value at the first sorted position whose key is >= i * mult (or `total` past the end): row pointers of
// the stored entries (via `ent`) or, with ent == nullptr, plain lower bounds (CSC pointers)
__global__ void lower_bound_ptr_kernel(long n_rows, u64 mult, const u64 *__restrict__ keys, long n_keys,
                                       const int *__restrict__ ent, const long *__restrict__ total_dev, long total_host,
                                       int *__restrict__ ptr) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_rows) return;
    const u64 want = (u64)i * mult;
    long lo = 0, hi = n_keys;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    const long total = total_dev ? *total_dev : total_host;
    ptr[i] = lo < n_keys ? (ent ? ent[lo] : (int)lo) : (int)total;
}

// keys[j] = col[j] (transpose sort key), rows[j] = head row of CSR entry j
__global__ void entry_rows_kernel(long nnz, long n_rows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                  u64 *__restrict__ keys, int *__restrict__ rows) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nnz) return;
    long lo = 0, hi = n_rows;       // last row with rowptr[row] <= j
    while (hi - lo > 1) {
        const long mid = (lo + hi) >> 1;
        if (rowptr[mid] <= j) lo = mid; else hi = mid;
    }
    rows[j] = (int)lo;
    keys[j] = (u64)(u32)col[j];
}

__global__ void emit_transpose_kernel(long nnz, const u32 *__restrict__ perm, const int *__restrict__ rows,
                                      int *__restrict__ t_col, int *__restrict__ t_perm) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const u32 j = perm[k];
    t_perm[k] = (int)j;
    t_col[k] = rows[j];
}

inline dim3 grid1d(long n) { return dim3((unsigned)ceil_div(std::max<long>(n, 1), 256)); }

}  // namespace

extern "C" int64_t lkg_csr_build_device_workspace(int64_t n_entities, int64_t n_edges) {
    (void)n_entities;
    const long e = std::max<long>(n_edges, 1);
    return sort_ws_bytes(e) + align_up(4 * e, 256) * 2 + 256;      // + flags + entry ids
}

extern "C" int lkg_csr_build_device(int64_t n_entities, int64_t n_edges, const int64_t *h, const int64_t *t,
                                    const int64_t *r, int32_t *rowptr, int32_t *col, int32_t *eptr, int32_t *rel,
                                    int32_t *order, int64_t *counts, void *workspace, int64_t workspace_bytes,
                                    void *stream) {
    LKG_REQUIRE(n_entities >= 0 && n_entities < INT32_MAX, "lkg_csr_build_device: n_entities %lld out of int32 range",
                (long long)n_entities);
    LKG_REQUIRE(n_edges >= 0 && n_edges < INT32_MAX, "lkg_csr_build_device: n_edges %lld out of int32 range",
                (long long)n_edges);
    LKG_REQUIRE(rowptr && counts && eptr, "lkg_csr_build_device: null pointer");
    LKG_REQUIRE(n_edges == 0 || (h && t && col && rel && order && workspace), "lkg_csr_build_device: null pointer");
    LKG_REQUIRE(workspace_bytes >= lkg_csr_build_device_workspace(n_entities, n_edges),
                "lkg_csr_build_device: workspace of %lld bytes is smaller than the %lld required",
                (long long)workspace_bytes, (long long)lkg_csr_build_device_workspace(n_entities, n_edges));
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, 2 * sizeof(int64_t), s) != hipSuccess) {
        lkg_set_error("lkg_csr_build_device: hipMemsetAsync failed");
        return LKG_ERR_HIP;
    }
    if (n_edges == 0) {
        if (hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (n_entities + 1), s) != hipSuccess ||
            hipMemsetAsync(eptr, 0, sizeof(int32_t), s) != hipSuccess) {
            lkg_set_error("lkg_csr_build_device: hipMemsetAsync failed");
            return LKG_ERR_HIP;
        }
        return LKG_OK;
    }
    const long e = n_edges;
    char *ws = (char *)workspace;
    const SortWs w = carve(ws, e);
    int *flag = (int *)(ws + sort_ws_bytes(e));
    int *ent = flag + align_up(4 * e, 256) / 4;
    long *nnz_dev = (long *)counts, *bad_dev = nnz_dev + 1;
    hipLaunchKernelGGL(make_keys_kernel, grid1d(e), dim3(256), 0, s, e, (long)n_entities, (const long *)h,
                       (const long *)t, (const long *)r, w.k0, bad_dev);
    u64 *keys;
    u32 *pos;
    const u64 max_key = (u64)n_entities * (u64)n_entities - 1ull;
    int rc = radix_sort(w, e, key_bits(max_key), &keys, &pos, s);
    if (rc != LKG_OK) return rc;
    hipLaunchKernelGGL(mark_entries_kernel, grid1d(e), dim3(256), 0, s, e, keys, pos, (const long *)r, flag, rel, order);
    rc = exclusive_scan(flag, ent, e, w.sums, nnz_dev, s);
    if (rc != LKG_OK) return rc;
    hipLaunchKernelGGL(emit_entries_kernel, grid1d(e), dim3(256), 0, s, e, (long)n_entities, keys, flag, ent, col, eptr,
                       nnz_dev);
    hipLaunchKernelGGL(lower_bound_ptr_kernel, grid1d(n_entities + 1), dim3(256), 0, s, (long)n_entities,
                       (u64)n_entities, keys, e, ent, nnz_dev, 0L, rowptr);
    LKG_CHECK_LAUNCH("lkg_csr_build_device");
    return LKG_OK;
}

extern "C" int64_t lkg_csr_transpose_device_workspace(int64_t n_cols, int64_t nnz) {
    (void)n_cols;
    const long e = std::max<long>(nnz, 1);
    return sort_ws_bytes(e) + align_up(4 * e, 256) + 256;           // + head row of every entry
}

extern "C" int lkg_csr_transpose_device(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr,
                                        const int32_t *col, int32_t *t_rowptr, int32_t *t_col, int32_t *t_perm,
                                        void *workspace, int64_t workspace_bytes, void *stream) {
    LKG_REQUIRE(n_rows >= 0 && n_cols >= 0 && n_cols < INT32_MAX && nnz >= 0 && nnz < INT32_MAX,
                "lkg_csr_transpose_device: bad sizes");
    LKG_REQUIRE(rowptr && t_rowptr && (nnz == 0 || (col && t_col && t_perm && workspace)),
                "lkg_csr_transpose_device: null pointer");
    LKG_REQUIRE(workspace_bytes >= lkg_csr_transpose_device_workspace(n_cols, nnz),
                "lkg_csr_transpose_device: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (nnz == 0) {
        if (hipMemsetAsync(t_rowptr, 0, sizeof(int32_t) * (n_cols + 1), s) != hipSuccess) {
            lkg_set_error("lkg_csr_transpose_device: hipMemsetAsync failed");
            return LKG_ERR_HIP;
        }
        return LKG_OK;
    }
    char *ws = (char *)workspace;
    const SortWs w = carve(ws, nnz);
    int *rows = (int *)(ws + sort_ws_bytes(nnz));
    hipLaunchKernelGGL(entry_rows_kernel, grid1d(nnz), dim3(256), 0, s, (long)nnz, (long)n_rows, rowptr, col, w.k0,
                       rows);
    u64 *keys;
    u32 *perm;
    int rc = radix_sort(w, nnz, key_bits((u64)std::max<int64_t>(n_cols, 1) - 1ull), &keys, &perm, s);   // stable: heads stay ascending
    if (rc != LKG_OK) return rc;
    hipLaunchKernelGGL(emit_transpose_kernel, grid1d(nnz), dim3(256), 0, s, (long)nnz, perm, rows, t_col, t_perm);
    hipLaunchKernelGGL(lower_bound_ptr_kernel, grid1d(n_cols + 1), dim3(256), 0, s, (long)n_cols, 1ull, keys, (long)nnz,
                       (const int *)nullptr, (const long *)nullptr, (long)nnz, t_rowptr);
    LKG_CHECK_LAUNCH("lkg_csr_transpose_device");
    return LKG_OK;
}

// ---------------------------------------------------------------------------------------------------
// The loader's initial attention on the device (dataloader.py:449-495): A_in = sum_r norm(A_r) over the structure the
// device build left in HBM -- the host form (lkg_laplacian_f32) walks numpy mirrors of it.  Two passes, one wave per head
// row: out-degree of every (entity, relation) pair into deg[n x n_rel]; then every stored entry sums its raw edges'
// 1 / d_r(h)  (random-walk)  or  d_r(h)^-1/2 d_r(t)^-1/2  (symmetric, the ROW sums on both sides as the reference) in f64,
// the arithmetic (and so the bits) of the host form.
namespace {
__global__ __launch_bounds__(256) void lap_degree_kernel(long n, int n_rel, const int *__restrict__ rowptr,
                                                          const int *__restrict__ eptr, const int *__restrict__ rel,
                                                          int *__restrict__ deg) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int j0 = rowptr[i], j1 = rowptr[i + 1];
    if (j0 == j1) return;
    const int e0 = eptr ? eptr[j0] : j0, e1 = eptr ? eptr[j1] : j1;       // the row's raw edges are contiguous
    for (int e = e0 + lane; e < e1; e += 64) atomicAdd(deg + i * n_rel + rel[e], 1);
}

__global__ __launch_bounds__(256) void lap_values_kernel(long n, int n_rel, int kind, const int *__restrict__ rowptr,
                                                          const int *__restrict__ col, const int *__restrict__ eptr,
                                                          const int *__restrict__ rel, const int *__restrict__ deg,
                                                          float *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int j0 = rowptr[i], j1 = rowptr[i + 1];
    for (int j = j0 + lane; j < j1; j += 64) {
        const int e0 = eptr ? eptr[j] : j, e1 = eptr ? eptr[j + 1] : j + 1;
        double acc = 0.0;
        for (int e = e0; e < e1; ++e) {
            const double dh = deg[i * n_rel + rel[e]];
            if (kind == 0) {
                acc += 1.0 / dh;
            } else {
                const double dt = deg[(long)col[j] * n_rel + rel[e]];
                if (dt > 0) acc += 1.0 / sqrt(dh) / sqrt(dt);
            }
        }
        val[j] = (float)acc;
    }
}
}  // namespace

extern "C" int lkg_laplacian_device_f32(int64_t n_entities, int64_t nnz, int32_t n_rel, const int32_t *rowptr,
                                        const int32_t *col, const int32_t *eptr, const int32_t *rel, int32_t kind,
                                        int32_t *deg_workspace, float *val_out, void *stream) {
    LKG_REQUIRE(n_entities >= 0 && n_entities < INT32_MAX && nnz >= 0 && n_rel > 0 && (kind == 0 || kind == 1),
                "lkg_laplacian_device_f32: bad arguments");
    if (nnz == 0 || n_entities == 0) return LKG_OK;
    LKG_REQUIRE(rowptr && col && rel && deg_workspace && val_out, "lkg_laplacian_device_f32: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(deg_workspace, 0, sizeof(int32_t) * (size_t)n_entities * n_rel, s) != hipSuccess) {
        lkg_set_error("lkg_laplacian_device_f32: hipMemsetAsync failed");
        return LKG_ERR_HIP;
    }
    const unsigned blocks = (unsigned)((n_entities + 3) / 4);
    hipLaunchKernelGGL(lap_degree_kernel, dim3(blocks), dim3(256), 0, s, (long)n_entities, n_rel, rowptr, eptr, rel,
                       deg_workspace);
    hipLaunchKernelGGL(lap_values_kernel, dim3(blocks), dim3(256), 0, s, (long)n_entities, n_rel, kind, rowptr, col, eptr,
                       rel, deg_workspace, val_out);
    LKG_CHECK_LAUNCH("lkg_laplacian_device_f32");
    return LKG_OK;
}

// ---------------------------------------------------------------------------------------------------
// The int64 [2, nnz] index tensor of the reference's coalesced A_in (model.py:462-468: rows = heads, cols = tails, sorted by
// (row, col)) straight from the CSR: one wave per head row writes its entries' (row, col) pairs.
namespace {
__global__ __launch_bounds__(256) void coo_indices_kernel(long n, long nnz, const int *__restrict__ rowptr,
                                                           const int *__restrict__ col, long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int j0 = rowptr[i], j1 = rowptr[i + 1];
    for (int j = j0 + lane; j < j1; j += 64) {
        out[j] = i;
        out[nnz + j] = col[j];
    }
}
}  // namespace

extern "C" int lkg_csr_coo_indices_i64(int64_t n_rows, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                                       int64_t *indices_out, void *stream) {
    LKG_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX && nnz >= 0, "lkg_csr_coo_indices_i64: bad sizes");
    if (nnz == 0 || n_rows == 0) return LKG_OK;
    LKG_REQUIRE(rowptr && col && indices_out, "lkg_csr_coo_indices_i64: null pointer");
    hipLaunchKernelGGL(coo_indices_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (long)n_rows,
                       (long)nnz, rowptr, col, reinterpret_cast<long *>(indices_out));
    LKG_CHECK_LAUNCH("lkg_csr_coo_indices_i64");
    return LKG_OK;
}

// lkg_preload(): HIP loads a translation unit's code object on the first use of one of its kernels; asking for a kernel's
// attributes is such a use (no launch).
int lkg_internal_preload_csr_device() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&scan_totals_kernel)) == hipSuccess ? 0 : 1;
}
