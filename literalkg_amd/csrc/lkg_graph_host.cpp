// Host-side KG structure build: (head, tail)-sorted CSR with merged duplicate pairs,
// its CSC transpose, and nnz-balanced row-range cuts.  Plain C++ (no device code).
//
// What it replaces in the reference: the per-relation torch.where / cat / stack /
// sparse COO assembly + coalesce() of LiteralKG.update_attention (model.py:451-470)
// and the scipy COO -> tensor assembly of DataLoader (dataloader.py:449-495).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "lkg_common.h"

static thread_local char g_err[512] = "";

void lkg_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *lkg_last_error(void) { return g_err; }
extern "C" int lkg_version(void) { return 100; /* 0.1.0 */ }

namespace {
template <class F>
void parallel_rows(int64_t n, F &&fn) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 32u)), std::max<int64_t>(1, n / 65536));
    if (nt <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    int64_t step = (n + nt - 1) / nt;
    for (int i = 0; i < nt; ++i) {
        int64_t lo = i * step, hi = std::min(n, lo + step);
        if (lo < hi) th.emplace_back([=, &fn] { fn(lo, hi); });
    }
    for (auto &x : th) x.join();
}
}  // namespace

extern "C" int lkg_csr_build(int64_t n_entities, int64_t n_edges, const int64_t *h,
                             const int64_t *t, const int64_t *r, int32_t *rowptr, int32_t *col,
                             int32_t *eptr, int32_t *rel, int64_t *order, int64_t *nnz_out) {
    LKG_REQUIRE(n_entities >= 0 && n_entities < INT32_MAX, "lkg_csr_build: n_entities %lld out of int32 range",
                (long long)n_entities);
    LKG_REQUIRE(n_edges >= 0 && n_edges < INT32_MAX, "lkg_csr_build: n_edges %lld out of int32 range",
                (long long)n_edges);
    LKG_REQUIRE(rowptr && nnz_out && (n_edges == 0 || (h && t && col && eptr && rel && order)),
                "lkg_csr_build: null pointer");
    for (int64_t e = 0; e < n_edges; ++e) {
        if ((uint64_t)h[e] >= (uint64_t)n_entities || (uint64_t)t[e] >= (uint64_t)n_entities) {
            lkg_set_error("lkg_csr_build: edge %lld (%lld -> %lld) outside [0, %lld)", (long long)e,
                          (long long)h[e], (long long)t[e], (long long)n_entities);
            return LKG_ERR_INVALID_ARG;
        }
        if (r && (r[e] < 0 || r[e] > INT32_MAX)) {
            lkg_set_error("lkg_csr_build: relation id %lld of edge %lld out of range", (long long)r[e],
                          (long long)e);
            return LKG_ERR_INVALID_ARG;
        }
    }
    // 1. bucket raw edges by head (stable in input order)
    std::vector<int64_t> start;
    try {
        start.assign((size_t)n_entities + 1, 0);
    } catch (...) {
        lkg_set_error("lkg_csr_build: out of host memory");
        return LKG_ERR_NOMEM;
    }
    for (int64_t e = 0; e < n_edges; ++e) start[h[e] + 1]++;
    for (int64_t i = 0; i < n_entities; ++i) start[i + 1] += start[i];
    {
        std::vector<int64_t> pos(start.begin(), start.end() - 1);
        for (int64_t e = 0; e < n_edges; ++e) order[pos[h[e]]++] = e;
    }
    // 2. inside every head row, order by tail (ties keep input order)
    parallel_rows(n_entities, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            int64_t a = start[i], b = start[i + 1];
            if (b - a > 1)
                std::stable_sort(order + a, order + b, [&](int64_t x, int64_t y) { return t[x] < t[y]; });
        }
    });
    // 3. emit stored entries, merging equal (head, tail) neighbours
    int64_t nnz = 0;
    for (int64_t i = 0; i < n_entities; ++i) {
        rowptr[i] = (int32_t)nnz;
        int64_t prev_t = -1;
        for (int64_t k = start[i]; k < start[i + 1]; ++k) {
            int64_t e = order[k];
            rel[k] = r ? (int32_t)r[e] : 0;
            if (t[e] != prev_t) {
                col[nnz] = (int32_t)t[e];
                eptr[nnz] = (int32_t)k;
                ++nnz;
                prev_t = t[e];
            }
        }
    }
    rowptr[n_entities] = (int32_t)nnz;
    if (n_edges > 0 || eptr) {
        if (eptr) eptr[nnz] = (int32_t)n_edges;
    }
    *nnz_out = nnz;
    return LKG_OK;
}

extern "C" int lkg_csr_transpose(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr,
                                 const int32_t *col, int32_t *t_rowptr, int32_t *t_col,
                                 int32_t *t_perm) {
    LKG_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && nnz < INT32_MAX, "lkg_csr_transpose: bad sizes");
    LKG_REQUIRE(rowptr && t_rowptr && (nnz == 0 || (col && t_col && t_perm)), "lkg_csr_transpose: null pointer");
    LKG_REQUIRE(rowptr[n_rows] == nnz, "lkg_csr_transpose: rowptr[n_rows]=%d != nnz=%lld", rowptr[n_rows],
                (long long)nnz);
    std::vector<int32_t> pos((size_t)n_cols + 1, 0);
    for (int64_t j = 0; j < nnz; ++j) {
        if ((uint32_t)col[j] >= (uint64_t)n_cols) {
            lkg_set_error("lkg_csr_transpose: col[%lld]=%d outside [0,%lld)", (long long)j, col[j],
                          (long long)n_cols);
            return LKG_ERR_INVALID_ARG;
        }
        pos[col[j] + 1]++;
    }
    for (int64_t c = 0; c < n_cols; ++c) pos[c + 1] += pos[c];
    std::memcpy(t_rowptr, pos.data(), sizeof(int32_t) * ((size_t)n_cols + 1));
    for (int64_t i = 0; i < n_rows; ++i)
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            int32_t p = pos[col[j]]++;
            t_col[p] = (int32_t)i;
            t_perm[p] = j;
        }
    return LKG_OK;
}

extern "C" int lkg_row_partition(int64_t n_rows, const int32_t *rowptr, int32_t n_parts, int64_t *cuts) {
    LKG_REQUIRE(n_rows >= 0 && n_parts >= 1 && rowptr && cuts, "lkg_row_partition: bad arguments");
    int64_t nnz = rowptr[n_rows] - rowptr[0];
    cuts[0] = 0;
    for (int32_t p = 1; p < n_parts; ++p) {
        int64_t want = rowptr[0] + (nnz * p) / n_parts;
        const int32_t *it = std::lower_bound(rowptr, rowptr + n_rows + 1, (int32_t)want);
        int64_t row = it - rowptr;
        cuts[p] = std::max(cuts[p - 1], std::min<int64_t>(row, n_rows));
    }
    cuts[n_parts] = n_rows;
    return LKG_OK;
}
