// Host-side KG structure build: (head, tail)-sorted CSR with merged duplicate pairs,
// its CSC transpose, nnz-balanced row-range cuts, and the text ingestion of "h r t" files
// with the loader's initial attention values.  Plain C++ (no device code).
//
// What it replaces in the reference: the per-relation torch.where / cat / stack /
// sparse COO assembly + coalesce() of LiteralKG.update_attention (model.py:451-470)
// and the scipy COO -> tensor assembly of DataLoader (dataloader.py:449-495).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
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

int lkg_internal_preload_spmm();
int lkg_internal_preload_attention();
int lkg_internal_preload_batch();
int lkg_internal_preload_csr_device();
int lkg_internal_preload_gemm();
int lkg_internal_preload_gemm_tall();
int lkg_internal_preload_gemm_wgrad();
int lkg_internal_preload_rowwise();
int lkg_internal_preload_score();
int lkg_internal_preload_layer();

extern "C" int lkg_preload(void) {
    const int failed = lkg_internal_preload_spmm() + lkg_internal_preload_attention() + lkg_internal_preload_batch() + lkg_internal_preload_csr_device() + lkg_internal_preload_gemm() + lkg_internal_preload_gemm_tall() + lkg_internal_preload_gemm_wgrad() + lkg_internal_preload_rowwise() + lkg_internal_preload_score() + lkg_internal_preload_layer();
    if (failed) {
        lkg_set_error("lkg_preload: %d of the library's code objects could not be loaded on the current device", failed);
        return LKG_ERR_HIP;
    }
    return LKG_OK;
}

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
    if (eptr) eptr[nnz] = (int32_t)n_edges;
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

// ---------------------------------------------------------------------------------------------------
// f3  graph ingestion: the reference parses "h r t" text with pandas (engine='python'),
// drops duplicate rows and walks the frame with iterrows() (dataloader.py:186-190, 369-424), then builds
// per-relation scipy matrices for the initial A_in (dataloader.py:449-495).  Here: one pass over the file
// with a hand-rolled integer parser, an index sort for the duplicate drop, and the Laplacian values
// computed directly on the (head, tail)-sorted structure.
namespace {
struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() {
        if (p && n) munmap(const_cast<char *>(p), n);
        if (fd >= 0) close(fd);
    }
};

int map_file(const char *path, Mapped &m) {
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) {
        lkg_set_error("cannot open %s: %s", path, strerror(errno));
        return LKG_ERR_INVALID_ARG;
    }
    struct stat st;
    if (fstat(m.fd, &st) != 0) {
        lkg_set_error("cannot stat %s: %s", path, strerror(errno));
        return LKG_ERR_INVALID_ARG;
    }
    m.n = (size_t)st.st_size;
    if (m.n == 0) return LKG_OK;
    void *a = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (a == MAP_FAILED) {
        m.n = 0;
        lkg_set_error("cannot mmap %s: %s", path, strerror(errno));
        return LKG_ERR_NOMEM;
    }
    m.p = (const char *)a;
    return LKG_OK;
}

// one line "h r t" (single spaces / tabs between fields, optional trailing \r); returns false on a blank line
inline bool parse_line(const char *&c, const char *end, int64_t out[3], bool &bad) {
    while (c < end && (*c == ' ' || *c == '\t' || *c == '\r')) ++c;
    if (c >= end || *c == '\n') {
        if (c < end) ++c;
        return false;
    }
    for (int f = 0; f < 3; ++f) {
        while (c < end && (*c == ' ' || *c == '\t')) ++c;
        bool neg = false;
        if (c < end && *c == '-') {
            neg = true;
            ++c;
        }
        if (c >= end || *c < '0' || *c > '9') {
            bad = true;
            return false;
        }
        int64_t v = 0;
        while (c < end && *c >= '0' && *c <= '9') v = v * 10 + (*c++ - '0');
        out[f] = neg ? -v : v;
    }
    while (c < end && (*c == ' ' || *c == '\t' || *c == '\r')) ++c;
    if (c < end && *c != '\n') {
        bad = true;
        return false;
    }
    if (c < end) ++c;
    return true;
}
}  // namespace

namespace {
// The file cut into byte ranges that start at line starts: range k = [cut[k], cut[k+1]).
std::vector<size_t> line_chunks(const Mapped &m, int n_chunks) {
    std::vector<size_t> cut((size_t)n_chunks + 1, m.n);
    cut[0] = 0;
    for (int k = 1; k < n_chunks; ++k) {
        size_t p = std::max(cut[k - 1], m.n / n_chunks * (size_t)k);
        while (p < m.n && m.p[p] != '\n') ++p;
        cut[k] = std::min(m.n, p + 1);
    }
    return cut;
}

int parser_threads(size_t bytes) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::max<size_t>(1, std::min<size_t>(std::min(hw, 32u), bytes / (4u << 20)));   // >= 4 MB of text per thread
}

// Parse the lines of [lo, hi) (lo is a line start); store them from slot `at` on when h is given.  Returns the number of
// triples, or -1 - (byte offset of the malformed line).
int64_t parse_range(const Mapped &m, size_t lo, size_t hi, int64_t *h, int64_t *r, int64_t *t, int64_t at) {
    const char *c = m.p + lo, *end = m.p + hi;
    int64_t n = 0, v[3];
    bool bad = false;
    while (c < end) {
        if (parse_line(c, end, v, bad)) {
            if (h) {
                h[at + n] = v[0];
                r[at + n] = v[1];
                t[at + n] = v[2];
            }
            ++n;
        }
        if (bad) return -1 - (int64_t)(c - m.p);
    }
    return n;
}

// every chunk parsed by its own thread; counts[k] = triples of chunk k (negative: malformed)
void parse_chunks(const Mapped &m, const std::vector<size_t> &cut, std::vector<int64_t> &counts, int64_t *h, int64_t *r,
                  int64_t *t, const std::vector<int64_t> *offsets) {
    const int nc = (int)cut.size() - 1;
    counts.assign((size_t)nc, 0);
    std::vector<std::thread> th;
    for (int k = 0; k < nc; ++k)
        th.emplace_back([&, k] { counts[k] = parse_range(m, cut[k], cut[k + 1], h, r, t, offsets ? (*offsets)[k] : 0); });
    for (auto &x : th) x.join();
}
}  // namespace

// (the text is parsed by up to 32 threads over line-aligned byte ranges of the mmap'd file: 100 M triples / 1.8 GB in
// well under a second per pass on the GPU box's host instead of the ~4 s of one thread)
extern "C" int lkg_triples_count(const char *path, int64_t *n_lines) {
    LKG_REQUIRE(path && n_lines, "lkg_triples_count: null pointer");
    Mapped m;
    int rc = map_file(path, m);
    if (rc != LKG_OK) return rc;
    const auto cut = line_chunks(m, parser_threads(m.n));
    std::vector<int64_t> counts;
    parse_chunks(m, cut, counts, nullptr, nullptr, nullptr, nullptr);
    int64_t n = 0;
    for (int64_t c : counts) {
        if (c < 0) {
            lkg_set_error("%s: malformed line near byte %lld (expected 'h r t')", path, (long long)(-1 - c));
            return LKG_ERR_INVALID_ARG;
        }
        n += c;
    }
    *n_lines = n;
    return LKG_OK;
}

extern "C" int lkg_triples_read(const char *path, int64_t capacity, int64_t *h, int64_t *r, int64_t *t,
                                int64_t *n_read) {
    LKG_REQUIRE(path && n_read && (capacity == 0 || (h && r && t)), "lkg_triples_read: null pointer");
    Mapped m;
    int rc = map_file(path, m);
    if (rc != LKG_OK) return rc;
    const auto cut = line_chunks(m, parser_threads(m.n));
    std::vector<int64_t> counts, offsets(cut.size() - 1, 0);
    parse_chunks(m, cut, counts, nullptr, nullptr, nullptr, nullptr);          // pass 1: where every chunk's triples go
    int64_t n = 0;
    for (size_t k = 0; k < counts.size(); ++k) {
        if (counts[k] < 0) {
            lkg_set_error("%s: malformed line near byte %lld (expected 'h r t')", path, (long long)(-1 - counts[k]));
            return LKG_ERR_INVALID_ARG;
        }
        offsets[k] = n;
        n += counts[k];
    }
    if (n > capacity) {
        lkg_set_error("%s holds more than the %lld triples the caller sized for", path, (long long)capacity);
        return LKG_ERR_INVALID_ARG;
    }
    parse_chunks(m, cut, counts, h, r, t, &offsets);                            // pass 2: file order kept
    *n_read = n;
    return LKG_OK;
}

// drop_duplicates(keep='first') of dataloader.py:189: keep[] lists, in input order, the first occurrence of
// every distinct (h, r, t).  Large inputs: the triples are hashed into one bucket per thread, every bucket is sorted by
// value (records side by side: no indirection) and flags the first occurrence of each of its distinct triples; the flags are
// swept once.  100 M triples: ~3 s on the GPU box's host instead of ~25 s for one std::sort of an index list.
extern "C" int lkg_triples_dedup(int64_t n, const int64_t *h, const int64_t *r, const int64_t *t, int64_t *keep,
                                 int64_t *n_keep) {
    LKG_REQUIRE(n >= 0 && n_keep && (n == 0 || (h && r && t && keep)), "lkg_triples_dedup: bad arguments");
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int nt = (int)std::min<int64_t>(std::min(hw, 32u), n / (1 << 18));
    if (nt <= 1) {
        std::vector<int64_t> idx((size_t)n);
        for (int64_t i = 0; i < n; ++i) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {
            if (h[a] != h[b]) return h[a] < h[b];
            if (r[a] != r[b]) return r[a] < r[b];
            if (t[a] != t[b]) return t[a] < t[b];
            return a < b;
        });
        int64_t m = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t a = idx[i];
            if (i == 0 || h[a] != h[idx[i - 1]] || r[a] != r[idx[i - 1]] || t[a] != t[idx[i - 1]]) keep[m++] = a;
        }
        std::sort(keep, keep + m);
        *n_keep = m;
        return LKG_OK;
    }
    struct Rec {
        int64_t h, r, t, i;
    };
    auto bucket_of = [nt](int64_t a, int64_t b, int64_t c) {
        uint64_t x = (uint64_t)a * 0x9E3779B97F4A7C15ull ^ ((uint64_t)b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full ^
                     (uint64_t)c * 0x165667B19E3779F9ull;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        return (int)(x % (uint64_t)nt);
    };
    std::vector<uint8_t> first;
    std::vector<std::vector<std::vector<Rec>>> part((size_t)nt, std::vector<std::vector<Rec>>((size_t)nt));
    std::atomic<bool> oom{false};             // (an exception must not leave a thread)
    try {
        first.assign((size_t)n, 0);
        std::vector<std::thread> th;
        for (int s = 0; s < nt; ++s)
            th.emplace_back([&, s] {
                try {
                    const int64_t lo = n * s / nt, hi = n * (s + 1) / nt;
                    for (auto &v : part[s]) v.reserve((size_t)((hi - lo) / nt + (hi - lo) / (8 * nt) + 16));
                    for (int64_t i = lo; i < hi; ++i) part[s][bucket_of(h[i], r[i], t[i])].push_back(Rec{h[i], r[i], t[i], i});
                } catch (const std::bad_alloc &) {
                    oom = true;
                }
            });
        for (auto &x : th) x.join();
        th.clear();
        if (oom) throw std::bad_alloc();
        for (int b = 0; b < nt; ++b)
            th.emplace_back([&, b] {
              try {
                size_t total = 0;
                for (int s = 0; s < nt; ++s) total += part[s][b].size();
                std::vector<Rec> v;
                v.reserve(total);
                for (int s = 0; s < nt; ++s) {
                    v.insert(v.end(), part[s][b].begin(), part[s][b].end());
                    std::vector<Rec>().swap(part[s][b]);
                }
                std::sort(v.begin(), v.end(), [](const Rec &x, const Rec &y) {
                    if (x.h != y.h) return x.h < y.h;
                    if (x.r != y.r) return x.r < y.r;
                    if (x.t != y.t) return x.t < y.t;
                    return x.i < y.i;
                });
                for (size_t k = 0; k < v.size(); ++k)
                    if (k == 0 || v[k].h != v[k - 1].h || v[k].r != v[k - 1].r || v[k].t != v[k - 1].t) first[(size_t)v[k].i] = 1;
              } catch (const std::bad_alloc &) {
                oom = true;
              }
            });
        for (auto &x : th) x.join();
        if (oom) throw std::bad_alloc();
    } catch (const std::bad_alloc &) {
        lkg_set_error("lkg_triples_dedup: out of host memory for %lld triples", (long long)n);
        return LKG_ERR_NOMEM;
    }
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        if (first[(size_t)i]) keep[m++] = i;
    *n_keep = m;
    return LKG_OK;
}

// Initial attention values  A_in = sum_r norm(A_r)  on the (head, tail)-sorted structure
// (dataloader.py:449-495).  A_r is the binary adjacency of relation r; with d_r(x) = its ROW sum at x:
//   kind 0 "random-walk":  D_r^-1 A_r            -> entry (h,t) += 1 / d_r(h)
//   kind 1 "symmetric"  :  D_r^-1/2 A_r D_r^-1/2 -> entry (h,t) += d_r(h)^-1/2 * d_r(t)^-1/2  (0 when d_r(t) = 0;
//                          the reference uses the row-sum vector on both sides)
// Duplicate (h,r,t) raw edges must have been dropped (lkg_triples_dedup), as the loader does.
extern "C" int lkg_laplacian_f32(int64_t n_entities, int64_t n_raw, int64_t nnz, const int32_t *rowptr,
                                 const int32_t *col, const int32_t *eptr, const int32_t *rel, int32_t kind,
                                 float *val_out) {
    LKG_REQUIRE(n_entities >= 0 && n_raw >= 0 && nnz >= 0 && (kind == 0 || kind == 1), "lkg_laplacian_f32: bad arguments");
    if (nnz == 0) return LKG_OK;
    LKG_REQUIRE(rowptr && col && rel && val_out, "lkg_laplacian_f32: null pointer");
    int32_t n_rel = 0;
    for (int64_t k = 0; k < n_raw; ++k) n_rel = std::max(n_rel, rel[k] + 1);
    // out-degree of every (entity, relation) pair
    std::vector<int32_t> deg;
    try {
        deg.assign((size_t)n_entities * n_rel, 0);
    } catch (...) {
        lkg_set_error("lkg_laplacian_f32: out of host memory for %lld x %d degree table", (long long)n_entities, n_rel);
        return LKG_ERR_NOMEM;
    }
    for (int64_t i = 0; i < n_entities; ++i)
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const int32_t e0 = eptr ? eptr[j] : j, e1 = eptr ? eptr[j + 1] : j + 1;
            for (int32_t e = e0; e < e1; ++e) deg[(size_t)i * n_rel + rel[e]]++;
        }
    parallel_rows(n_entities, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i)
            for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
                const int32_t e0 = eptr ? eptr[j] : j, e1 = eptr ? eptr[j + 1] : j + 1;
                double acc = 0.0;
                for (int32_t e = e0; e < e1; ++e) {
                    const double dh = deg[(size_t)i * n_rel + rel[e]];
                    if (kind == 0) {
                        acc += 1.0 / dh;
                    } else {
                        const double dt = deg[(size_t)col[j] * n_rel + rel[e]];
                        if (dt > 0) acc += 1.0 / std::sqrt(dh) / std::sqrt(dt);
                    }
                }
                val_out[j] = (float)acc;
            }
    });
    return LKG_OK;
}
