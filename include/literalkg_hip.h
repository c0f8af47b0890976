/*
 * literalkg_hip.h -- C ABI of the MI355X (gfx950) LiteralKG hot-path library.
 *
 * The reference (NSLab-CUK/LiteralKG) is pure Python on PyTorch and has NO native
 * boundary of its own.  Its "plugin API" for this path is the
 * nn.Module surface LiteralKG.forward(*input, device=, mode=) (model.py:521-532);
 * literalkg_amd/model.py mirrors that surface and binds the entry points below
 * with ctypes.  Each entry point names the reference lines whose ATen call
 * sequence it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only, no torch types; every device function takes
 *     the hipStream_t to launch on (as void*) and is asynchronous on it;
 *   - no allocation, no host sync inside device functions (graph-capturable);
 *   - return 0 on success, a negative lkg_status otherwise; lkg_last_error()
 *     returns a thread-local message for the last failure on the calling thread;
 *   - all floating point is fp32, entity/column ids in the KG structure are int32,
 *     batch triple ids are int64 (the dtype the reference's DataLoader hands over);
 *   - "ld*" arguments are row strides in ELEMENTS so that callers can address
 *     column slices of a concatenated table without a copy.
 */
#ifndef LITERALKG_HIP_H
#define LITERALKG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LKG_OK = 0,
    LKG_ERR_INVALID_ARG = -1,
    LKG_ERR_HIP = -2,
    LKG_ERR_UNSUPPORTED = -3,
    LKG_ERR_NOMEM = -4
} lkg_status;

/* library version: major*10000 + minor*100 + patch */
int lkg_version(void);
const char *lkg_last_error(void);
/* Load every code object of the library on the CURRENT device now.  HIP loads a translation unit's device code on the first
 * launch of one of its kernels -- tens of milliseconds that would otherwise land inside the first call that needs them (the
 * reference's first mode='update_att', main_pretraining.py:139: a dozen sort / scan kernels of the structure build).  The
 * Python binding calls it once per process, before the first entry point that takes a stream.  No launch, no allocation. */
int lkg_preload(void);

/* ------------------------------------------------------------------ host --
 * KG structure build.  Replaces the per-relation torch.where / cat / stack /
 * sparse_coo_tensor / coalesce sequence of LiteralKG.update_attention
 * (model.py:451-468) and the scipy COO assembly of DataLoader
 * (dataloader.py:449-495): triples are sorted by (head, tail); duplicate
 * (head, tail) pairs under different relations are merged into ONE stored
 * entry (their logits are summed later, as coalesce() does).
 *
 * in : n_entities, n_edges, h/t/r int64[n_edges] (host; r may be NULL -> 0)
 * out: rowptr int32[n_entities+1], col int32[<=n_edges] (tails, ascending per row),
 *      eptr int32[<=n_edges+1] (entry j covers sorted raw edges eptr[j]..eptr[j+1]),
 *      rel int32[n_edges] (relation of each sorted raw edge),
 *      order int64[n_edges] (sorted raw edge k is input edge order[k]),
 *      *nnz_out = number of stored entries.
 * All out buffers are caller-allocated with the upper-bound sizes above.      */
int lkg_csr_build(int64_t n_entities, int64_t n_edges, const int64_t *h, const int64_t *t,
                  const int64_t *r, int32_t *rowptr, int32_t *col, int32_t *eptr, int32_t *rel,
                  int64_t *order, int64_t *nnz_out);

/* CSC of the same pattern, for the backward pass grad_ego = A^T grad_side
 * (autograd of model.py:106).  t_rowptr int32[n_cols+1], t_col int32[nnz]
 * (heads, ascending per tail), t_perm int32[nnz]: transposed entry k is CSR
 * entry t_perm[k].                                                          */
int lkg_csr_transpose(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr,
                      const int32_t *col, int32_t *t_rowptr, int32_t *t_col, int32_t *t_perm);

/* The same two builds on the DEVICE (lkg_csr_device.hip): h / t / r and every output are device pointers, outputs bit
 * for bit those of lkg_csr_build / lkg_csr_transpose (stable (head, tail) order, ties in input order; heads ascending
 * inside every tail of the CSC).  A hand-written LSD radix sort (8-bit digits) over the key head * n_entities + tail;
 * no allocation: `workspace` holds at least lkg_csr_*_device_workspace(...) bytes.  order is int32[n_edges] here.
 * counts (device int64[2]): counts[0] = number of stored entries, counts[1] = number of triples with an id outside
 * [0, n_entities) or a relation outside int32 (the host build rejects such input; callers must treat counts[1] > 0 as
 * the same error -- the outputs are then unspecified but in bounds).  The caller reads counts after the stream has
 * reached this point (one host sync per edge list) to size col / eptr.                                              */
int64_t lkg_csr_build_device_workspace(int64_t n_entities, int64_t n_edges);
int lkg_csr_build_device(int64_t n_entities, int64_t n_edges, const int64_t *h, const int64_t *t, const int64_t *r,
                         int32_t *rowptr, int32_t *col, int32_t *eptr, int32_t *rel, int32_t *order,
                         int64_t *counts, void *workspace, int64_t workspace_bytes, void *stream);
int64_t lkg_csr_transpose_device_workspace(int64_t n_cols, int64_t nnz);
int lkg_csr_transpose_device(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                             int32_t *t_rowptr, int32_t *t_col, int32_t *t_perm, void *workspace,
                             int64_t workspace_bytes, void *stream);

/* indices_out int64[2 * nnz] (device) = the [2, nnz] index tensor of the coalesced sparse COO A_in the reference keeps
 * (model.py:462-468, 257-261): row 0 the head of every stored entry, row 1 its tail, in the structure's entry order.   */
int lkg_csr_coo_indices_i64(int64_t n_rows, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                            int64_t *indices_out, void *stream);

/* Cut [0, n_rows) into n_parts contiguous row ranges balanced by stored entries
 * (cuts only at row boundaries so a softmax row never straddles two GPUs,
 * SURVEY.md 8e).  cuts int64[n_parts+1].                                     */
int lkg_row_partition(int64_t n_rows, const int32_t *rowptr, int32_t n_parts, int64_t *cuts);

/* f3  graph ingestion.  Text file of "h r t" lines (single spaces), as
 * DataLoader.load_graph reads with pandas (dataloader.py:186-190).  Two calls: count, then read into
 * caller-sized int64 arrays.  A malformed line is an error (LKG_ERR_INVALID_ARG).              */
int lkg_triples_count(const char *path, int64_t *n_lines);
int lkg_triples_read(const char *path, int64_t capacity, int64_t *h, int64_t *r, int64_t *t,
                     int64_t *n_read);
/* drop_duplicates(keep='first') (dataloader.py:189): keep int64[<=n] lists, in input order, the first
 * occurrence of every distinct (h, r, t).                                                       */
int lkg_triples_dedup(int64_t n, const int64_t *h, const int64_t *r, const int64_t *t, int64_t *keep,
                      int64_t *n_keep);
/* Initial attention values of the loader, A_in = sum_r norm(A_r) (dataloader.py:449-495), written in
 * the entry order of the structure built by lkg_csr_build.  kind 0 = 'random-walk' (D_r^-1 A_r),
 * kind 1 = 'symmetric' (D_r^-1/2 A_r D_r^-1/2 with the ROW sums on both sides, as the reference).
 * All pointers are HOST memory.                                                                 */
int lkg_laplacian_f32(int64_t n_entities, int64_t n_raw, int64_t nnz, const int32_t *rowptr,
                      const int32_t *col, const int32_t *eptr, const int32_t *rel, int32_t kind,
                      float *val_out);
/* The same on the DEVICE, over the arrays lkg_csr_build_device left in HBM (all pointers device memory): n_rel = number of
 * relations (max relation id + 1), deg_workspace int32[n_entities * n_rel] scratch.  Same f64 arithmetic as the host form:
 * the values agree bit for bit.                                                                                      */
int lkg_laplacian_device_f32(int64_t n_entities, int64_t nnz, int32_t n_rel, const int32_t *rowptr, const int32_t *col,
                             const int32_t *eptr, const int32_t *rel, int32_t kind, int32_t *deg_workspace,
                             float *val_out, void *stream);

/* ---------------------------------------------------------------- device --
 * K3/K4  neighbour aggregation  out[i,:] = sum_{j in row i} val[j] * x[col[j],:]
 * Replaces torch.matmul(A_in, ego) (model.py:106); called with the CSC arrays
 * it is the backward A^T grad.  Rows without entries are written as zeros.
 * rowptr is int32[n_rows+1] holding offsets into col/val; x has >= max(col)+1 rows -- or is a SHIFTED view
 * (x - offset*ldx) that covers at least every col value stored in the entries rowptr[0]..rowptr[n_rows] of THIS call
 * (a row-range shard that holds only its own rows of x): the kernels never gather through an entry outside that range.
 * Rows without entries come out as exact zeros whatever x holds.  The grid is limited to 2^32 threads.
 * long_rows (nullable, device int32[n_long]): the rows (relative to rowptr) holding more than
 * long_thresh entries; each of them gets a whole workgroup instead of one wave so that a skewed
 * degree distribution does not leave one wave as the tail of the launch.  The list must contain
 * exactly the rows with > long_thresh entries (KGStructure builds it on the host).
 * self (nullable, n_rows x d, stride ld_self): out[i,:] = self[i,:] + sum ... , i.e. the
 * `ego + side` of the gcn / bi-interaction / gin layers (model.py:109, 123, 132) without a second
 * pass; in the backward the same flag adds the incoming gradient (d(ego+side)/d ego = I + A^T).
 * `self` may alias `out` (out += A x).  One kernel launch per call when d <= 128 or d is a multiple of 128
 * (128-column slabs in slab-major workgroup order), else one launch per 128-column slab; rows of at most 32 floats
 * take eight rows per wave.  Deterministic (no atomics): the same inputs give the same bits.          */
int lkg_spmm_csr_f32(int64_t n_rows, int32_t d, const int32_t *rowptr, const int32_t *col,
                     const float *val, const float *x, int64_t ldx, float *out, int64_t ldo,
                     const float *self, int64_t ld_self, const int32_t *long_rows, int32_t n_long,
                     int32_t long_thresh, void *stream);

/* The same with two optional row-wise extras in the epilogue (both nullable), so that an aggregation layer's
 * neighbours in the autograd graph cost no pass of their own:
 *   add2               out[i,:] += add2[i,:]  -- in the backward, the gradient that reaches `ego` through its OTHER use
 *                      (the concatenated table keeps a copy of the layer input, model.py:300-309), which autograd
 *                      would otherwise add in a separate N x D pass; ld_add2 = 0 adds the SAME row to every output
 *                      row: the bias of a gcn layer whose Linear was applied BEFORE the aggregation
 *                      ((ego + A ego) W^T + b = p + A p + b with p = ego W^T, model.py:108-111, when out_dim < in_dim);
 *   add2_rows          (nullable, uint8[n_rows]) add2 is read for the rows whose byte is non-zero only: the loss's
 *                      gradient of the concatenated table touches <= 3B rows (lkg_fill_rows_f32 keeps the flags);
 *   copy_src/copy_dst  copy_dst[i,:] = copy_src[i,:] -- in the forward, that copy itself (the raw entity table into
 *                      column slot 0 of the concatenated table when no gate is configured);
 *   rowmax_out         float[n_rows] (cleared here): max |out[i,:]| -- the row scale the tall GEMM of the layer's
 *                      Linear needs of its input (lkg_gemm_tall_f32), produced while the row is in registers;
 *   x_rows, self_rows  (nullable, uint8 per row of x / of self) a zero byte promises that the row is all zero: its
 *                      entries are skipped without touching x (16-byte path; otherwise ignored).  The backward of the
 *                      LAST aggregation layer: the loss's gradient reaches <= 3B of the N rows, so all but a fraction
 *                      of a percent of the transpose SpMM's gathers would fetch zeros;
 *   out_rows           (nullable, uint8[n_rows], cleared here; needs x_rows and the 16-byte path, d <= 1024) receives 1
 *                      for every row that got a contribution (a flagged entry, a flagged self / add2 row); the OTHER
 *                      rows of out are not written: out is a table the caller keeps all-zero there.  The set of
 *                      flagged rows is the gradient's next frontier, handed on to the layer below;
 *   rows_with_entries / rows_without_entries
 *                      (both nullable, int32 lists that together hold every row once; not with x_rows / out_rows) for a
 *                      structure whose rows are mostly EMPTY (the reference's data/Small: 766 k entity rows, 84 % never
 *                      a head): one wave per LISTED row with entries, the rows without entries in one streaming pass
 *                      (out = self + add2, copy, row maximum) -- one wave per empty row left the launch bound by the rate
 *                      at which workgroups start.                                                                  */
int lkg_spmm_csr_fused_f32(int64_t n_rows, int32_t d, const int32_t *rowptr, const int32_t *col,
                           const float *val, const float *x, int64_t ldx, float *out, int64_t ldo,
                           const float *self, int64_t ld_self, const float *add2, int64_t ld_add2,
                           const uint8_t *add2_rows, const float *copy_src, int64_t ld_copy_src, float *copy_dst,
                           int64_t ld_copy_dst, float *rowmax_out, const uint8_t *x_rows, const uint8_t *self_rows,
                           uint8_t *out_rows, const int32_t *long_rows, int32_t n_long, int32_t long_thresh,
                           const int32_t *rows_with_entries, int64_t n_rows_with_entries,
                           const int32_t *rows_without_entries, int64_t n_rows_without_entries, void *stream);

/* Batch-pruned step (exact; literalkg_amd/pruned.py): the loss reads <= 3B rows of the last layer, so a
 * layer only needs the rows its consumers read.  lkg_csr_extract_rows copies the entries of the (sorted,
 * int64) rows sel_rows into a compact CSR whose offsets out_rowptr int32[n_sel+1] the caller has already
 * computed (exclusive scan of the row lengths); col keeps the ORIGINAL ids.
 * lkg_spmm_csr_scatter_bwd_f32 is the backward of out = A_sub @ x for such a small sub-CSR without
 * building its transpose every step: g_x[col[j],:] += val[j] * g_out[row,:] (f32 atomics; g_x is
 * zero-initialised by the caller).                                                                */
int lkg_csr_extract_rows(int64_t n_sel, const int64_t *sel_rows, const int32_t *rowptr, const int32_t *col,
                         const float *val, const int32_t *out_rowptr, int32_t *out_col, float *out_val,
                         void *stream);
int lkg_spmm_csr_scatter_bwd_f32(int64_t n_rows, int32_t d, const int32_t *rowptr, const int32_t *col,
                                 const float *val, const float *g_out, int64_t ldg, float *g_x,
                                 int64_t ldx, void *stream);

/* K1+K2  attention refresh, fused: per stored entry
 *     v = sum over its raw edges e of  sum_d ent[t,d] * tanh(ent[h,d] + relemb[rel[e],d])
 * then softmax of v over the stored entries of each head row.
 * Replaces update_attention_batch + coalesce + torch.sparse.softmax(A.cpu(), dim=1)
 * (model.py:430-471).  eptr may be NULL when no (h,t) pair is duplicated
 * (then rel is indexed by entry, and rel_first / dup_* / nnz are ignored).  With eptr: rel_first
 * int32[nnz] holds rel[eptr[j]] per entry; dup_entries int32[n_dup] lists the entries covering more than
 * one raw edge and dup_rows their head rows (relative to row_offset); a pre-pass adds their further
 * relations' terms so that the main kernel's hot loop never chases eptr -> rel; [entry_lo, entry_hi) =
 * rowptr[0] .. rowptr[n_rows] (host-known) is the entry range this call refreshes: only those elements of
 * val_out are cleared and rewritten, the rest is left untouched.  logits_out (nullable) receives the merged
 * pre-softmax logits, val_out the attention values, both float[nnz].
 * row_offset: rowptr holds n_rows+1 offsets for head rows row_offset..row_offset+n_rows
 * (a row-range shard of a larger graph; ent always holds the full table).
 * long_rows / n_long / long_thresh: as for lkg_spmm_csr_f32 (one workgroup per long row).
 * n_rel: the rows of relemb (nn.Embedding(n_relations, relation_dim), model.py:276); every relation id in rel /
 * rel_first is below it (the caller checks).  A table of at most 16 KB is staged in LDS per workgroup;
 * 0 = unknown (the rows are fetched from global memory per entry).  */
int lkg_edge_softmax_f32(int64_t n_rows, int64_t row_offset, int32_t d, const int32_t *rowptr,
                         const int32_t *col, const int32_t *eptr, const int32_t *rel,
                         const int32_t *rel_first, const int32_t *dup_entries,
                         const int32_t *dup_rows, int32_t n_dup, int64_t entry_lo, int64_t entry_hi,
                         const float *ent, int64_t ld_ent, const float *relemb, int64_t ld_rel,
                         float *val_out, float *logits_out, const int32_t *long_rows, int32_t n_long,
                         int32_t long_thresh, int32_t n_rel, void *stream);

/* dst[i] = src[perm[i]] for i < n  (attention values into CSC order after a refresh, or into the entry order of a part
 * of the structure).  n_src = the extent of src: an index outside [0, n_src) is not dereferenced and NaN is stored for it,
 * so an index list that does not belong to src can never make the device read outside src.                       */
int lkg_permute_f32(int64_t n, const int32_t *perm, int64_t n_src, const float *src, float *dst, void *stream);

/* Structure check (a debugging / hardening aid; no counterpart in the reference, whose torch.sparse tensors validate
 * themselves): *bad_out (device int32, cleared here) receives the number of rows of the CSR view rowptr[0 .. n_rows] whose
 * offsets are not 0 <= rowptr[i] <= rowptr[i+1] <= nnz plus the number of entries of that view whose column id minus
 * col_offset lies outside [0, n_cols) -- i.e. 0 exactly when lkg_spmm_csr_f32 over (rowptr, col, val[nnz], x[n_cols rows]
 * handed over with that row offset) stays inside its operands.  literalkg_amd.ops.spmm_raw runs it before every launch
 * when LKG_CHECK_STRUCTURES=1 (the GPU test tier sets it).                                                          */
int lkg_csr_check_i32(int64_t n_rows, const int32_t *rowptr, int64_t nnz, const int32_t *col, int64_t col_offset,
                      int64_t n_cols, int32_t *bad_out, void *stream);

/* K8  TransE-form triple scoring (model_bce.py:329-368; same math baselines.py:33-61)
 *   pos_b = |e_h + r - e_p|^2, neg_b = |e_h + r - e_n|^2,
 *   reg_b = (|e_h|^2 + |r|^2 + |e_p|^2 + |e_n|^2)/2,  rank_b = -logsigmoid(neg_b - pos_b)
 * emb is the N x dim entity-side table (stride ld_emb), relemb the relation table.
 * Outputs are float[batch] each.                                               */
int lkg_transe_score_fwd_f32(int64_t batch, int32_t dim, const float *emb, int64_t ld_emb,
                             const float *relemb, int64_t ld_rel, const int64_t *h,
                             const int64_t *r, const int64_t *pos_t, const int64_t *neg_t,
                             float *pos, float *neg, float *reg, float *rank, void *stream);

/* loss = mean(rank) + lambda * mean(reg); one float written to loss_out.
 * Replaces torch.mean / _L2_loss_mean / add (model.py:8-9, 420-426).           */
int lkg_loss_reduce_f32(int64_t batch, const float *rank, const float *reg, float lambda,
                        float *loss_out, void *stream);

/* Backward of the two calls above.  g_loss is a device pointer to the upstream
 * scalar gradient.  Accumulates (atomic add) into g_emb (N x dim, stride ld_gemb)
 * and g_rel; both must be zero-initialised (or hold gradients to add to).       */
int lkg_transe_score_bwd_f32(int64_t batch, int32_t dim, const float *emb, int64_t ld_emb,
                             const float *relemb, int64_t ld_rel, const int64_t *h,
                             const int64_t *r, const int64_t *pos_t, const int64_t *neg_t,
                             const float *pos, const float *neg, float lambda,
                             const float *g_loss, float *g_emb, int64_t ld_gemb, float *g_rel,
                             int64_t ld_grel, void *stream);

/* K7  TransR projection W_r = gat_trans_M[r] (model.py:372, 390-395) without ever materialising the
 * B x C x D gather: the batch is grouped by relation (stable counting sort), its entity rows are
 * gathered in that order, and ONE grouped MFMA GEMM applies W_r per relation segment.
 *
 * lkg_group_by_key_i64: perm int32[n] lists input positions in key order (stable), seg int32[n_keys+1]
 * the first position of each key.  n_bad (nullable, device int32[1]) receives the number of keys outside
 * [0, n_keys): the reference's gat_trans_M[r] raises IndexError on such a batch, so callers must treat n_bad > 0 as
 * an error (ops.py checks it without a host sync, one call later); the kernel itself groups such a key with the
 * nearest valid one only to keep the launches queued behind it in bounds.                           */
int lkg_group_by_key_i64(int64_t n, int32_t n_keys, const int64_t *keys, int32_t *perm, int32_t *seg,
                         int32_t *n_bad, void *stream);
/* dst[i,:] = src[idx[perm[i]],:]   (idx and/or perm may be NULL = identity)                         */
int lkg_gather_rows_f32(int64_t n, int32_t d, const float *src, int64_t lds, const int64_t *idx,
                        const int32_t *perm, float *dst, int64_t ldd, void *stream);
/* dst[idx[perm[i]],:] += src[i,:]  (f32 atomics; autograd of the row gathers model.py:382-384)     */
int lkg_scatter_add_rows_f32(int64_t n, int32_t d, const float *src, int64_t lds, const int64_t *idx,
                             const int32_t *perm, float *dst, int64_t ldd, void *stream);
/* dst[idx[i],:] = value (dst nullable) and flags[idx[i]] = (flag != 0) (flags nullable, uint8 per table row): marks or
 * resets the <= 3B rows a loss gradient touches in an N-row table that is otherwise kept all-zero between steps, so that
 * no N x C fill runs per step (ops._RowScratch).  Negative ids are skipped (padding of de-duplicated lists).   */
int lkg_fill_rows_f32(int64_t n, int32_t d, const int64_t *idx, float *dst, int64_t ldd, float value, uint8_t *flags,
                      int32_t flag, void *stream);
/* Row-range forms for a table sharded by rows over the ranks of one node (literalkg_amd/distributed.py): this rank
 * holds rows [row_lo, row_hi) and src / dst point at row row_lo.
 *   gather : dst[i,:] = src[idx[i] - row_lo,:] when idx[i] is in the range, else 0 (the rows owned by other ranks are
 *            added by the all-reduce that follows);
 *   scatter: dst[idx[i] - row_lo,:] += src[i,:] for the idx[i] in the range (f32 atomics), the others are skipped.  */
int lkg_gather_rows_range_f32(int64_t n, int32_t d, const float *src, int64_t lds, const int64_t *idx,
                              int64_t row_lo, int64_t row_hi, float *dst, int64_t ldd, void *stream);
int lkg_scatter_add_rows_range_f32(int64_t n, int32_t d, const float *src, int64_t lds, const int64_t *idx,
                                   int64_t row_lo, int64_t row_hi, float *dst, int64_t ldd, void *stream);
/* dst[i] = src[perm[i]] for int64 ids                                                              */
int lkg_gather_i64(int64_t n, const int64_t *src, const int32_t *perm, int64_t *dst, void *stream);

/* f2  device-side batch / negative sampler with the semantics of
 * DataLoader.generate_kg_batch (dataloader.py:249-330): for each of the n_groups heads one positive
 * (relation, tail) drawn uniformly from the head's triples and neg_rate negative tails drawn like
 * random.choice(training_tails), rejecting (tail, relation) positives of the head and repeats inside the
 * group; h / r / pos_t are repeated neg_rate times.  Outputs are int64[n_groups * neg_rate].  A head outside
 * [0, n_entities) or without a triple (kg_dict[h] raises KeyError in the reference) yields a sentinel group:
 * r = pos_t = neg_t = -1, nothing is drawn.  Counter-based RNG: the batch is a pure function of (seed, heads).  */
int lkg_sample_kg_batch(int64_t n_groups, int32_t neg_rate, uint64_t seed, const int64_t *heads,
                        int64_t n_entities, const int32_t *rowptr, const int32_t *col, const int32_t *eptr, const int32_t *rel,
                        int64_t nnz, int64_t n_raw, int64_t *out_h, int64_t *out_r, int64_t *out_pos_t,
                        int64_t *out_neg_t, void *stream);

/* Grouped fp32 MFMA GEMM over segments seg[g]..seg[g+1] (device int32[n_groups+1], no host sync):
 *  mode 1 (rows): C[seg rows,:] = alpha * A[seg rows,:] opB(B + (g % b_period)*stride_b) + beta*C   (A row-major)
 *                 max_seg_len bounds the longest segment (e.g. the batch size); b_period (0: = n_groups) lets several
 *                 row ranges share a B block: the head, positive-tail and negative-tail rows of a TransR batch, each
 *                 sorted by relation, are 3 R groups over the R matrices of gat_trans_M -- one launch.
 *  mode 2 (k)   : (C + g*stride_c)[m,n] = alpha * sum_{k in seg} A[k,m] B[k,n] + beta*C
 *                 (A given transposed, trans_a = 1; B k-major, trans_b = 0): g_W[r] = X_r^T G_r.   */
int lkg_grouped_gemm_f32(int32_t mode, int32_t n_groups, const int32_t *seg, int64_t max_seg_len,
                         int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, float alpha,
                         const float *a, int64_t lda, const float *b, int64_t ldb, int64_t stride_b,
                         int32_t b_period, float beta, float *c, int64_t ldc, int64_t stride_c, void *stream);

/* f1  fine-tuning head (model.py:316-348): dot-product BPR on table rows
 *   pos_b = <e_h, e_p>, neg_b = <e_h, e_n>, reg_b = (|e_h|^2 + |e_p|^2 + |e_n|^2)/2,
 *   rank_b = -logsigmoid(pos_b - neg_b);  loss via lkg_loss_reduce_f32 (lambda = fine_tuning_l2loss_lambda).
 * Backward accumulates (atomic) into g_emb (zero-initialised by the caller).          */
int lkg_dot_score_fwd_f32(int64_t batch, int32_t dim, const float *emb, int64_t ld_emb, const int64_t *h,
                          const int64_t *pos_t, const int64_t *neg_t, float *pos, float *neg, float *reg,
                          float *rank, void *stream);
int lkg_dot_score_bwd_f32(int64_t batch, int32_t dim, const float *emb, int64_t ld_emb, const int64_t *h,
                          const int64_t *pos_t, const int64_t *neg_t, const float *pos, const float *neg,
                          float lambda, const float *g_loss, float *g_emb, int64_t ld_gemb, void *stream);

/* f1  MLP head (model.py:499-519; model_bce.py:255-260, 423-436): y = BatchNorm1d(relu(z)) fused.
 * training != 0: batch statistics (needs n > 1), running_mean / running_var updated with `momentum` (unbiased
 * variance), like nn.BatchNorm1d; training == 0: the running buffers normalise.  save_mean / save_invstd
 * float[d] feed the backward, which writes g_z and OVERWRITES g_gamma / g_beta.                          */
int lkg_relu_batchnorm_fwd_f32(int64_t n, int32_t d, const float *z, int64_t ldz, const float *gamma,
                               const float *beta, float eps, int32_t training, float momentum,
                               float *running_mean, float *running_var, float *y, int64_t ldy,
                               float *save_mean, float *save_invstd, void *stream);
int lkg_relu_batchnorm_bwd_f32(int64_t n, int32_t d, const float *z, int64_t ldz, const float *gamma,
                               const float *save_mean, const float *save_invstd, int32_t training,
                               const float *g_y, int64_t ldgy, float *g_z, int64_t ldgz, float *g_gamma,
                               float *g_beta, void *stream);

/* Scores on already-projected rows (TransR form, model.py:413-426): same outputs
 * as lkg_transe_score_fwd_f32 but ph/pp/pn are dense matrices.
 * rows_per_group = K >= 1: the batch is batch/K groups of K consecutive rows that share (h, r, t+) -- the layout
 * DataLoader.generate_kg_batch produces (dataloader.py:318-330, each sampled head repeated neg_rate times).  Then
 * ph / pp / r (and g_ph / g_pp in the backward) hold ONE row per group (batch/K rows), pn / g_pn one row per triple:
 * the head and the positive tail are projected once per group instead of K times (2(2+K)/K B C D flops instead of
 * the reference's 6 B C D).  K = 1 is the general batch.  pos / neg / reg / rank are per triple either way.    */
int lkg_dense_score_fwd_f32(int64_t batch, int32_t rows_per_group, int32_t dim, const float *ph, const float *pp,
                            const float *pn, int64_t ld, const float *relemb, int64_t ld_rel,
                            const int64_t *r, float *pos, float *neg, float *reg, float *rank,
                            void *stream);
int lkg_dense_score_bwd_f32(int64_t batch, int32_t rows_per_group, int32_t dim, const float *ph, const float *pp,
                            const float *pn, int64_t ld, const float *relemb, int64_t ld_rel,
                            const int64_t *r, const float *pos, const float *neg, float lambda,
                            const float *g_loss, float *g_ph, float *g_pp, float *g_pn,
                            int64_t ldg, float *g_rel, int64_t ld_grel, void *stream);
/* Helpers of the grouped form.  lkg_check_grouped_i64: *n_bad (device int32) = number of rows that differ from the
 * first row of their group of rows_per_group in h, r or pos_t (0 = the batch has the layout above).
 * lkg_expand_groups_i32: from the key order of the GROUPS (perm / seg of lkg_group_by_key_i64 over one key per group)
 * the row order of the batch: perm_out[p*K + j] = perm[p]*K + j (int32[n_groups*K]), seg_out[s] = seg[s]*K
 * (int32[n_seg]).                                                                                                 */
int lkg_check_grouped_i64(int64_t n, int32_t rows_per_group, const int64_t *h, const int64_t *r,
                          const int64_t *pos_t, int32_t *n_bad, void *stream);
int lkg_expand_groups_i32(int64_t n_groups, int32_t rows_per_group, int32_t n_seg, const int32_t *perm,
                          const int32_t *seg, int32_t *perm_out, int32_t *seg_out, void *stream);

/* Caller-supplied row ids (the batch's h / pos_t / neg_t, model.py:366-372; head / tail ids of the scoring heads,
 * model.py:473-477) before any kernel gathers or scatters through them: out[i] = ids[i] when lo <= ids[i] < hi, else
 * lo (a valid row), and *n_bad (device int32, cleared here) = the number of ids outside the range.  The reference
 * raises IndexError / a device-side assert for such an id (its embedding lookups are bounds-checked by ATen); here
 * the kernels only ever see in-range ids and the host raises once it has read the counter (one call late, no sync
 * inside the step).                                                                                              */
int lkg_sanitize_ids_i64(int64_t n, const int64_t *ids, int64_t lo, int64_t hi, int64_t *out, int32_t *n_bad,
                         void *stream);

/* K5  row-wise epilogue of an aggregation layer (model.py:111, 161, 305):
 *   a   = leaky_relu(z, slope)                    (z = Linear output, n x d)
 *   y   = layer_norm(a) * gamma + beta            (eps)
 *   yn  = y / max(|y|_2, norm_eps)                (nullable: the L2-normalised copy)
 * save_mean / save_rstd float[n] are kept for the backward.
 * drop_p > 0: message dropout (nn.Dropout, model.py:28/161) is applied to y BEFORE the normalised
 *   copy is taken, y *= keep(seed, row*d + col) / (1 - drop_p) with a counter-based mask that the
 *   backward regenerates from the same seed.
 * y may be NULL (with yn given): the last layer's un-normalised output is read by nobody.   */
int lkg_act_layernorm_fwd_f32(int64_t n, int32_t d, const float *z, int64_t ldz, float slope,
                              const float *gamma, const float *beta, float eps, float *y,
                              int64_t ldy, float *yn, int64_t ldyn, float norm_eps,
                              float *save_mean, float *save_rstd, float drop_p, uint64_t seed,
                              void *stream);

/* Backward of the epilogue.  y may be NULL when the forward did not keep it (the last layer's output is read by
 * nobody: lkg_act_layernorm_fwd_f32 accepts y == NULL when yn is given); it is then recomputed from z, the saved
 * statistics, gamma, beta and the dropout seed for the rows that need it.  g_y and g_yn (nullable) are the upstream gradients of
 * the two outputs; writes g_z (n x d) and ACCUMULATES g_gamma / g_beta (atomic,
 * zero-initialised by the caller).  g_z_rowmax (nullable, float[n]) receives max |g_z[i,:]|:
 * the row scale of the data-gradient GEMM that consumes g_z (lkg_gemm_tall_f32), for free.
 * g_yn_rows (nullable, uint8[n]): g_yn is known to be zero outside the rows whose byte is non-zero (the loss's
 * row-sparse gradient, lkg_fill_rows_f32) -- those rows skip the g_yn / y reads, and with g_y == NULL the whole
 * row (g_z = 0, no z read either).  sparse_out != 0 (needs g_yn_rows, g_y == NULL, g_z_rowmax == NULL): g_z is a
 * table the caller keeps all-zero outside the flagged rows, so the zero rows are not written either; row_ids
 * (nullable, int64[n_row_ids], with sparse_out) then lists the rows to visit (negative entries = padding): the launch
 * is over the <= 3B rows the loss reaches -- or over the gradient's frontier, when g_y is row-sparse too (its rows
 * must then be in the list; g_y is read for every listed row) -- instead of N.                              */
int lkg_act_layernorm_bwd_f32(int64_t n, int32_t d, const float *z, int64_t ldz, float slope,
                              const float *gamma, const float *beta, const float *y, int64_t ldy,
                              const float *save_mean, const float *save_rstd, const float *g_y,
                              int64_t ldgy, const float *g_yn, int64_t ldgyn, float norm_eps,
                              float *g_z, int64_t ldgz, float *g_gamma, float *g_beta,
                              float drop_p, uint64_t seed, float *g_z_rowmax, const uint8_t *g_yn_rows,
                              int32_t sparse_out, const int64_t *row_ids, int64_t n_row_ids, void *stream);

/* K6  literal-gate blend (gate.py:24-26, 47-49) on the two pre-activations
 *   out = (1 - sigmoid(zpre)) * x + sigmoid(zpre) * tanh(gpre)
 * gpre / zpre already hold the Linear outputs INCLUDING their biases.              */
int lkg_gate_blend_fwd_f32(int64_t n, int32_t d, const float *x, int64_t ldx, const float *gpre,
                           int64_t ldg, const float *zpre, int64_t ldz, float *out, int64_t ldo,
                           void *stream);
/* activated != 0: gpre / zpre hold tanh(g) / sigmoid(z) (what the fused gate epilogue of lkg_gemm_tall_f32 keeps).
 * g_pre_rowmax (nullable, float[n], cleared here): max over row i of |g_gpre[i,:]| and |g_zpre[i,:]| -- the row scale of
 * the two-panel data-gradient GEMM that consumes them (lkg_gemm_tall_f32), without a pass of its own.               */
int lkg_gate_blend_bwd_f32(int64_t n, int32_t d, const float *x, int64_t ldx, const float *gpre,
                           int64_t ldg, const float *zpre, int64_t ldz, const float *g_out,
                           int64_t ldgo, float *g_x, int64_t ldgx, float *g_gpre, int64_t ldgg,
                           float *g_zpre, int64_t ldgz, int32_t activated, float *g_pre_rowmax, void *stream);

/* lkg_gate_blend_bwd_f32 (activated form) with the column statistics of what it reads and writes riding along, so that
 * the bias and weight gradients that follow need no pass of their own over [g_gpre | g_zpre] (2 d wide) and x:
 *   stats[0, 2d)         column sums of [g_gpre | g_zpre]        = the bias gradients of g and gate_* (gate.py:24-25)
 *   stats[2d, 4d)        column maxima |.| of [g_gpre | g_zpre]  }  the column scales of lkg_gemm_wgrad_f32
 *   stats[4d, 5d)        column maxima |.| of x                  }
 *   stats[5d + j 2d, ..) sum_r [g_gpre | g_zpre][r, :] * w[r, j] = the weight gradient of a NARROW literal panel w
 *                        (n_w <= 4 columns, nullable: the numeric literals, gate.py:23)
 * d a multiple of 4, <= 1024, 16-byte aligned operands.  workspace: 1024 (5 + 2 n_w) d floats (per-workgroup partial
 * vectors folded by a second tiny launch: no float atomics, the statistics are deterministic).                   */
int lkg_gate_blend_bwd_stats_f32(int64_t n, int32_t d, const float *x, int64_t ldx, const float *gpre, int64_t ldg,
                                 const float *zpre, int64_t ldz, const float *g_out, int64_t ldgo, float *g_x,
                                 int64_t ldgx, float *g_gpre, int64_t ldgg, float *g_zpre, int64_t ldgz,
                                 int32_t activated, float *g_pre_rowmax, const float *w, int64_t ldw, int32_t n_w,
                                 float *workspace, int64_t workspace_floats, float *stats, void *stream);

/* Weight-gradient product on the fp16 matrix cores (lkg_gemm_wgrad.hip):  C[m, n] = sum over the k rows of
 * A[k, m] * B[k, n]  (both operands k-major: dW = dY^T X of nn.Linear with k = rows, model.py:93-149, gate.py:22-25),
 * f32 in / f32 out, C overwritten.  a_colmax / b_colmax (device float[m] / float[n]): max |A[:, j]| / max |B[:, j]| --
 * every column is scaled by the power of two that brings its maximum to [2^13, 2^14), split exactly into two fp16
 * halves and multiplied with three fp16 MFMAs per product (f32-accurate normwise; lkg_gemm_f32 uses six bf16 MFMAs and
 * needs no scales).  A bound that is too small by more than a factor 4 can overflow fp16: pass true maxima (emitted by
 * the operand's producer -- lkg_gate_blend_bwd_f32, lkg_row_absmax_f32 -- or lkg_col_absmax_f32 for constant tables). */
int lkg_gemm_wgrad_f32(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *a_colmax,
                       const float *b, int64_t ldb, const float *b_colmax, float *c, int64_t ldc, void *stream);
/* The same product without scales ("bf16 x 3", six bf16 MFMAs per product as in lkg_gemm_f32) on a 256 x 128 tile per
 * CU with three tiles of loads in flight and the split hidden behind the MFMAs: the long-k engine of every wide weight
 * gradient of the path.  lkg_gemm_longk_ok tells whether it takes a product (k >= 8192 in whole 16-row tiles, widths and row strides
 * multiples of 4 floats, 16-byte aligned operands); otherwise lkg_gemm_f32(trans_a = 1) serves it.  C is overwritten.  */
int lkg_gemm_longk_ok(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb);
int lkg_gemm_longk_f32(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb,
                       float *c, int64_t ldc, void *stream);
/* The same product for a NARROW A (m <= 64: dW = dY^T X of an aggregation layer whose conv_dim is 32 or 64 -- the
 * reference's default stacks eight layers of 32, argument_pretraining.py:54-58): exact f32 FMAs on the VALU over LDS-staged
 * 32-row tiles, every thread an (m / 16) x 4 block of a 64-column chunk of C, slices of k combined by f32 atomics -- the
 * matrix-core engines would spend a 256 x 128 tile on it.  lkg_gemm_smallm_ok: m <= 64, k >= 4096, widths and strides
 * multiples of 4 floats, 16-byte aligned operands.  C is overwritten.                                                  */
int lkg_gemm_smallm_ok(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb);
int lkg_gemm_smallm_f32(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb,
                        float *c, int64_t ldc, void *stream);

/* The dense backward of a NARROW aggregation layer in one launch: y = Dropout(LayerNorm(LeakyReLU(x W^T + b))) with its
 * L2-normalised copy yn, 32 columns in and out (the reference's default conv_dim: model.py:108-111, 161, 305 over
 * argument_pretraining.py:54-58).  Replaces, for such a layer, lkg_act_layernorm_bwd_f32 + lkg_gemm_skinny_f32 (g_x = g_z W) +
 * lkg_gemm_smallm_f32 (g_W = g_z^T x) + lkg_colsum_f32 (g_b): g_z stays in LDS, both products run as f32 MFMAs.
 * x, z: the Linear's input and output rows as the forward pass kept them; y: required where g_yn is given; g_y and/or g_yn:
 * the incoming gradients (either may be null); g_yn_rows: optional row flags of g_yn (0 = that row of g_yn is zero).
 * Outputs: g_x [n, 32]; g_w [32 x 32, contiguous], g_bias (may be null), g_gamma, g_beta: OVERWRITTEN with the sums, which are
 * added up in a fixed order (every workgroup's partial sums go through `workspace`, lkg_narrow_layer_bwd_workspace(n) floats).
 * lkg_narrow_layer_bwd_ok: n >= 4096, d_in = d_out = 32, row strides multiples of 4 floats, 16-byte aligned operands. */
int64_t lkg_narrow_layer_bwd_workspace(int64_t n);
int lkg_narrow_layer_bwd_ok(int64_t n, int32_t d_in, int32_t d_out, const float *x, int64_t ldx, const float *z, int64_t ldz,
                            const float *y, int64_t ldy, const float *g_y, int64_t ldgy, const float *g_yn, int64_t ldgyn);
int lkg_narrow_layer_bwd_f32(int64_t n, int32_t d_in, int32_t d_out, const float *x, int64_t ldx, const float *w, int64_t ldw,
                             const float *z, int64_t ldz, float slope, const float *gamma, const float *y, int64_t ldy,
                             const float *save_mean, const float *save_rstd, const float *g_y, int64_t ldgy,
                             const float *g_yn, int64_t ldgyn, float norm_eps, float drop_p, uint64_t seed,
                             const uint8_t *g_yn_rows, float *g_x, int64_t ldgx, float *g_w, float *g_bias,
                             float *g_gamma, float *g_beta, float *workspace, int64_t workspace_floats, void *stream);
/* Skinny products over many rows: C[m, n] = A[m, k] . op(B) (+ bias) (+ beta C) for k, n <= 64 (op(B) = B[k, n], or
 * B[n, k]^T with trans_b: an nn.Linear weight) -- the 32 x 32 Linears, data gradients and residual mixes of narrow aggregation
 * layers (model.py:93-130 at the reference's default conv_dim 32).  Exact f32 FMAs on the VALU at streaming rate: op(B) stays
 * in LDS, 64 rows of A per turn, one row x n / 4 columns per thread.  lkg_gemm_skinny_ok: m >= 4096, k and n <= 64 in
 * multiples of 4, rows of A and C 16-byte aligned.                                                                  */
int lkg_gemm_skinny_ok(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *c, int64_t ldc);
int lkg_gemm_skinny_f32(int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb,
                        int32_t trans_b, float beta, float *c, int64_t ldc, const float *bias, void *stream);
/* out[c] = max_r |x[r,c]|  (out is overwritten)                                                     */
int lkg_col_absmax_f32(int64_t n, int32_t d, const float *x, int64_t ldx, float *out, void *stream);

/* out[c] = sum_r x[r,c]  (bias gradients of nn.Linear; out is overwritten)                       */
int lkg_colsum_f32(int64_t n, int32_t d, const float *x, int64_t ldx, float *out, void *stream);
/* out_w[c, j] = sum_r x[r,c] * w[r,j] for a NARROW w (1 <= n_w <= 8 columns) and, when out_sum != NULL,
 * out_sum[c] = sum_r x[r,c] in the same pass: the weight gradient of a Linear's narrow input panel (gate.py:22-25: the
 * 2 numeric literals) together with its bias gradient -- one read of x instead of a long-k GEMM that would spend a
 * 128-wide tile on n_w columns, plus a column-sum pass.  Both outputs are overwritten (f32 atomics inside).     */
int lkg_colsum_weighted_f32(int64_t n, int32_t d, const float *x, int64_t ldx, const float *w, int64_t ldw,
                            int32_t n_w, float *out_sum, float *out_w, int64_t ld_out_w, void *stream);

/* bi-interaction's element-wise front (model.py:123-128) fused with the GCNII-style residual's mix (model.py:94):
 *   out_sum = c (ego + side) + alpha h0p,  out_prod = c (ego * side) + alpha h0p,  c = 1 - alpha
 * (h0p NULL: c = 1 and no h0p term: the plain sum and product) in one pass over ego / side / h0p, and its backward
 *   g_ego = c (g_sum + g_prod * side),  g_side = c (g_sum + g_prod * ego),  g_h0p = alpha (g_sum + g_prod)
 * in one pass (g_ego, g_side, g_h0p: dense n x d outputs; g_h0p only with has_h0).  Replaces four element-wise ops of
 * the reference per layer (and eight-plus autograd kernels behind them).                                       */
int lkg_bi_mix_fwd_f32(int64_t n, int32_t d, const float *ego, int64_t lde, const float *side, int64_t lds,
                       const float *h0p, int64_t ldh, float alpha, float *out_sum, int64_t ldos, float *out_prod,
                       int64_t ldop, void *stream);
int lkg_bi_mix_bwd_f32(int64_t n, int32_t d, const float *ego, int64_t lde, const float *side, int64_t lds,
                       const float *g_sum, int64_t ldgs, const float *g_prod, int64_t ldgp, int32_t has_h0, float alpha,
                       float *g_ego, float *g_side, float *g_h0p, void *stream);

/* Small element-wise steps of the layers (row-major n x d, row strides in elements):
 *   op 0  out = alpha * a + beta * b   (b NULL: alpha * a + beta)   GCNII residual mix, model.py:94-96; 'gin' sums
 *   op 1  out = a * b                                                ego * side of 'bi-interaction', model.py:127
 *   op 2  out = leaky(a, alpha) + leaky(b, alpha)  (b NULL: one term) model.py:125-130, 310
 *   op 3  out = a * (b > 0 ? 1 : alpha)                              LeakyReLU backward (a gradient, b pre-activation) */
int lkg_eltwise_f32(int32_t op, int64_t n, int32_t d, const float *a, int64_t lda, const float *b,
                    int64_t ldb, float alpha, float beta, float *out, int64_t ldo, void *stream);

/* f4  one fused Adam step over a dense contiguous tensor (the reference runs dense torch.optim.Adam over
 * the N x D entity table every step, main_pretraining.py:47,119): torch.optim.Adam's arithmetic,
 * `step` is the 1-based step count used for the bias corrections; weight_decay is the L2 form.          */
int lkg_adam_step_f32(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                      void *stream);

/* Dense fp32 GEMM on the matrix cores, row-major, f32 in / f32 out / f32 accumulation:
 *   C[m,n] = alpha * sum_k opA(A)[m,k] * opB(B)[k,n] + beta * C[m,n] (+ bias[n])
 * Engines (chosen from the shape, results agree to f32 rounding): the f32-input MFMA (v_mfma_f32_32x32x2_f32), and two
 * "bf16 x 3" engines that form the f32 product from six v_mfma_f32_32x32x16_bf16 over an exact three-way bf16 split
 * of every operand (A row-major with m >= 16384 and a small B -- Linear forward / data gradient; trans_a = 1,
 * trans_b = 0 with k >= 2048 -- weight gradients).  The first needs lkg_gemm_workspace(trans_a, m, n, k) bytes of
 * caller-provided device workspace for B's planes (0 = the engine does not apply to this shape); without it (NULL or
 * too small) the product runs on the f32-input MFMA.  No allocation inside.
 * LKG_GEMM_F32_ONLY=1 in the environment (read at the first call) keeps every product on the f32-input MFMA.
 * trans_a / trans_b: 0 = stored as written, 1 = stored transposed (A is k x m / B is n x k).
 * Used for nn.Linear forward (trans_b = 1), its data gradient and its weight
 * gradient (model.py:111 etc., gate.py:24-25, linear_gat model.py:309).          */
int64_t lkg_gemm_workspace(int32_t trans_a, int64_t m, int64_t n, int64_t k);

/* C[m,n] = opA(A)[m,k] opB(B)[k,n] with float64 accumulation, rounded to float32 once; SMALL products only (m n <= 2^24,
 * m n k <= 2^34).  The GCNII-style residual's weight fold W_lin (W')^T of the device path (model.py:95-98 evaluates
 * Linear(mixed @ W'); the device multiplies the two small matrices first and runs ONE N-row product): W' has near-equal
 * entries, and an fp32 fold's accumulated rounding is what ill-conditioned residual configurations' scores then see.   */
int lkg_gemm_f64acc_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, const float *a, int64_t lda,
                        const float *b, int64_t ldb, float *c, int64_t ldc, void *stream);
int lkg_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, float alpha,
                 const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                 int64_t ldc, const float *bias, void *workspace, int64_t workspace_bytes, void *stream);

/* Tall GEMM of the layers' dense part (lkg_gemm_tall.hip): m rows (entities), n <= a few hundred output columns,
 *   C[m, n] = epilogue( sum over K-panels p of  A_p[m, ka[p]] . B_p[n, ka[p]]^T )            f32 in, f32 out.
 * Arithmetic "f16 x 2": rows of A and of B are scaled by powers of two, every element split exactly into two fp16
 * (11 + 11 significant bits), three v_mfma_f32_32x32x16_f16 per 16 k into a main and a correction accumulator (f32),
 * unscaled exactly in the epilogue: within ~2x of an f32 GEMM's rounding error against f64, at 3/8 of the f32 MFMA's
 * instruction count on a 16x faster pipe.
 *   a / lda / ka       host arrays of n_panels (1..3) device pointers / row strides / widths: A's K-panels, possibly
 *                      from different arrays (nn.Linear over a concatenated input WITHOUT the concatenation:
 *                      gate.py:23 [x | num | txt], model.py:116 [ego | side]); any alignment, any width;
 *   a_rowmax           device float[m]: max_k |A[i, k]| over all panels (lkg_row_absmax_f32; accumulate = 1 folds a
 *                      further panel in).  It only has to BOUND the row (a stale larger value costs precision, a
 *                      smaller one overflows fp16): callers cache it for constant panels;
 *   b / ldb            host arrays of n_groups * n_panels device pointers: block (group, panel) of B.  trans_b = 1:
 *                      stored [rows][ka[p]] (an nn.Linear weight or a column slice of one), 0: stored [ka[p]][rows]
 *                      (the data gradient  gy . W).  n_groups = 1: rows = n.  Epilogue 1 (gate): n_groups = 2 row
 *                      groups of n / 2 (the g and the z projection of gate.py:22-25), bias = [b_g | b_z], and
 *                      C[m, n/2] = (1 - sigmoid(z)) gate_x + sigmoid(z) tanh(g)   (gate.py:26; gate_x IS the first K-panel, a[0] --
 *                      its values are kept on chip while they pass through the staging registers) with tanh(g) / sigmoid(z)
 *                      optionally kept in gate_g / gate_z for lkg_gate_blend_bwd_f32(activated = 1);
 *   epilogue 0         C = alpha * product + beta * C + bias[n];
 *   epilogue bits 8-15 the tiling, chosen per call: 0 = the library's default, else 1 + {0 "256x2", 1 "128x1", 2 "256x1",
 *                      3 "256x1w", 4 "ws"} (tests and tools run them side by side; nothing in the environment selects code).
 *                      "ws" is the wave-specialised form: one 8-wave workgroup per CU, four loader waves keep the raw A windows
 *                      of up to six k steps in flight through an LDS ring (across tile boundaries) and split them into fp16
 *                      planes, four compute waves of 64 x 128 stream B's planes, run the MFMAs and the epilogue;
 *   workspace          >= lkg_gemm_tall_workspace(n, n_panels, ka, epilogue) bytes (B's fp16 planes: no allocation here). */
int lkg_row_absmax_f32(int64_t n, int32_t d, const float *x, int64_t ldx, float *out, int32_t accumulate,
                       void *stream);
int64_t lkg_gemm_tall_workspace(int32_t n, int32_t n_panels, const int32_t *ka, int32_t epilogue);
int lkg_gemm_tall_f32(int64_t m, int32_t n, int32_t n_panels, const float *const *a, const int64_t *lda,
                      const int32_t *ka, const float *a_rowmax, int32_t n_groups, const float *const *b,
                      const int64_t *ldb, int32_t trans_b, float alpha, float beta, float *c, int64_t ldc,
                      const float *bias, int32_t epilogue, const float *gate_x, int64_t ld_x, float *gate_g,
                      int64_t ld_g, float *gate_z, int64_t ld_z, void *workspace, int64_t workspace_bytes,
                      void *stream);

/* K5 as surveyed ("fused epilogue around an MFMA GEMM"): an aggregation layer's whole dense part in ONE launch,
 *     z  = sum_p A_p W_p^T + bias                      (nn.Linear over a column-concatenated input, model.py:108-110, 116-123)
 *     y  = Dropout_p(LayerNorm(LeakyReLU_slope(z)))    (model.py:111, 161; eps, gamma, beta: nn.LayerNorm's)
 *     yn = y / max(|y|_2, norm_eps)                    (F.normalize, model.py:305)
 * for output widths n <= 256 and at most two K-panels: the 128 x 256 tile of the tall GEMM holds whole rows; its epilogue
 * sends them through LDS in slabs of 16 rows, row-major, and runs the row-wise kernel's own arithmetic on them (one wave per
 * row, the same lane <-> column map and reduction tree), so y, yn, mean and rstd are BIT-IDENTICAL to lkg_gemm_tall_f32 (its
 * default tiling) followed by lkg_act_layernorm_fwd_f32 whenever that kernel takes its 16-byte path (n % 4 == 0, aligned
 * rows), and equal to fp32 rounding otherwise.  z is NOT written (nor read back by a second kernel): one HBM pass fewer.
 * y and yn are nullable (not both); save_mean / save_rstd float[m] are what lkg_act_layernorm_bwd_f32 needs beside a
 * recomputed z.  The dropout mask is lkg_act_layernorm_fwd_f32's (same seed, same mask).  Arguments a .. a_rowmax, workspace:
 * as for lkg_gemm_tall_f32 (w: one [n, ka[p]] block per panel, nn.Linear layout).                                         */
int64_t lkg_linear_act_layernorm_workspace(int32_t n, int32_t n_panels, const int32_t *ka);
int lkg_linear_act_layernorm_fwd_f32(int64_t m, int32_t n, int32_t n_panels, const float *const *a, const int64_t *lda,
                                     const int32_t *ka, const float *a_rowmax, const float *const *w, const int64_t *ldw,
                                     const float *bias, float slope, const float *gamma, const float *beta, float eps,
                                     float *y, int64_t ldy, float *yn, int64_t ldyn, float norm_eps, float *save_mean,
                                     float *save_rstd, float drop_p, uint64_t seed, void *workspace,
                                     int64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LITERALKG_HIP_H */
