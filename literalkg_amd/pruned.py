"""Batch-pruned evaluation of the encoder (exact, opt-in: ``model.prune_to_batch = True``).

The reference recomputes all N rows of every layer each step (model.py:380 -> 298-314) although the loss
reads only the <= 3B batch rows of ``gat_embed`` (model.py:382-384).  Row i of layer k depends on row i and
on the out-neighbours of i in layer k-1, so it suffices to evaluate

    R_L = unique(batch ids),   R_{k-1} = R_k  U  tails(R_k)          (k = L .. 1)

i.e. layer k on the rows R_k only, reading the compact output of layer k-1 on R_{k-1}.  Everything runs on
the SAME kernels as the dense path, on compact tensors: a per-step sub-CSR (rows R_k, columns relabelled to
positions in R_{k-1}) feeds ``lkg_spmm_csr_f32``; its backward is a scatter (``lkg_spmm_csr_scatter_bwd_f32``)
so no transpose is built per step.  Loss, scores and every parameter gradient equal the dense path's up to
fp32 summation order (tests/test_gpu_parity.py::test_pruned_*); only the dropout mask indexes compact rows.

Index bookkeeping (unique / searchsorted / cumsum over a few thousand ids) uses torch device ops.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch.autograd import Function

from . import _native as N
from . import ops
from .graph import KGStructure


class SubLayer:
    """Rows R_k of one layer as a compact CSR over the positions of R_{k-1}."""

    def __init__(self, rowptr, col, val, n_rows, n_src, self_pos):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.n_rows, self.n_src = n_rows, n_src
        self.self_pos = self_pos          # int64[n_rows]: position of each row of R_k inside R_{k-1}


class BatchSubgraph:
    def __init__(self, rows: List[torch.Tensor], layers: List[Optional[SubLayer]]):
        self.rows = rows                  # rows[k] = sorted int64 ids R_k, k = 0..L
        self.layers = layers              # layers[k] for k = 1..L (layers[0] is None)

    def positions(self, ids: torch.Tensor, level: int = -1) -> torch.Tensor:
        """Positions of entity ids inside R_level (ids must belong to it)."""
        return torch.searchsorted(self.rows[level], ids.long())

    def rows_in(self, k: int, base: int) -> torch.Tensor:
        """Positions of R_k inside R_base (base <= k, so R_k is a subset)."""
        return torch.searchsorted(self.rows[base], self.rows[k])


@torch.no_grad()
def build_batch_subgraph(graph: KGStructure, val: torch.Tensor, ids: torch.Tensor, n_layers: int,
                         max_rows: Optional[int] = None) -> Optional[BatchSubgraph]:
    """max_rows: give up (None) as soon as a level's row set outgrows it -- the caller takes the dense path, and a deep model
    does not pay for eight levels of bookkeeping to find that out (the reference's default architecture: two levels instead)."""
    ops._need_gpu(ids, val)
    rows = [None] * (n_layers + 1)
    layers: List[Optional[SubLayer]] = [None] * (n_layers + 1)
    rows[n_layers] = torch.unique(ids.long())
    rp = graph.rowptr
    for k in range(n_layers, 0, -1):
        rk = rows[k]
        deg = rp[rk + 1] - rp[rk]
        out_rowptr = torch.zeros(rk.numel() + 1, dtype=torch.int32, device=rk.device)
        out_rowptr[1:] = torch.cumsum(deg, 0)
        m = int(out_rowptr[-1])                                   # the one host sync per layer
        out_col = torch.empty(max(m, 1), dtype=torch.int32, device=rk.device)
        out_val = torch.empty(max(m, 1), dtype=torch.float32, device=rk.device)
        if m:                                    # (no entry to extract: an empty matrix, or rows that are nobody's head)
            N.call("lkg_csr_extract_rows", rk.numel(), N.ptr(rk), N.ptr(rp), N.ptr(graph.col), N.ptr(val),
                   N.ptr(out_rowptr), N.ptr(out_col), N.ptr(out_val), ops._stream())
        col_ids = out_col[:m].long()
        prev = torch.unique(torch.cat([rk, col_ids]))
        if max_rows is not None and prev.numel() > max_rows:
            return None
        rows[k - 1] = prev
        layers[k] = SubLayer(out_rowptr, torch.searchsorted(prev, col_ids).int(), out_val[:m], rk.numel(),
                             prev.numel(), torch.searchsorted(prev, rk))
    return BatchSubgraph(rows, layers)


class _GatherRows(Function):
    """table[ids] with the dense scatter-add backward of torch's `index` (model.py:382-384)."""

    @staticmethod
    def forward(ctx, table, ids):
        ctx.save_for_backward(ids)
        ctx.shape = table.shape
        return ops.gather_rows(table, ids)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        g = ops._f32_rows(g)
        out = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        N.call("lkg_scatter_add_rows_f32", ids.numel(), g.shape[1], N.ptr(g), ops._ld(g), N.ptr(ids), None,
               N.ptr(out), ops._ld(out), ops._stream())
        return out, None


def gather_rows(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    return _GatherRows.apply(table, ops._i64(ids))


class _CompactAggregate(Function):
    """out = A_sub @ source (+ own rows); backward scatters into a zeroed compact gradient."""

    @staticmethod
    def forward(ctx, source, own, sub: SubLayer):
        ctx.sub = sub
        ctx.src_shape = source.shape
        ctx.has_own = own is not None
        return ops.spmm_raw(sub.rowptr, sub.col, sub.val, source, sub.n_rows, add_self=own)

    @staticmethod
    def backward(ctx, g):
        sub = ctx.sub
        g = ops._f32_rows(g)
        gs = torch.zeros(ctx.src_shape, dtype=torch.float32, device=g.device)
        if sub.col.numel():                      # (a batch whose rows have no stored entry: nothing reaches the source rows)
            N.call("lkg_spmm_csr_scatter_bwd_f32", sub.n_rows, g.shape[1], N.ptr(sub.rowptr), N.ptr(sub.col),
                   N.ptr(sub.val), N.ptr(g), ops._ld(g), N.ptr(gs), ops._ld(gs), ops._stream())
        return gs, (g if ctx.has_own else None), None


class CompactAttention:
    """Stands in for AttentionCSR inside Aggregator.forward: the layer receives ITS rows of the previous
    layer as `ego`, while the neighbour sum gathers from the whole compact previous layer."""

    def __init__(self, sub: SubLayer, source: torch.Tensor):
        self.sub, self.source = sub, source

    def aggregate(self, ego: torch.Tensor, plus_self: bool = False) -> torch.Tensor:
        return _CompactAggregate.apply(self.source, ego if plus_self else None, self.sub)
