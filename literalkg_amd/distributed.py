"""Multi-GPU ``LiteralKG`` inside one node: one process per GPU over ``torch.distributed`` (backend "nccl" = RCCL over
xGMI; the same code runs under "gloo").  Every mode of the reference's module is served: ``pre_training``,
``update_att``, ``fine_tuning``, ``predict``, ``mlp`` (model.py:521-532).

The reference scales by wrapping the whole model in ``nn.DataParallel`` (main_pretraining.py:69-71,
main_finetuning.py:71): every replica recomputes all N rows.  Here the ROWS are the unit of parallelism:
rank g owns the entity rows [lo_g, hi_g) of every N-row tensor -- its shard of ``entity_embed`` (and of that shard's
Adam state), its rows of the literals, of every layer output and of the concatenated table -- and runs the dense part of
every layer (gate, Linear, LeakyReLU, LayerNorm, normalised copy, ``linear_gat``) on that row block with the module's own
code.  Row blocks are cut by equal rows or balanced by stored entries (``lkg_row_partition``); collectives move blocks
padded to the longest one, and every structure lives in PADDED coordinates (row i of rank g sits at g * block + i), so a
gathered table is indexed by the structure's column ids as it arrives -- no compaction pass.

Only the aggregation ``A_in @ ego`` needs other ranks' rows.  FORWARD, two exchange schemes:

  scheme "rows"      (the north star's edge-range sharding) all-gather of the layer input (N x D), SpMM over the rank's
                     head rows.
  scheme "features"  row block -> column slab (N x D/G), SpMM over the WHOLE graph on the rank's columns, column slab ->
                     row block; both exchanges pipelined with the SpMM (sharding.FeatureShardedAggregation: every incoming
                     block arrives in row sub-ranges and the SpMM runs part by part behind them, the last part leaves
                     owner range by owner range, this rank's own rows last).  8x less traffic than gathering N x D tables at
                     G = 8.

BACKWARD, chosen per call by all ranks together (one tiny all-gather of message counts):

  frontier exchange  the loss reads <= 3B rows (model.py:382-384), so the gradient that reaches an aggregation is zero
                     outside a few rows F of this rank's block.  A^T g then touches only the tails of F's entries: the rank
                     extracts those rows of its CSR, sums the contributions per tail (a compact scatter over |F| x degree
                     entries -- no pass over N rows or over the edge list), and sends every tail's row to its owner
                     (all-to-all of a few thousand rows).  The owner adds what it receives into a table that is kept
                     all-zero elsewhere and tells the layer below which rows those are, so the next aggregation's backward
                     is a frontier exchange again.  Exact: only zero addends are dropped.
  dense fallback     (a rank without a row set, or a frontier that is no longer small) rows: transpose SpMM of the rank's
                     slice into the padded N x D table + REDUCE-SCATTER (the north star's all-reduce of the
                     entity-gradient table, every rank keeping the rows whose optimizer state it owns); features: the
                     pipelined exchange above on the CSC.

The loss heads read a few thousand rows of the concatenated table: every rank contributes the requested rows it owns
(zeros elsewhere), one all-reduce makes them whole everywhere, every rank evaluates the (small) head on the full batch
with the module's own head code, and the row gradients flow back to their owners without communication.  Gradients of the
replicated layer weights are partial sums over row blocks: ``sync_gradients()`` all-reduces them in one flat bucket over a
FIXED parameter list; the heads' own parameters (``gat_trans_M``, ``relation_embed``, the MLP head) see the whole batch on
every rank and need no exchange.

``TRAFFIC`` counts the payload bytes handed to the collective library by kind (tests hold the step to its budget).
Kernels are injected (``kernels=``) so that the CPU rehearsal tests can run the same distributed logic on torch ops; the
product default is the HIP library and nothing else.
"""
from __future__ import annotations

import collections
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Function

from .graph import KGStructure
from .sharding import FeatureShardedAggregation
from .transport import InFlight, Pending, Transport

TRAFFIC: collections.Counter = collections.Counter()      # bytes sent by this rank, by kind of exchange


def _sent(kind: str, nbytes: int):
    TRAFFIC[kind] += int(nbytes)


class HipKernels:
    """The library's ops behind the small interface the distributed step needs."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def spmm(self, rowptr, col, val, x, n_rows, out=None, x_row_offset=0, long_rows=None, add_self=None, add2=None):
        return self.ops.spmm_raw(rowptr, col, val, x, n_rows, out=out, x_row_offset=x_row_offset,
                                 long_rows=long_rows, add_self=add_self, add2=add2)

    def permute(self, val, perm):
        return self.ops.permute_values(val, perm)

    def edge_softmax(self, g, ent, rel, row_lo=0, row_hi=None):
        return self.ops.edge_softmax(g, ent, rel, row_lo=row_lo, row_hi=row_hi)[0]

    def add(self, a, b):
        return self.ops._elt(0, a, b, 1.0, 1.0)

    def checked_ids(self, n_rows, ids, what):
        return self.ops.checked_ids(n_rows, ids, what=what)[0]

    def gather_rows_range(self, block, ids, lo, hi):
        return self.ops.gather_rows_range(block, self.ops._i64(ids), int(lo), int(hi))

    def gather_rows(self, table, ids):
        return self.ops.gather_rows(table, ids)

    # ---- row sets: which rows of a gradient may be non-zero
    def row_set(self, grad, force: bool = False):
        """(sorted unique int64 row ids, or None = treat as dense).  The ids come from the RowSet a producer of this
        package left on the tensor; ``force`` skips the size test (small test graphs)."""
        ops = self.ops
        rs = ops.tagged_rows(grad)
        if rs is None or not (force or ops.rows_worth_compacting(rs, grad.shape[0])):
            return None
        ids = rs.compact_ids()
        return torch.unique(ids[ids >= 0])

    def rows_table(self, n, d, device, id_lists: Sequence[torch.Tensor], row_lists: Sequence[torch.Tensor], pool: str):
        """An n x d table holding sum of the given rows at the given ids and zero elsewhere -- the shared kept-zero
        table of ops._RowScratch (no n x d fill), tagged with its row set for the consumers of this package."""
        from . import _native as N
        ops = self.ops
        id_lists = [ops._i64(i) for i in id_lists]
        flags = torch.zeros(n, dtype=torch.uint8, device=device)
        for ids in id_lists:
            if ids.numel():
                N.call("lkg_fill_rows_f32", ids.numel(), 0, N.ptr(ids), None, 0, 0.0, N.ptr(flags), 1, ops._stream())
        t = ops.zero_table_for(ops.RowSet(flags, id_lists), n, d, device, pool)
        for ids, rows in zip(id_lists, row_lists):
            if ids.numel():
                rows = ops._f32_rows(rows)
                N.call("lkg_scatter_add_rows_range_f32", ids.numel(), d, N.ptr(rows), ops._ld(rows), N.ptr(ids), 0, int(n),
                       N.ptr(t), ops._ld(t), ops._stream())
        return t

    def frontier_messages(self, rowptr, col, val, grad, rows, row0: int):
        """A[rows, :]^T grad[rows - row0, :] as (tails, one summed row per tail): ``rows`` = sorted ids of the structure's
        rows whose gradient row may be non-zero.  The entries of those rows are extracted into a compact CSR, relabelled to
        the positions of their distinct tails, and the contributions val * grad[row] are scatter-added per tail: work
        proportional to the frontier, nothing proportional to N or to the edge list."""
        from . import _native as N
        ops = self.ops
        d = grad.shape[1]
        if rows.numel() == 0:
            return rows, torch.zeros((0, d), dtype=torch.float32, device=grad.device)
        deg = rowptr[rows + 1] - rowptr[rows]
        out_rowptr = torch.zeros(rows.numel() + 1, dtype=torch.int32, device=rows.device)
        out_rowptr[1:] = torch.cumsum(deg, 0)
        m = int(out_rowptr[-1])                                    # (host sync: the frontier's size decides the shapes)
        out_col = torch.empty(max(m, 1), dtype=torch.int32, device=rows.device)
        out_val = torch.empty(max(m, 1), dtype=torch.float32, device=rows.device)
        if m:
            N.call("lkg_csr_extract_rows", rows.numel(), N.ptr(rows), N.ptr(rowptr), N.ptr(col), N.ptr(val),
                   N.ptr(out_rowptr), N.ptr(out_col), N.ptr(out_val), ops._stream())
        tails, pos = torch.unique(out_col[:m].long(), return_inverse=True)
        gc = ops.gather_rows_range(grad, rows - row0, 0, grad.shape[0])
        buf = torch.zeros((tails.numel(), d), dtype=torch.float32, device=grad.device)
        pos32 = pos.int()                                          # (a local: it outlives the launch that reads it)
        if m:
            N.call("lkg_spmm_csr_scatter_bwd_f32", rows.numel(), d, N.ptr(out_rowptr), N.ptr(pos32), N.ptr(out_val),
                   N.ptr(gc), ops._ld(gc), N.ptr(buf), ops._ld(buf), ops._stream())
        return tails, buf


class RowPartition:
    """Rank g owns rows [cuts[g], cuts[g+1]) (equal blocks R = ceil(N / G) unless ``cuts`` says otherwise).  Collectives
    move blocks padded to ``block`` = the longest one; PADDED coordinates put row i of rank g at g * block + i."""

    def __init__(self, n: int, rank: int, world: int, cuts: Optional[Sequence[int]] = None):
        self.n, self.rank, self.world = int(n), int(rank), int(world)
        if cuts is None:
            eq = -(-self.n // self.world)
            cuts = [min(self.n, g * eq) for g in range(self.world + 1)]
        self.cuts = [int(c) for c in cuts]
        if (len(self.cuts) != self.world + 1 or self.cuts[0] != 0 or self.cuts[-1] != self.n
                or any(b < a for a, b in zip(self.cuts, self.cuts[1:]))):
            raise ValueError(f"row cuts {self.cuts} do not partition [0, {self.n}) into {self.world} ranges")
        self.block = max(1, max(b - a for a, b in zip(self.cuts, self.cuts[1:])))
        self.n_pad = self.world * self.block
        self.lo, self.hi = self.cuts[self.rank], self.cuts[self.rank + 1]
        self.rows = self.hi - self.lo
        self.shift = [g * self.block - self.cuts[g] for g in range(self.world)]
        self.identity = all(s_ == 0 for s_ in self.shift)
        self.pad_lo = self.rank * self.block                     # my rows in padded coordinates: [pad_lo, pad_lo + rows)
        self._dev = {}

    @classmethod
    def balanced(cls, n: int, rank: int, world: int, head_ids: torch.Tensor) -> "RowPartition":
        """Cuts balanced by stored entries: ``head_ids`` = the head of every entry."""
        from . import _native as N
        counts = np.bincount(head_ids.detach().cpu().numpy().astype(np.int64), minlength=n)
        rowptr = np.zeros(n + 1, np.int32)
        np.cumsum(counts, out=rowptr[1:])
        cuts = np.empty(world + 1, np.int64)
        N.call("lkg_row_partition", int(n), N.ptr(rowptr), int(world), N.ptr(cuts))
        return cls(n, rank, world, [int(c) for c in cuts])

    def bounds(self, g: int):
        return self.cuts[g], self.cuts[g + 1]

    def _tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.tensor(self.cuts[1:-1], dtype=torch.int64, device=device),
                              torch.tensor(self.shift, dtype=torch.int64, device=device))
        return self._dev[key]

    def owner(self, ids: torch.Tensor) -> torch.Tensor:
        inner, _ = self._tables(ids.device)
        return torch.bucketize(ids, inner, right=True)

    def to_padded(self, ids: torch.Tensor) -> torch.Tensor:
        """global row ids -> padded coordinates (monotone: sorted lists stay sorted)"""
        ids = ids.long()
        if self.identity:
            return ids
        _, shift = self._tables(ids.device)
        return ids + shift[self.owner(ids)]

    def from_padded(self, ids: torch.Tensor) -> torch.Tensor:
        ids = ids.long()
        if self.identity:
            return ids
        _, shift = self._tables(ids.device)
        return ids - shift[torch.div(ids, self.block, rounding_mode="floor")]

    def pad(self, x: torch.Tensor) -> torch.Tensor:
        """my block of a table, padded with zero rows to ``block`` rows"""
        if x.shape[0] == self.block:
            return x.contiguous()
        out = x.new_zeros((self.block,) + tuple(x.shape[1:]))
        out[:x.shape[0]] = x
        return out


# ----------------------------------------------------------------------------------------------- collectives
# Every exchange goes through ``transport.Transport`` (which alone knows whether RCCL or gloo moves the bytes); the helpers
# below only add the per-kind traffic count.
def _counted(tp: Transport, kind: str, pend: Pending, before: int) -> Pending:
    _sent(kind, tp.bytes_sent - before)
    return pend


def _all_gather(x: torch.Tensor, world: int, group, kind: str = "all_gather", async_op: bool = False):
    """[R, ...] blocks -> [world * R, ...]; async_op: returns a Pending issued now (the caller waits where it needs it)."""
    tp = Transport(group)
    pend = _counted(tp, kind, tp.all_gather(x), 0)
    return pend if async_op else pend.wait()


def _reduce_scatter(x: torch.Tensor, world: int, group, kind: str = "reduce_scatter") -> torch.Tensor:
    """x: [world * R, ...] partial sums -> this rank's [R, ...] block of the sum."""
    tp = Transport(group)
    return _counted(tp, kind, tp.reduce_scatter(x), 0).wait()


def _all_to_all_rows(rows: torch.Tensor, send_counts: List[int], recv_counts: List[int], group, kind: str) -> torch.Tensor:
    """rows [sum(send_counts), ...] sorted by destination -> [sum(recv_counts), ...] sorted by source."""
    tp = Transport(group)
    out = torch.empty((sum(recv_counts),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    return _counted(tp, kind, tp.all_to_all(out, rows, recv_counts, send_counts), 0).wait()


def _all_reduce(x: torch.Tensor, group, kind: str = "all_reduce") -> torch.Tensor:
    tp = Transport(group)
    return _counted(tp, kind, tp.all_reduce(x), 0).wait()


# ----------------------------------------------------------------------------------------------- aggregation
class DistributedAttention:
    """What a layer needs of A_in under row sharding: ``aggregate(ego_block, plus_self)`` -> side rows of this rank.
    The structure lives in padded coordinates.  scheme "rows": the rank's head rows only; scheme "features": the whole
    graph, replicated.  ``sparse_backward``: "auto" (frontier exchange when every rank's gradient carries a small row
    set), "always" (whenever it carries one: small test graphs), "never"."""

    def __init__(self, scheme: str, graph: KGStructure, val: torch.Tensor, part: RowPartition, kernels, group=None,
                 sparse_backward: str = "auto", pipelined: bool = True):
        if scheme not in ("rows", "features"):
            raise ValueError(scheme)
        if sparse_backward not in ("auto", "always", "never"):
            raise ValueError(sparse_backward)
        if graph.n != part.n_pad:
            raise ValueError(f"the structure has {graph.n} rows, the padded row space {part.n_pad}")
        self.scheme, self.graph, self.part, self.kernels, self.group = scheme, graph, part, kernels, group
        self.sparse_backward = sparse_backward
        self.pipelined = bool(pipelined)
        self.fs = None
        if scheme == "features":       # (its width argument only sizes the plain slab methods, which are not used here)
            self.fs = FeatureShardedAggregation(graph, val, part.rank, part.world, part.world,
                                                [g * part.block for g in range(part.world + 1)],
                                                spmm=kernels.spmm, permute=kernels.permute, group=group)
            self.val, self.val_t = val, self.fs.val_t
        else:
            self.set_values(val)

    def set_values(self, val: torch.Tensor):
        self.val = val
        if self.fs is not None:
            self.fs.set_values(val)
            self.val_t = self.fs.val_t
        else:
            self.val_t = self.kernels.permute(val, self.graph.t_perm)

    def aggregate(self, ego: torch.Tensor, plus_self: bool = False) -> torch.Tensor:
        return _Aggregate.apply(ego, self, plus_self)

    def features_pass(self, transposed: bool, panels: torch.Tensor, plus_self: bool) -> torch.Tensor:
        """side (transposed: A^T g) of scheme "features" as the row block [G, block, dg] of this rank's rows.
        pipelined (default): both exchanges folded into the SpMM (FeatureShardedAggregation.exchange_aggregate).
        pipelined = False: the same result from three plain steps -- ONE all-to-all (row block -> column slab), ONE
        whole-structure SpMM, ONE all-to-all back: no side stream, no point-to-point group.  It exists so that a first run on
        real links has a form to bisect against (and a fallback that uses nothing but ``all_to_all_single``)."""
        fs = self.fs
        if self.pipelined:
            return fs.exchange_aggregate(transposed, block_in=panels, plus_self=plus_self)[1]
        G, rows, dg = panels.shape
        g = self.graph
        slab = torch.empty((g.n, dg), dtype=panels.dtype, device=panels.device)
        counts = [rows] * G
        fs.tp.all_to_all(slab, panels.view(G * rows, dg), counts, counts).wait()
        rp, col, val = (g.t_rowptr, g.t_col, fs.val_t) if transposed else (g.rowptr, g.col, fs.val)
        side = self.kernels.spmm(rp, col, val, slab, g.n, long_rows=g.long_rows(transposed),
                                 add_self=slab if plus_self else None)
        out = torch.empty_like(panels)
        fs.tp.all_to_all(out.view(G * rows, dg), side, counts, counts).wait()
        return out

    # ---- layout helpers of the features scheme: my rows [rows, D] <-> G column panels [G, block, dg]
    def slab_width(self, d: int) -> int:
        dg = -(-int(d) // self.part.world)
        return -(-dg // 4) * 4                                   # 16-byte rows for the vector path of the SpMM

    def to_panels(self, x: torch.Tensor) -> torch.Tensor:
        p = self.part
        rows, d = x.shape
        dg = self.slab_width(d)
        panels = torch.empty((p.world, p.block, dg), dtype=x.dtype, device=x.device)
        if rows < p.block:
            panels[:, rows:].zero_()
        if d == p.world * dg:
            panels[:, :rows].copy_(x.view(rows, p.world, dg).permute(1, 0, 2))        # ONE strided copy
        else:                                                    # a width the ranks do not divide: zero columns pad it
            wide = x.new_zeros((rows, p.world * dg))
            wide[:, :d] = x
            panels[:, :rows].copy_(wide.view(rows, p.world, dg).permute(1, 0, 2))
        return panels

    def from_panels(self, panels: torch.Tensor, d: int) -> torch.Tensor:
        p = self.part
        dg = panels.shape[2]
        if d == p.world * dg:
            out = torch.empty((p.rows, d), dtype=panels.dtype, device=panels.device)
            out.view(p.rows, p.world, dg).copy_(panels[:, :p.rows].permute(1, 0, 2))  # ONE strided copy
            return out
        wide = torch.empty((p.rows, p.world * dg), dtype=panels.dtype, device=panels.device)
        wide.view(p.rows, p.world, dg).copy_(panels[:, :p.rows].permute(1, 0, 2))
        return wide[:, :d].contiguous()


class _Aggregate(Function):
    """side rows of this rank = (A @ table)[my rows] (+ ego); see the module docstring for the exchanges."""

    @staticmethod
    def forward(ctx, ego, att: DistributedAttention, plus_self):
        p, k, g = att.part, att.kernels, att.graph
        ctx.att, ctx.plus_self = att, plus_self
        ego = ego.contiguous()
        if att.scheme == "rows":
            with InFlight(ego.device) as fl:       # (an exception below leaves with the gather waited for)
                pend = fl.add(_all_gather(p.pad(ego), p.world, att.group, "aggregate_forward", async_op=True))   # issued now ...
                long_rows = g.long_rows(False, p.pad_lo, p.pad_lo + p.rows)
                table = pend.wait()                                                                  # ... needed here
                return k.spmm(g.rowptr[p.pad_lo:p.pad_lo + p.rows + 1], g.col, att.val, table, p.rows, long_rows=long_rows,
                              add_self=ego if plus_self else None)
        fs = att.fs
        before = fs.bytes_sent
        out = att.features_pass(False, att.to_panels(ego), plus_self)
        _sent("aggregate_forward", fs.bytes_sent - before)
        return att.from_panels(out, ego.shape[1])

    @staticmethod
    def backward(ctx, grad):
        att = ctx.att
        p, k, g = att.part, att.kernels, att.graph
        grad = grad.contiguous()
        d = grad.shape[1]
        dev = grad.device
        # ---- does every rank hold a small row set?  (one all-gather of G + 1 integers; the answer is the same everywhere)
        rows = None if att.sparse_backward == "never" else k.row_set(grad, force=att.sparse_backward == "always")
        tails = msgs = None
        counts = torch.full((p.world + 1,), -1, dtype=torch.int64)
        if rows is not None:
            # my flagged rows F (local ids) -> the tails of their entries, one summed row per tail (padded coordinates)
            tails, msgs = k.frontier_messages(g.rowptr, g.col, att.val, grad, rows + p.pad_lo, p.pad_lo)
            edges = torch.tensor([g * p.block for g in range(p.world + 1)], dtype=torch.int64, device=tails.device)
            at = torch.searchsorted(tails, edges)
            counts = torch.cat([(at[1:] - at[:-1]).cpu(), torch.tensor([rows.numel()])])
        cdev = Transport(att.group).control_device(dev)
        every = _all_gather(counts.to(cdev), p.world, att.group, "frontier_counts").view(p.world, p.world + 1).cpu()
        # (frontier rows + flagged rows over all ranks against the table the dense exchange would move: N_pad rows per rank)
        small = att.sparse_backward == "always" or int(every[:, :p.world].sum() + every[:, p.world].sum()) <= p.n_pad // 2
        if bool((every >= 0).all()) and small:
            send = [int(c) for c in every[p.rank, :p.world]]
            recv = [int(c) for c in every[:, p.rank]]
            if int(every[:, :p.world].sum()) == 0:               # (every rank sees the same matrix: nobody has anything to send)
                got_ids = torch.zeros(0, dtype=torch.int64, device=dev)
                got_rows = torch.zeros((0, d), dtype=grad.dtype, device=dev)
            else:
                got_ids = _all_to_all_rows(tails, send, recv, att.group, "frontier_ids") - p.pad_lo
                got_rows = _all_to_all_rows(msgs, send, recv, att.group, "frontier_rows")
            id_lists, row_lists = [got_ids], [got_rows]
            if ctx.plus_self:                                    # ego + side: the gradient also reaches ego's own rows F
                id_lists.append(rows)
                row_lists.append(k.gather_rows(grad, rows))
            return k.rows_table(p.rows, d, dev, id_lists, row_lists, "g_agg_dist"), None, None
        # ---- dense
        if att.scheme == "rows":
            partial = torch.empty((p.n_pad, d), dtype=grad.dtype, device=dev)
            k.spmm(g.t_rowptr, g.t_col, att.val_t, grad, p.n_pad, out=partial, x_row_offset=p.pad_lo,
                   long_rows=g.long_rows(True))
            mine = _reduce_scatter(partial, p.world, att.group, "aggregate_backward")[:p.rows]
            if ctx.plus_self:
                mine = k.add(mine, grad)
            return mine, None, None
        fs = att.fs
        before = fs.bytes_sent
        out = att.features_pass(True, att.to_panels(grad), ctx.plus_self)
        _sent("aggregate_backward", fs.bytes_sent - before)
        return att.from_panels(out, d), None, None


# ----------------------------------------------------------------------------------------------- loss rows
class _GatherBatchRows(Function):
    """rows[i] = table[ids[i]] for GLOBAL ids over a row-sharded table: owned rows + all-reduce.  Backward: every rank
    holds the same row gradients (the head is evaluated on the whole batch everywhere) and keeps the rows it owns -- in a
    table that is zero elsewhere and says so (the frontier exchange of the layers below starts from that row set)."""

    @staticmethod
    def forward(ctx, block, ids, part, kernels, group):
        rows = kernels.gather_rows_range(block, ids, part.lo, part.hi)
        _all_reduce(rows, group, "head_rows")
        ctx.save_for_backward(ids)
        ctx.meta = (part, kernels, tuple(block.shape), block.device)
        return rows

    @staticmethod
    def backward(ctx, grad):
        (ids,) = ctx.saved_tensors
        part, kernels, shape, device = ctx.meta
        mine = (ids >= part.lo) & (ids < part.hi)
        local = torch.where(mine, ids - part.lo, torch.full_like(ids, -1))          # -1: another rank's row (skipped)
        return kernels.rows_table(shape[0], shape[1], device, [local], [grad.contiguous()], "g_head_dist"), None, None, None, None


# ----------------------------------------------------------------------------------------------- the module
HEAD_PARAMETERS = ("gat_trans_M", "relation_embed.weight", "fc1.", "fc2.", "fc3.", "norm1.", "norm2.")
LOSS_HEAD = HEAD_PARAMETERS      # (name kept for callers of round 2)


class ShardedLiteralKG(nn.Module):
    """One rank's share of a ``LiteralKG`` over the GPUs of a node.  Build it from the state_dict of the full model
    (``from_full``; every rank passes the same one) and drive it like the reference's module -- the SAME global ids on
    every rank:

        loss = model(h, r, pos_t, neg_t, device=dev, mode="pre_training")
        loss.backward(); model.sync_gradients(); optimizer.step()           # optimizer over model.parameters()
        model(h_list, t_list, r_list, relations, device=dev, mode="update_att")
        model(h, pos_t, neg_t, device=dev, mode="fine_tuning");  model(heads, tails, device=dev, mode="predict" | "mlp")

    ``model.parameters()`` = this rank's rows of ``entity_embed`` + the replicated weights, so any optimizer keeps the
    entity table's state sharded (no optimizer traffic).  ``full_state_dict()`` gathers a reference-shaped checkpoint."""

    def __init__(self, local, part: RowPartition, scheme: str, kernels, group=None, sparse_backward: str = "auto",
                 pipelined: bool = True):
        super().__init__()
        self.local = local                   # a LiteralKG over this rank's rows
        self.part, self.scheme, self.kernels, self.group = part, scheme, kernels, group
        self.sparse_backward = sparse_backward
        self.pipelined = bool(pipelined)     # False: plain all-to-all exchanges around one SpMM (DistributedAttention.features_pass)
        self.n_entities = part.n
        self._att: Optional[DistributedAttention] = None
        local._attention = self._attention   # the layers aggregate through the distributed exchange
        local.prune_to_batch = False
        local.id_space = part.n              # callers hand in GLOBAL entity ids
        # the heads read rows of the concatenated table by id: they get the gathered rows and positions into them --
        # the contract of the batch-pruned path (model.LiteralKG._embeddings_and_ids), so the module's own head code runs
        local._can_prune = lambda: True
        local._embeddings_and_ids = self._rows_and_positions

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_full(cls, args, n_entities: int, n_relations: int, state: Dict[str, torch.Tensor], numerical_literals=None,
                  text_literals=None, scoring: str = "transr", scheme: str = "features", device=None, group=None,
                  kernels=None, rank: Optional[int] = None, world: Optional[int] = None,
                  partition: Optional[str] = None, sparse_backward: str = "auto",
                  pipelined: bool = True) -> "ShardedLiteralKG":
        """partition: "rows" (equal row blocks), "entries" (blocks balanced by the stored entries of ``A_in``: the SpMM work of
        scheme "rows") or an explicit list of cut points; default: "entries" for scheme "rows" when the state holds an A_in,
        else "rows".  A rank may own no row at all (more ranks than rows, a cut that leaves a range empty)."""
        from .model import LiteralKG
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
        a_in = state.get("A_in")
        has_a = a_in is not None and a_in._nnz() > 0
        if partition is None:
            partition = "entries" if (scheme == "rows" and has_a) else "rows"
        if isinstance(partition, (list, tuple)):                 # explicit cut points (world + 1 of them, ascending, 0 .. N)
            part = RowPartition(n_entities, rank, world, partition)
        elif partition == "entries":
            if not has_a:
                raise ValueError("partition='entries' needs an A_in in the state_dict")
            part = RowPartition.balanced(n_entities, rank, world, a_in.coalesce().indices()[0])
        elif partition == "rows":
            part = RowPartition(n_entities, rank, world)
        else:
            raise ValueError(partition)
        kernels = kernels if kernels is not None else HipKernels()
        device = torch.device(device if device is not None else "cpu")
        sl = slice(part.lo, part.hi)
        num = numerical_literals[sl] if numerical_literals is not None else None
        txt = text_literals[sl] if text_literals is not None else None
        local = LiteralKG(args, part.rows, n_relations, None, num, txt, scoring=scoring)
        if any(k.startswith("fc1.") for k in state) and not hasattr(local, "fc1"):
            local.initialize_MLP()
        own = {}
        for k, v in state.items():
            if k == "A_in":
                continue
            own[k] = v[sl].clone() if k == "entity_embed.weight" else v
        missing = local.load_state_dict(own, strict=False)
        if [k for k in missing.missing_keys if k != "A_in"] or missing.unexpected_keys:
            raise KeyError(f"state_dict does not match the model: {missing}")
        local.to(device)
        model = cls(local, part, scheme, kernels, group, sparse_backward, pipelined)
        if has_a:
            model.set_attention(a_in.coalesce(), device)
        return model

    def _structure(self, h, t, r, device) -> KGStructure:
        """This rank's structure in padded coordinates: its own head rows (scheme "rows") or every triple."""
        p = self.part
        if self.scheme == "rows":
            mine = (h >= p.lo) & (h < p.hi)
            h, t, r = h[mine], t[mine], (r[mine] if r is not None else None)
        return KGStructure.from_triples(p.n_pad, p.to_padded(h), p.to_padded(t), r, device=device)

    def set_attention(self, a_in: torch.Tensor, device):
        """Adopt a full sparse A_in (the loader's Laplacian, or a checkpoint's): every rank keeps what its scheme needs."""
        idx, vals = a_in.indices().to(device), a_in.values().to(device=device, dtype=torch.float32)
        if self.scheme == "rows":
            keep = (idx[0] >= self.part.lo) & (idx[0] < self.part.hi)
            vals = vals[keep]
        g = self._structure(idx[0], idx[1], None, device)
        self._att = DistributedAttention(self.scheme, g, vals.contiguous(), self.part, self.kernels, self.group,
                                         self.sparse_backward, self.pipelined)

    def _attention(self) -> DistributedAttention:
        if self._att is None:
            raise RuntimeError("no attention matrix yet: pass A_in in the state_dict or run mode='update_att' first")
        return self._att

    # ------------------------------------------------------------------ modes
    def entity_table(self, padded: bool = False) -> torch.Tensor:
        """The whole entity table (all-gather of the shards): N x D, or the padded n_pad x D the structures index."""
        w = self.local.entity_embed.weight.detach()
        p = self.part
        table = _all_gather(p.pad(w), p.world, self.group, "entity_table")
        if padded or p.identity and p.n == p.n_pad:
            return table
        return torch.cat([table[g * p.block:g * p.block + (p.cuts[g + 1] - p.cuts[g])] for g in range(p.world)])

    def update_attention(self, h_list, t_list, r_list, relations):
        """model.py:444-471 over the shards: rank g refreshes the softmax rows it needs -- its own head rows (scheme
        "rows", no exchange beyond the table gather) or all of them (scheme "features": replicated values)."""
        m = self.local
        if relations is not None and any(not 0 <= int(x) < m.n_relations for x in relations):
            raise IndexError(f"update_att: relation id outside [0, {m.n_relations}) in `relations`")
        dev = m.entity_embed.weight.device
        h, t, r = (torch.as_tensor(x).to(dev).long() for x in (h_list, t_list, r_list))
        if relations is not None:
            keep = torch.isin(r, torch.as_tensor(list(relations), dtype=r.dtype, device=dev))
            h, t, r = h[keep], t[keep], r[keep]
        elif hasattr(self.kernels, "checked_ids"):               # every triple's relation id is used as it is: checked
            r = self.kernels.checked_ids(m.n_relations, r, "relation")
        p = self.part
        # entity ids against the GLOBAL id space, before they are moved into padded coordinates: the padded structure has
        # n_pad >= n rows, so an id in [n, n_pad) would otherwise be taken for one of another rank's padding rows (the single
        # module raises for it inside KGStructure.from_triples(n_entities, ...), model.py:446-449 of the reference indexes with it)
        for name, ids in (("head", h), ("tail", t)):
            if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= p.n):
                from ._native import LkgError
                raise LkgError(f"update_att: {name} id outside [0, {p.n}) (the structure build of the single module "
                               "refuses the same triple)")
        g = self._structure(h, t, r, dev)
        table = self.entity_table(padded=True)
        rel = m.relation_embed.weight.detach()
        if self.scheme == "rows":
            val = self.kernels.edge_softmax(g, table, rel, p.pad_lo, p.pad_lo + p.rows)
        else:
            val = self.kernels.edge_softmax(g, table, rel)
        self._att = DistributedAttention(self.scheme, g, val, p, self.kernels, self.group, self.sparse_backward,
                                         self.pipelined)
        m._eval_cache = None

    def batch_rows(self, ids: torch.Tensor) -> torch.Tensor:
        """Rows ``ids`` (global) of the concatenated table, whole on every rank, differentiable."""
        block = self.local._table_for_inference()
        return _GatherBatchRows.apply(block, ids.contiguous(), self.part, self.kernels, self.group)

    def _rows_and_positions(self, *id_lists):
        """(the rows the lists ask for -- every distinct id once --, the lists as positions into them): what the module's
        heads consume in place of (full table, ids).  ``local.gat_rows`` lists the ids of the rows, like the pruned path."""
        flat = torch.cat([i.reshape(-1).long() for i in id_lists])
        uniq, inv = torch.unique(flat, return_inverse=True)
        self.local.gat_rows = uniq
        out, o = [], 0
        for i in id_lists:
            out.append(inv[o:o + i.numel()].view(i.shape))
            o += i.numel()
        return self.batch_rows(uniq), tuple(out)

    def forward(self, *input, device, mode):
        if mode == "update_att":
            self.local.device = device
            return self.update_attention(*input)
        # pre_training / fine_tuning / predict / mlp: the module's own code over the gathered rows (unknown modes: None)
        return self.local(*input, device=device, mode=mode)

    # ------------------------------------------------------------------ gradients / checkpoints
    def partial_grad_parameters(self) -> List[nn.Parameter]:
        """Replicated weights whose gradient is a partial sum over this rank's rows (a FIXED list: the same on every rank
        whatever received a gradient this step)."""
        out = []
        for name, p in self.local.named_parameters():
            if name.startswith(HEAD_PARAMETERS) or name == "entity_embed.weight" or name == "A_in" or not p.requires_grad:
                continue
            out.append(p)
        return out

    def sync_gradients(self):
        """Sum the partial weight gradients over ranks: ONE all-reduce of a flat bucket (a few MB at most).  The bucket
        covers every partial-sum parameter, zeros standing in for a gradient this rank did not produce (a rank without
        rows, a parameter no row of it touched), so that its layout never differs between ranks."""
        ps = self.partial_grad_parameters()
        if not ps or self.part.world == 1:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
        _all_reduce(flat, self.group, "weight_gradients")
        off = 0
        for p in ps:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n

    def full_state_dict(self) -> Dict[str, torch.Tensor]:
        """A reference-shaped state_dict (model.py:169-263 names / shapes) gathered from the shards."""
        sd = {k: v for k, v in self.local.state_dict().items() if k != "A_in"}
        sd["entity_embed.weight"] = self.entity_table()
        if self._att is not None:
            sd["A_in"] = self._full_a_in()
        return sd

    def _full_a_in(self) -> torch.Tensor:
        att, n, p = self._att, self.n_entities, self.part
        idx = p.from_padded(att.graph.coo_indices().reshape(-1)).view(2, -1)
        if self.scheme == "features":
            return torch.sparse_coo_tensor(idx, att.val, (n, n), is_coalesced=True)
        # rows scheme: gather (indices, values) of every rank's rows; row ranges are disjoint and ascending
        cdev = Transport(self.group).control_device(idx.device)
        cnt = torch.tensor([idx.shape[1]], dtype=torch.int64, device=cdev)
        counts = _all_gather(cnt, p.world, self.group, "checkpoint").tolist()
        cap = max(counts)
        pad_i = torch.zeros((cap, 2), dtype=torch.int64, device=idx.device)
        pad_v = torch.zeros(cap, dtype=torch.float32, device=idx.device)
        pad_i[:idx.shape[1]] = idx.t()
        pad_v[:idx.shape[1]] = att.val
        all_i = _all_gather(pad_i, p.world, self.group, "checkpoint").view(p.world, cap, 2)
        all_v = _all_gather(pad_v, p.world, self.group, "checkpoint").view(p.world, cap)
        ii = torch.cat([all_i[g, :c] for g, c in enumerate(counts)]).t().contiguous()
        vv = torch.cat([all_v[g, :c] for g, c in enumerate(counts)])
        return torch.sparse_coo_tensor(ii, vv, (n, n), is_coalesced=True)
