"""Multi-GPU ``pre_training`` / ``update_att`` of the drop-in module inside one node: one process per GPU over
``torch.distributed`` (backend "nccl" = RCCL over xGMI; the same code runs under "gloo").

The reference scales by wrapping the whole model in ``nn.DataParallel`` (main_pretraining.py:69-71): every replica
recomputes all N rows.  Here the ROWS are the unit of parallelism: rank g owns the entity rows
[lo_g, hi_g) of every N-row tensor -- its shard of ``entity_embed`` (and of that shard's Adam state), its rows of the
literals, of every layer output and of the concatenated table -- and runs the dense part of every layer (gate, Linear,
LeakyReLU, LayerNorm, normalised copy, ``linear_gat``) on that row block with the module's own code.  Only the
aggregation ``A_in @ ego`` needs other ranks' rows; two exchange schemes, both autograd Functions:

  scheme "rows"      (the north star's edge-range sharding) forward: all-gather of the layer input (N x D), SpMM over
                     the rank's head rows; backward: transpose SpMM of the rank's slice gives a partial N x D table,
                     REDUCE-SCATTER of the entity-gradient table over ranks (the all-reduce of the north star with
                     every rank keeping only the rows whose optimizer state it owns).
  scheme "features"  forward: all-to-all row block -> column slab (N x D/G), SpMM over the WHOLE graph on the rank's
                     columns (no collective in the SpMM), all-to-all back; backward the same with the CSC.  16x less
                     traffic than moving N x D tables at G = 8 (sharding.py).

The loss reads <= 3B rows of the concatenated table: every rank contributes the batch rows it owns (zeros elsewhere),
one all-reduce makes them whole everywhere, every rank evaluates the (small) loss head on the full batch, and the row
gradients flow back to their owners without communication.  Gradients of the replicated layer weights are partial sums
over row blocks: ``sync_gradients()`` all-reduces them in one flat bucket; the loss head's parameters
(``gat_trans_M``, ``relation_embed``) see the whole batch on every rank and need no exchange.

Kernels are injected (``kernels=``) so that the CPU rehearsal tests can run the same distributed logic on torch ops;
the product default is the HIP library and nothing else.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Function

from .graph import KGStructure


class HipKernels:
    """The library's ops behind the small interface the distributed step needs."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def spmm(self, rowptr, col, val, x, n_rows, out=None, x_row_offset=0, long_rows=None, add_self=None):
        return self.ops.spmm_raw(rowptr, col, val, x, n_rows, out=out, x_row_offset=x_row_offset,
                                 long_rows=long_rows, add_self=add_self)

    def permute(self, val, perm):
        return self.ops.permute_values(val, perm)

    def edge_softmax(self, g, ent, rel, row_lo=0, row_hi=None):
        return self.ops.edge_softmax(g, ent, rel, row_lo=row_lo, row_hi=row_hi)[0]

    def add(self, a, b):
        return self.ops._elt(0, a, b, 1.0, 1.0)

    def gather_rows_range(self, block, ids, lo, hi):
        from . import _native as N
        ops = self.ops
        block = ops._f32_rows(block)
        ids = ops._i64(ids)
        out = torch.empty((ids.numel(), block.shape[1]), dtype=torch.float32, device=block.device)
        N.call("lkg_gather_rows_range_f32", ids.numel(), block.shape[1], N.ptr(block), ops._ld(block), N.ptr(ids),
               int(lo), int(hi), N.ptr(out), block.shape[1], ops._stream())
        return out

    def scatter_add_rows_range(self, rows, ids, lo, hi, like):
        from . import _native as N
        ops = self.ops
        rows = ops._f32_rows(rows)
        ids = ops._i64(ids)
        out = torch.zeros_like(like, memory_format=torch.contiguous_format)
        N.call("lkg_scatter_add_rows_range_f32", ids.numel(), rows.shape[1], N.ptr(rows), ops._ld(rows), N.ptr(ids),
               int(lo), int(hi), N.ptr(out), out.shape[1], ops._stream())
        return out


class RowPartition:
    """Equal row blocks: rank g owns [g R, min(N, (g+1) R)), R = ceil(N / G); collectives move blocks padded to R."""

    def __init__(self, n: int, rank: int, world: int):
        self.n, self.rank, self.world = int(n), int(rank), int(world)
        self.block = -(-self.n // self.world)
        self.lo = min(self.n, self.rank * self.block)
        self.hi = min(self.n, self.lo + self.block)
        self.rows = self.hi - self.lo

    def bounds(self, g: int):
        lo = min(self.n, g * self.block)
        return lo, min(self.n, lo + self.block)

    def pad(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[0] == self.block:
            return x.contiguous()
        out = x.new_zeros((self.block,) + tuple(x.shape[1:]))
        out[:x.shape[0]] = x
        return out


def _staged(t: torch.Tensor, group) -> bool:
    """gloo moves host memory only: device tensors are staged through the host (1-GPU rehearsal of the N > 1 path)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather(x: torch.Tensor, world: int, group) -> torch.Tensor:
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if _staged(x, group):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, x.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def _reduce_scatter(x: torch.Tensor, world: int, group) -> torch.Tensor:
    """x: [world * R, ...] partial sums -> this rank's [R, ...] block of the sum."""
    out = torch.empty((x.shape[0] // world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    backend = dist.get_backend(group)
    if backend == "gloo":      # no reduce_scatter in gloo: all-reduce and slice
        buf = x.cpu() if x.is_cuda else x.clone()
        dist.all_reduce(buf, group=group)
        r = dist.get_rank(group)
        out.copy_(buf[r * out.shape[0]:(r + 1) * out.shape[0]])
    else:
        dist.reduce_scatter_tensor(out, x.contiguous(), group=group)
    return out


def _all_to_all(x: torch.Tensor, group) -> torch.Tensor:
    """x: [world, ...] chunk j goes to rank j; returns [world, ...] with chunk i received from rank i."""
    out = torch.empty_like(x)
    if _staged(x, group):
        host = torch.empty(x.shape, dtype=x.dtype)
        dist.all_to_all_single(host, x.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, x.contiguous(), group=group)
    return out


def _all_reduce(x: torch.Tensor, group) -> torch.Tensor:
    if _staged(x, group):
        host = x.cpu()
        dist.all_reduce(host, group=group)
        x.copy_(host)
    else:
        dist.all_reduce(x, group=group)
    return x


# ----------------------------------------------------------------------------------------------- aggregation
class _RowAggregate(Function):
    """side rows [lo, hi) = A[lo:hi, :] @ table, table = all-gather of the row blocks; backward = reduce-scatter of
    the partial entity-gradient table A[lo:hi, :]^T g."""

    @staticmethod
    def forward(ctx, ego, att, plus_self):
        p, k, g = att.part, att.kernels, att.graph
        table = _all_gather(p.pad(ego), p.world, att.group)
        side = k.spmm(g.rowptr[p.lo:p.hi + 1], g.col, att.val, table, p.rows, long_rows=g.long_rows(False, p.lo, p.hi),
                      add_self=ego if plus_self else None)
        ctx.att, ctx.plus_self = att, plus_self
        return side

    @staticmethod
    def backward(ctx, grad):
        att = ctx.att
        p, k, g = att.part, att.kernels, att.graph
        grad = grad.contiguous()
        partial = torch.zeros((p.world * p.block, grad.shape[1]), dtype=grad.dtype, device=grad.device)
        k.spmm(g.t_rowptr, g.t_col, att.val_t, grad, g.n, out=partial[:g.n], x_row_offset=p.lo,
               long_rows=g.long_rows(True))
        mine = _reduce_scatter(partial, p.world, att.group)[:p.rows]
        if ctx.plus_self:
            mine = k.add(mine, grad)
        return mine, None, None


def _to_panels(block: torch.Tensor, world: int) -> torch.Tensor:
    """[R, D] -> [G, R, D/G] (panel j = columns of rank j)"""
    r, d = block.shape
    return block.view(r, world, d // world).permute(1, 0, 2).contiguous()


def _from_panels(panels: torch.Tensor) -> torch.Tensor:
    g, r, dg = panels.shape
    return panels.permute(1, 0, 2).reshape(r, g * dg)


class _FeatureAggregate(Function):
    """Row block -> column slab (all-to-all), SpMM over the whole graph on D/G columns, and back."""

    @staticmethod
    def forward(ctx, ego, att, plus_self):
        p, k, g = att.part, att.kernels, att.graph
        slab = _all_to_all(_to_panels(p.pad(ego), p.world), att.group).view(p.world * p.block, -1)[:g.n]
        side = k.spmm(g.rowptr, g.col, att.val, slab, g.n, long_rows=g.long_rows(False),
                      add_self=slab if plus_self else None)
        back = torch.zeros((p.world * p.block, side.shape[1]), dtype=side.dtype, device=side.device)
        back[:g.n] = side
        out = _from_panels(_all_to_all(back.view(p.world, p.block, -1), att.group))[:p.rows]
        ctx.att, ctx.plus_self = att, plus_self
        return out.contiguous()

    @staticmethod
    def backward(ctx, grad):
        att = ctx.att
        p, k, g = att.part, att.kernels, att.graph
        slab = _all_to_all(_to_panels(p.pad(grad.contiguous()), p.world), att.group).view(p.world * p.block, -1)[:g.n]
        gs = k.spmm(g.t_rowptr, g.t_col, att.val_t, slab, g.n, long_rows=g.long_rows(True),
                    add_self=slab if ctx.plus_self else None)
        back = torch.zeros((p.world * p.block, gs.shape[1]), dtype=gs.dtype, device=gs.device)
        back[:g.n] = gs
        out = _from_panels(_all_to_all(back.view(p.world, p.block, -1), att.group))[:p.rows]
        return out.contiguous(), None, None


class DistributedAttention:
    """What a layer needs of A_in under row sharding: ``aggregate(ego_block, plus_self)`` -> side rows of this rank.
    scheme "rows": graph = the rank's head rows only (global ids); scheme "features": the whole graph, replicated."""

    def __init__(self, scheme: str, graph: KGStructure, val: torch.Tensor, part: RowPartition, kernels, group=None):
        if scheme not in ("rows", "features"):
            raise ValueError(scheme)
        self.scheme, self.graph, self.part, self.kernels, self.group = scheme, graph, part, kernels, group
        self.set_values(val)

    def set_values(self, val: torch.Tensor):
        self.val = val
        self.val_t = self.kernels.permute(val, self.graph.t_perm)

    def aggregate(self, ego: torch.Tensor, plus_self: bool = False) -> torch.Tensor:
        if self.scheme == "features":
            if ego.shape[1] % self.part.world:
                raise ValueError(f"feature sharding: width {ego.shape[1]} does not divide over {self.part.world} ranks")
            return _FeatureAggregate.apply(ego, self, plus_self)
        return _RowAggregate.apply(ego, self, plus_self)


# ----------------------------------------------------------------------------------------------- loss rows
class _GatherBatchRows(Function):
    """rows[i] = table[ids[i]] for GLOBAL ids over a row-sharded table: owned rows + all-reduce.  Backward: every rank
    holds the same row gradients (the loss head is evaluated on the whole batch everywhere) and keeps the rows it owns."""

    @staticmethod
    def forward(ctx, block, ids, part, kernels, group):
        rows = kernels.gather_rows_range(block, ids, part.lo, part.hi)
        _all_reduce(rows, group)
        ctx.save_for_backward(ids, block)
        ctx.meta = (part, kernels)
        return rows

    @staticmethod
    def backward(ctx, grad):
        ids, block = ctx.saved_tensors
        part, kernels = ctx.meta
        return kernels.scatter_add_rows_range(grad.contiguous(), ids, part.lo, part.hi, block), None, None, None, None


# ----------------------------------------------------------------------------------------------- the module
LOSS_HEAD = ("gat_trans_M", "relation_embed.weight")     # see the whole batch on every rank: gradients already whole


class ShardedLiteralKG(nn.Module):
    """One rank's share of a ``LiteralKG`` trained over the GPUs of a node.  Build it from the state_dict of the full
    model (``from_full``; every rank passes the same one) and drive it like the reference's module:

        loss = model(h, r, pos_t, neg_t, device=dev, mode="pre_training")   # the SAME global batch on every rank
        loss.backward(); model.sync_gradients(); optimizer.step()           # optimizer over model.parameters()
        model(h_list, t_list, r_list, relations, device=dev, mode="update_att")

    ``model.parameters()`` = this rank's rows of ``entity_embed`` + the replicated weights, so any optimizer keeps the
    entity table's state sharded (no optimizer traffic).  ``full_state_dict()`` gathers a reference-shaped checkpoint."""

    def __init__(self, local, part: RowPartition, scheme: str, kernels, group=None):
        super().__init__()
        self.local = local                   # a LiteralKG over this rank's rows
        self.part, self.scheme, self.kernels, self.group = part, scheme, kernels, group
        self.n_entities = part.n
        self._att: Optional[DistributedAttention] = None
        local._attention = self._attention   # the layers aggregate through the distributed exchange
        local.prune_to_batch = False

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_full(cls, args, n_entities: int, n_relations: int, state: Dict[str, torch.Tensor], numerical_literals=None,
                  text_literals=None, scoring: str = "transr", scheme: str = "features", device=None, group=None,
                  kernels=None, rank: Optional[int] = None, world: Optional[int] = None) -> "ShardedLiteralKG":
        from .model import LiteralKG
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
        part = RowPartition(n_entities, rank, world)
        kernels = kernels if kernels is not None else HipKernels()
        device = torch.device(device if device is not None else "cpu")
        sl = slice(part.lo, part.hi)
        num = numerical_literals[sl] if numerical_literals is not None else None
        txt = text_literals[sl] if text_literals is not None else None
        local = LiteralKG(args, part.rows, n_relations, None, num, txt, scoring=scoring)
        own = {}
        for k, v in state.items():
            if k == "A_in":
                continue
            own[k] = v[sl].clone() if k == "entity_embed.weight" else v
        missing = local.load_state_dict(own, strict=False)
        if [k for k in missing.missing_keys if k != "A_in"] or missing.unexpected_keys:
            raise KeyError(f"state_dict does not match the model: {missing}")
        local.to(device)
        model = cls(local, part, scheme, kernels, group)
        a_in = state.get("A_in")
        if a_in is not None and a_in._nnz() > 0:
            model.set_attention(a_in.coalesce(), device)
        return model

    def set_attention(self, a_in: torch.Tensor, device):
        """Adopt a full sparse A_in (the loader's Laplacian, or a checkpoint's): every rank keeps what its scheme needs."""
        idx, vals = a_in.indices(), a_in.values()
        if self.scheme == "rows":
            keep = (idx[0] >= self.part.lo) & (idx[0] < self.part.hi)
            idx, vals = idx[:, keep], vals[keep]
        g = KGStructure.from_triples(self.n_entities, idx[0], idx[1], None, device=device)
        self._att = DistributedAttention(self.scheme, g, vals.to(device=device, dtype=torch.float32).contiguous(),
                                         self.part, self.kernels, self.group)

    def _attention(self) -> DistributedAttention:
        if self._att is None:
            raise RuntimeError("no attention matrix yet: pass A_in in the state_dict or run mode='update_att' first")
        return self._att

    # ------------------------------------------------------------------ modes
    def entity_table(self) -> torch.Tensor:
        """The whole N x D entity table (all-gather of the shards); update_att and evaluation read it."""
        w = self.local.entity_embed.weight.detach()
        return _all_gather(self.part.pad(w), self.part.world, self.group)[:self.n_entities]

    def update_attention(self, h_list, t_list, r_list, relations):
        """model.py:444-471 over the shards: rank g refreshes the softmax rows it needs -- its own head rows (scheme
        "rows", no exchange beyond the table gather) or all of them (scheme "features": replicated values)."""
        dev = self.local.entity_embed.weight.device
        h, t, r = (torch.as_tensor(x).to(dev) for x in (h_list, t_list, r_list))
        if relations is not None:
            keep = torch.isin(r, torch.as_tensor(list(relations), dtype=r.dtype, device=dev))
            h, t, r = h[keep], t[keep], r[keep]
        if self.scheme == "rows":
            mine = (h >= self.part.lo) & (h < self.part.hi)
            h, t, r = h[mine], t[mine], r[mine]
        g = KGStructure.from_triples(self.n_entities, h, t, r, device=dev)
        table = self.entity_table()
        rel = self.local.relation_embed.weight.detach()
        if self.scheme == "rows":
            val = self.kernels.edge_softmax(g, table, rel, self.part.lo, self.part.hi)
        else:
            val = self.kernels.edge_softmax(g, table, rel)
        self._att = DistributedAttention(self.scheme, g, val, self.part, self.kernels, self.group)

    def batch_rows(self, ids: torch.Tensor) -> torch.Tensor:
        """Rows ``ids`` (global) of the concatenated table, whole on every rank, differentiable."""
        block = self.local.gat_embeddings()
        self.local.gat_embed = block
        return _GatherBatchRows.apply(block, ids, self.part, self.kernels, self.group)

    def calc_triplet_loss(self, h, r, pos_t, neg_t):
        from . import ops
        m = self.local
        k = int(m.pre_training_neg_rate)
        b = h.numel()
        group = k if (m.scoring == "transr" and m.group_reuse and k >= m.group_reuse_min_rate
                      and ops.is_grouped_batch(h, r, pos_t, k)) else 1
        # the batch's rows, once per group for (h, t+): positions into the gathered rows replace the entity ids
        hg, pg = h[::group], pos_t[::group]
        n_g = hg.numel()
        rows = self.batch_rows(torch.cat([hg, pg, neg_t]))
        dev = rows.device
        pos_h = torch.arange(n_g, device=dev).repeat_interleave(group)
        pos_p = pos_h + n_g
        pos_n = torch.arange(b, device=dev) + 2 * n_g
        keep = m.last_scores if not m.training else None
        if m.scoring == "transr":
            return ops.transr_loss(rows, m.relation_embed.weight, m.gat_trans_M, pos_h, r, pos_p, pos_n,
                                   m.kg_l2loss_lambda, keep, group)
        return ops.transe_loss(rows, m.relation_embed.weight, pos_h, r, pos_p, pos_n, m.kg_l2loss_lambda, keep)

    def forward(self, *input, device, mode):
        self.local.device = device
        if mode == "pre_training":
            return self.calc_triplet_loss(*input)
        if mode == "update_att":
            return self.update_attention(*input)
        return None

    # ------------------------------------------------------------------ gradients / checkpoints
    def partial_grad_parameters(self) -> List[nn.Parameter]:
        """Replicated weights whose gradient is a partial sum over this rank's rows."""
        out = []
        for name, p in self.local.named_parameters():
            if name in LOSS_HEAD or name == "entity_embed.weight" or name == "A_in" or not p.requires_grad:
                continue
            out.append(p)
        return out

    def sync_gradients(self):
        """Sum the partial weight gradients over ranks: ONE all-reduce of a flat bucket (a few MB at most)."""
        ps = [p for p in self.partial_grad_parameters() if p.grad is not None]
        if not ps or self.part.world == 1:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        _all_reduce(flat, self.group)
        off = 0
        for p in ps:
            n = p.grad.numel()
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
        att, n = self._att, self.n_entities
        if self.scheme == "features":
            return torch.sparse_coo_tensor(att.graph.coo_indices(), att.val, (n, n), is_coalesced=True)
        # rows scheme: gather (indices, values) of every rank's rows; row ranges are disjoint and ascending
        idx = att.graph.coo_indices()
        cnt = torch.tensor([idx.shape[1]], dtype=torch.int64, device=idx.device)
        counts = _all_gather(cnt, self.part.world, self.group).tolist()
        cap = max(counts)
        pad_i = torch.zeros((cap, 2), dtype=torch.int64, device=idx.device)
        pad_v = torch.zeros(cap, dtype=torch.float32, device=idx.device)
        pad_i[:idx.shape[1]] = idx.t()
        pad_v[:idx.shape[1]] = att.val
        all_i = _all_gather(pad_i, self.part.world, self.group).view(self.part.world, cap, 2)
        all_v = _all_gather(pad_v, self.part.world, self.group).view(self.part.world, cap)
        ii = torch.cat([all_i[g, :c] for g, c in enumerate(counts)]).t().contiguous()
        vv = torch.cat([all_v[g, :c] for g, c in enumerate(counts)])
        return torch.sparse_coo_tensor(ii, vv, (n, n), is_coalesced=True)
