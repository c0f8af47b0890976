"""Multi-GPU sharding of the aggregation hot path inside one node.  One process per GPU;
collectives go through torch.distributed (backend "nccl" is RCCL over xGMI on ROCm; the same code runs
under "gloo" for the CPU rehearsal tests).  Two schemes:

FeatureShardedAggregation (default of bench.py) -- shard the FEATURE dimension.
  Every column of  side = A @ ego  is independent, so rank g keeps the whole CSR/CSC (a few GB at 100 M
  edges -- nothing next to 288 GB) and only the D/G columns [g D/G, (g+1) D/G) of every table.  Forward and
  backward SpMM then need NO collective, the entity-gradient slab is already where its optimizer state
  lives (no gradient all-reduce at all), and a 128-byte row gather runs at the same HBM efficiency as a
  1-KiB one (measured).  The exchange step moves to where the layer needs whole rows (Linear / LayerNorm):
  an all-to-all that turns the N x D/G column slab into a rows_g x D row block and back.  Each rank sends
  only its own slab, (G-1)/G * N*D*4/G bytes, spread over all 7 xGMI links at once -- 16x less traffic than
  all-reducing the N x D table on a ring.  The row block arrives as G column panels [G][rows_g][D/G]; the
  dense part consumes it panel by panel (accumulating MFMA GEMMs, ops.multi_linear) so no transpose copy
  is ever made.

ShardedAggregation -- shard the HEAD ROWS ("edge-range sharding", the reference-shaped data parallelism).

Partition: head rows are cut into contiguous ranges balanced by stored entries (``lkg_row_partition``),
so a softmax row never straddles two ranks and ``update_att`` needs no collective.  Every rank keeps
  * its CSR slice (rows lo..hi) and the transpose of THAT slice (all N tails x its heads),
  * a full replica of the source table it gathers from.
Forward  : rank g produces output rows [lo, hi) -- no exchange.
Backward : grad_ego = A^T grad_side.  The slice's transpose yields a partial N x D sum on every rank;
           the exchange step is the sum over ranks (an all-reduce of the dense entity-gradient table,
           issued per tail-row chunk so that the reduction of chunk c overlaps the SpMM of chunk c+1).
"""
from __future__ import annotations

import contextlib
from typing import Callable, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .graph import KGStructure


class ShardedAggregation:
    def __init__(self, graph: KGStructure, val: torch.Tensor, row_lo: int, row_hi: int,
                 spmm: Optional[Callable] = None, permute: Optional[Callable] = None, n_chunks: int = 4,
                 group=None):
        """graph: structure holding (at least) the entries of head rows [row_lo, row_hi) with GLOBAL ids,
        entries outside that range absent; val: attention values in its entry order."""
        if spmm is None:
            from . import ops
            spmm, permute = ops.spmm_raw, ops.permute_values
        self.spmm, self.graph = spmm, graph
        self.lo, self.hi = int(row_lo), int(row_hi)
        self.val = val
        self.val_t = permute(val, graph.t_perm)
        self.group = group
        n = graph.n
        self.chunks = [(int(a), int(b)) for a, b in zip(np.linspace(0, n, n_chunks + 1)[:-1].astype(np.int64),
                                                         np.linspace(0, n, n_chunks + 1)[1:].astype(np.int64)) if b > a]

    def forward(self, table: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """side rows [lo, hi) = A[lo:hi, :] @ table   (table: full N x D replica)"""
        g = self.graph
        return self.spmm(g.rowptr[self.lo:self.hi + 1], g.col, self.val, table, self.hi - self.lo, out=out,
                         long_rows=g.long_rows(False, self.lo, self.hi))

    def backward(self, grad_rows: torch.Tensor, out: Optional[torch.Tensor] = None, reduce: bool = True
                 ) -> torch.Tensor:
        """grad_rows: (hi-lo) x D gradient of this rank's output rows.  Returns the N x D gradient of the
        source table summed over all ranks (every rank ends with the full reduced table)."""
        g = self.graph
        d = grad_rows.shape[1]
        if out is None:
            out = torch.empty((g.n, d), dtype=grad_rows.dtype, device=grad_rows.device)
        works = []
        multi = reduce and dist.is_initialized() and dist.get_world_size(self.group) > 1
        for a, b in self.chunks:
            self.spmm(g.t_rowptr[a:b + 1], g.t_col, self.val_t, grad_rows, b - a, out=out[a:b],
                      x_row_offset=self.lo, long_rows=g.long_rows(True, a, b))
            if multi:
                works.append(dist.all_reduce(out[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        return out


def shard_bounds(graph: KGStructure, world: int) -> List[int]:
    return [int(c) for c in graph.row_cuts(world)]


class FeatureShardedAggregation:
    """Column-sharded aggregation: see the module docstring.

    graph / val: the FULL structure and attention values (replicated on every rank).
    rank, world, cuts: this rank, the group size and the head-row cut points used for the row layout
    (``shard_bounds``); d: full feature width (must divide by world)."""

    def __init__(self, graph: KGStructure, val: torch.Tensor, rank: int, world: int, d: int, cuts: List[int],
                 spmm: Optional[Callable] = None, permute: Optional[Callable] = None, group=None):
        if spmm is None:
            from . import ops
            spmm, permute = ops.spmm_raw, ops.permute_values
        if d % world:
            raise ValueError(f"feature width {d} does not divide over {world} ranks")
        self.spmm, self.graph, self.val = spmm, graph, val
        self.val_t = permute(val, graph.t_perm)
        self.rank, self.world, self.d, self.dg = rank, world, d, d // world
        self.cuts = [int(c) for c in cuts]
        self.rows = [self.cuts[i + 1] - self.cuts[i] for i in range(world)]
        self.my_rows = self.rows[rank]
        self.group = group

    def column_slab(self, table: torch.Tensor) -> torch.Tensor:
        return table[:, self.rank * self.dg:(self.rank + 1) * self.dg].contiguous()

    def forward(self, slab: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """side[:, my columns] = A @ ego[:, my columns] over ALL head rows; no communication."""
        g = self.graph
        return self.spmm(g.rowptr, g.col, self.val, slab, g.n, out=out, long_rows=g.long_rows(False))

    def backward(self, grad_slab: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """grad_ego[:, my columns] = A^T @ grad_side[:, my columns]; no communication."""
        g = self.graph
        return self.spmm(g.t_rowptr, g.t_col, self.val_t, grad_slab, g.n, out=out, long_rows=g.long_rows(True))

    def forward_to_row_block(self, slab: torch.Tensor, side_slab: Optional[torch.Tensor] = None,
                             out: Optional[torch.Tensor] = None, pieces: Optional[int] = None):
        """forward() fused with to_row_block(): the SpMM runs head-row range by head-row range, and as soon as a
        piece of the rows owned by rank j is done it leaves for rank j (point-to-point, RCCL's stream, one xGMI
        link per peer) while the next piece is being aggregated.  Every peer's range is cut into `pieces` parts
        and the parts are visited peer-major inside a part index, so all 7 links are busy from the first round
        on and only the LAST part of one peer (1 / (G * pieces) of the traffic) is exposed after the SpMM ends.
        Part p of round k computes rows of rank (rank + k) % G and receives from rank (rank - k) % G.
        Returns (side_slab [N, D/G], row_block [G, rows_g, D/G]).

        On the GPU the row-range launches alternate between two side streams: back to back on ONE stream every extra
        launch costs ~25 us of tail and gap (5 M rows x 100 M entries x 32 columns: 1 launch 2.16 ms, 8: 2.30,
        32: 2.94), on two alternating streams the tail of one overlaps the start of the next (32 launches: 2.25 ms)
        and each part still completes -- and leaves -- in order."""
        g = self.graph
        if pieces is None:
            pieces = 4
        if side_slab is None:
            side_slab = torch.empty((g.n, self.dg), dtype=slab.dtype, device=slab.device)
        if out is None:
            out = torch.empty((self.world, self.my_rows, self.dg), dtype=slab.dtype, device=slab.device)
        staged = slab.is_cuda and self.world > 1 and dist.get_backend(self.group) == "gloo"   # host-only transport

        def part(lo, hi, p):      # p-th of `pieces` sub-ranges of [lo, hi)
            n = hi - lo
            return lo + n * p // pieces, lo + n * (p + 1) // pieces

        streams = []
        if slab.is_cuda:
            if not hasattr(self, "_side_streams"):
                self._side_streams = [torch.cuda.Stream(device=slab.device) for _ in range(2)]
            streams = self._side_streams
            main = torch.cuda.current_stream(slab.device)
            for st in streams:
                st.wait_stream(main)
        works, host, step = [], [], 0
        for p in range(pieces):
            for k in range(self.world):
                j, i = (self.rank + k) % self.world, (self.rank - k) % self.world
                lo, hi = part(self.cuts[j], self.cuts[j + 1], p)
                ctx = torch.cuda.stream(streams[step % len(streams)]) if streams else contextlib.nullcontext()
                step += 1
                with ctx:      # the collective library orders its transfer behind the CURRENT stream, i.e. this part
                    if hi > lo:
                        self.spmm(g.rowptr[lo:hi + 1], g.col, self.val, slab, hi - lo, out=side_slab[lo:hi],
                                  long_rows=g.long_rows(False, lo, hi))
                    if k == 0:
                        mlo, mhi = part(0, self.my_rows, p)
                        out[self.rank, mlo:mhi].copy_(side_slab[lo:hi])
                        continue
                    rlo, rhi = part(0, self.my_rows, p)          # the matching part of MY rows, arriving from rank i
                    ops_ = []
                    if hi > lo:
                        snd = side_slab[lo:hi].cpu() if staged else side_slab[lo:hi]
                        ops_.append(dist.P2POp(dist.isend, snd, j, self.group))
                    if rhi > rlo:
                        if staged:
                            rcv = torch.empty((rhi - rlo, self.dg), dtype=out.dtype)
                            host.append((rcv, i, rlo, rhi))
                        else:
                            rcv = out[i, rlo:rhi]
                        ops_.append(dist.P2POp(dist.irecv, rcv, i, self.group))
                    if ops_:
                        works += dist.batch_isend_irecv(ops_)
        if streams:
            for st in streams:
                main.wait_stream(st)
        for w in works:
            w.wait()
        for rcv, i, rlo, rhi in host:
            out[i, rlo:rhi].copy_(rcv)
        return side_slab, out

    def backward_from_row_block(self, block: torch.Tensor, out: Optional[torch.Tensor] = None,
                                pieces: Optional[int] = None) -> torch.Tensor:
        """to_column_slab() fused with backward(): the gradient row block [G, rows_g, D/G] goes back to the column
        layout in `pieces` COLUMN pieces (every column of A^T g is independent), all exchanges queued at once on the
        collective's stream, and the transpose SpMM of piece p runs while piece p + 1 is still on the links.  A piece
        is never narrower than 32 columns (128-byte gathers; narrower rows cost the same time per entry), so the
        exchange of the N = 8 case (D/G = 32) stays in one piece.  Returns grad_ego[:, my columns]."""
        g = self.graph
        if pieces is None:
            pieces = max(1, min(4, self.dg // 32))
        while pieces > 1 and self.dg % pieces:
            pieces -= 1
        if out is None:
            out = torch.empty((g.n, self.dg), dtype=block.dtype, device=block.device)
        if self.world == 1 or pieces == 1:
            return self.backward(self.to_column_slab(block), out=out)
        w = self.dg // pieces
        staged = block.is_cuda and dist.get_backend(self.group) == "gloo"     # host-only transport (rehearsal)
        queue = []
        for p in range(pieces):
            src = block[:, :, p * w:(p + 1) * w].contiguous().view(self.world * self.my_rows, w)
            dst = torch.empty((g.n, w), dtype=block.dtype, device=block.device)
            if staged or not block.is_cuda:
                self._all_to_all(dst, src, self.rows, [self.my_rows] * self.world)
                work = None
            else:
                work = dist.all_to_all_single(dst, src, output_split_sizes=self.rows,
                                              input_split_sizes=[self.my_rows] * self.world, group=self.group,
                                              async_op=True)
            queue.append((work, dst, src))
        for p, (work, dst, _src) in enumerate(queue):
            if work is not None:
                work.wait()
            self.spmm(g.t_rowptr, g.t_col, self.val_t, dst, g.n, out=out[:, p * w:(p + 1) * w],
                      long_rows=g.long_rows(True))
        return out

    def to_row_block(self, slab: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """N x D/G column slab -> this rank's rows as G column panels, shape [G, rows_g, D/G]
        (panel i = columns of rank i).  One all-to-all."""
        if out is None:
            out = torch.empty((self.world, self.my_rows, self.dg), dtype=slab.dtype, device=slab.device)
        if self.world == 1:
            out[0].copy_(slab)
            return out
        self._all_to_all(out.view(self.world * self.my_rows, self.dg), slab, [self.my_rows] * self.world, self.rows)
        return out

    def to_column_slab(self, block: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[G, rows_g, D/G] panels of this rank's rows -> the N x D/G slab of this rank's columns."""
        if out is None:
            out = torch.empty((self.graph.n, self.dg), dtype=block.dtype, device=block.device)
        if self.world == 1:
            out.copy_(block[0])
            return out
        self._all_to_all(out, block.view(self.world * self.my_rows, self.dg), self.rows, [self.my_rows] * self.world)
        return out

    def _all_to_all(self, out, inp, out_splits, in_splits):
        if inp.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo moves host memory only (single-GPU rehearsal of the N>1 path): stage through the host
            host_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(host_out, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                                   group=self.group)
            out.copy_(host_out)
            return
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
