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

from typing import Callable, List, Optional

import numpy as np
import torch

from .graph import KGStructure
from .transport import InFlight, Transport


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
        self.tp = Transport(group)
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
        multi = reduce and self.tp.multi
        with InFlight(out.device) as fl:       # (leaves -- also by exception -- with every reduction waited for)
            for a, b in self.chunks:
                self.spmm(g.t_rowptr[a:b + 1], g.t_col, self.val_t, grad_rows, b - a, out=out[a:b],
                          x_row_offset=self.lo, long_rows=g.long_rows(True, a, b))
                if multi:
                    fl.add(self.tp.all_reduce(out[a:b]))
        return out


def shard_bounds(graph: KGStructure, world: int) -> List[int]:
    return [int(c) for c in graph.row_cuts(world)]


class _Whole:
    """The whole CSR / CSC behind the interface of a StructurePart."""

    def __init__(self, graph: KGStructure, transposed: bool):
        self.rowptr, self.col = (graph.t_rowptr, graph.t_col) if transposed else (graph.rowptr, graph.col)
        self.n, self.nnz = graph.n, graph.nnz
        self._graph, self._t = graph, transposed

    def long_rows(self, lo: int = 0, hi: Optional[int] = None):
        return self._graph.long_rows(self._t, lo, hi)


class FeatureShardedAggregation:
    """Column-sharded aggregation: see the module docstring.

    graph / val: the FULL structure and attention values (replicated on every rank).
    rank, world, cuts: this rank, the group size and the row cut points of the row layout (rank i owns rows
    [cuts[i], cuts[i+1]) of every N-row table in the row-block layout); d: full feature width, a multiple of world
    (``slab_width`` pads other widths).

    Layouts:  column slab  [N, D/G]         all rows, this rank's columns (what the SpMM works on);
              row block    [G, rows_r, D/G] this rank's rows as G column panels (panel i = the columns of rank i) --
                                            what leaves for / arrives from the other ranks, one panel per peer.

    The two exchanges of a pass are pipelined with the SpMM from both ends (``exchange_aggregate``):
      * IN  (row block -> column slab): xGMI is a point-to-point mesh, one link per peer, so the exchange lasts as long as
        ONE block on ONE link however the peers are grouped; what pipelines is a cut of every block into row sub-ranges.
        Batch q moves sub-range q of all G - 1 incoming blocks at once (all links busy; all batches queued up front), and
        the SpMM runs part by part -- part 0 = the entries that gather from the rank's OWN block (nothing to wait for),
        part q + 1 = those that gather from sub-range q of the other blocks (``KGStructure.structure_parts``), each
        accumulating onto the parts before it -- so part q + 1 runs while batch q + 1 is on the links;
      * OUT (column slab -> row block): the LAST part runs owner range by owner range, the other ranks' rows first (each
        range in `pieces` pieces, piece-major: all links busy from the first round on), every finished piece leaving
        point-to-point while the next one is computed, and the rank's own rows last: the final remote piece (1 / pieces of
        a block on its link) travels behind 1 / G of the pass.
    """

    def __init__(self, graph: KGStructure, val: torch.Tensor, rank: int, world: int, d: int, cuts: List[int],
                 spmm: Optional[Callable] = None, permute: Optional[Callable] = None, group=None):
        if spmm is None:
            from . import ops
            spmm, permute = ops.spmm_raw, ops.permute_values
        if d % world:
            raise ValueError(f"feature width {d} does not divide over {world} ranks (pad it: slab_width)")
        self.spmm, self.permute, self.graph = spmm, permute, graph
        self.rank, self.world, self.d, self.dg = rank, world, d, d // world
        self.cuts = [int(c) for c in cuts]
        if len(self.cuts) != world + 1 or self.cuts[0] != 0 or self.cuts[-1] != graph.n:
            raise ValueError("row cuts must run from 0 to the number of entities, one range per rank")
        self.rows = [self.cuts[i + 1] - self.cuts[i] for i in range(world)]
        self.my_rows = self.rows[rank]
        self.group = group
        self.tp = Transport(group)    # every byte that leaves this rank goes through it (and is counted there)
        self._parts = {}              # (transposed, batches) -> (StructureParts, their values)
        self.set_values(val)

    @property
    def bytes_sent(self) -> int:
        """payload bytes this rank handed to the collective library (tests count them)"""
        return self.tp.bytes_sent

    @staticmethod
    def slab_width(d: int, world: int) -> int:
        """Columns per rank for a table of d columns: ceil(d / world); the table is padded with zero columns to
        world * slab_width (the reference's default embed_dim = 300 on 8 GPUs: 38 columns per rank, 304 in all)."""
        return -(-int(d) // int(world))

    def set_values(self, val: torch.Tensor):
        """New attention values (an update_att): the CSC copy and the parts' copies follow."""
        self.val = val
        g = self.graph
        self.val_t = self.permute(val, g.t_perm) if g.t_perm is not None else None
        for key, (parts, _) in list(self._parts.items()):
            self._parts[key] = (parts, [self.permute(val, p.perm) for p in parts])

    # ------------------------------------------------------------------ parts of the structure by source rows
    def chunk_bounds(self, n_chunks: int) -> List[List[int]]:
        """[i][q] .. [i][q + 1]: row sub-range q of rank i's block (global rows), n_chunks of them per block."""
        out = []
        for i in range(self.world):
            lo, n = self.cuts[i], self.rows[i]
            out.append([lo + n * q // n_chunks for q in range(n_chunks + 1)])
        return out

    @staticmethod
    def default_chunks(world: int) -> int:
        """Batches of the incoming exchange: xGMI is a point-to-point mesh, ONE link per peer, so the exchange takes the time
        of one block over one link whatever the grouping by PEER -- what pipelines is a cut of EVERY peer's block into row
        sub-ranges: batch q moves sub-range q of all G - 1 blocks at once (all links busy, 1 / n_chunks of the time each)."""
        return 1 if world < 2 else 3

    def parts(self, transposed: bool, n_chunks: Optional[int] = None):
        """(n_chunks, StructureParts, their values): the CSR (transposed: the CSC) cut by the ROWS its entries gather from --
        part 0 = the rank's own block (no transfer), part q + 1 = row sub-range q of every other rank's block; built once
        per (direction, n_chunks) and kept."""
        n_chunks = self.default_chunks(self.world) if n_chunks is None else max(1, int(n_chunks))
        key = (bool(transposed), n_chunks)
        if key not in self._parts:
            bounds = self.chunk_bounds(n_chunks)
            cuts, part_of = [0], []
            for i in range(self.world):
                for q in range(n_chunks):
                    cuts.append(bounds[i][q + 1])
                    part_of.append(0 if i == self.rank else q + 1)
            parts = self.graph.structure_parts(transposed, cuts, part_of)
            while len(parts) < n_chunks + 1:          # (one rank: no remote part at all)
                parts.append(None)
            parts = [p for p in parts if p is not None]
            self._parts[key] = (parts, [self.permute(self.val, p.perm) for p in parts])
        parts, vals = self._parts[key]
        return n_chunks, parts, vals

    def head_parts(self, n_chunks: Optional[int] = None):
        return self.parts(True, n_chunks)

    # ------------------------------------------------------------------ the SpMM alone (no communication)
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

    # ------------------------------------------------------------------ SpMM pipelined with its exchanges
    def exchange_aggregate(self, transposed: bool, slab: Optional[torch.Tensor] = None,
                           block_in: Optional[torch.Tensor] = None, plus_self: bool = False, exchange_out: bool = True,
                           n_batches: Optional[int] = None, pieces: Optional[int] = None,
                           side_slab: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """side = A x (transposed: A^T x) on this rank's columns with both layout exchanges folded in.
          input   ``slab`` [N, D/G] (already a column slab: one whole-structure pass), or ``block_in`` [G, rows_r, D/G]
                  (this rank's rows as panels: they leave in n_batches row sub-ranges and the SpMM runs part by part behind
                  them);
          output  exchange_out: (side_slab, row block [G, rows_r, D/G]) -- the last part leaves owner range by owner range
                  (each range in ``pieces`` pieces); otherwise side_slab [N, D/G] alone.
        plus_self: side = x + A x (the layer's ``ego + side``, model.py:109).
        Every transfer and side stream of the pass belongs to one ``InFlight``: the pass returns -- or raises -- with all of
        them waited for and joined."""
        g, r, G = self.graph, self.rank, self.world
        if (slab is None) == (block_in is None):
            raise ValueError("exchange_aggregate: exactly one of slab / block_in")
        src = slab if slab is not None else block_in
        dev, dtype, dg = src.device, src.dtype, int(src.shape[-1])     # (any width: the structure's parts do not depend on it)
        whole = _Whole(g, transposed)
        if block_in is not None and G > 1 and (tuple(block_in.shape) != (G, self.my_rows, dg) or not block_in.is_contiguous()):
            raise ValueError(f"row block of shape {tuple(block_in.shape)}, expected a contiguous {(G, self.my_rows, dg)}")
        with InFlight(dev) as fl:
            if block_in is not None and G > 1:
                n_chunks, parts, vals = self.parts(transposed, n_batches)
                bounds = self.chunk_bounds(n_chunks)
                slab = torch.empty((g.n, dg), dtype=dtype, device=dev)
                slab[self.cuts[r]:self.cuts[r + 1]].copy_(block_in[r])               # the own block: no transfer
                queued = [None]                  # (part 0 = the own block waits for nothing)
                for q in range(n_chunks):        # ALL batches are queued now, in order, on the collective library's stream
                    sends, recvs = [], []
                    mlo, mhi = bounds[r][q] - self.cuts[r], bounds[r][q + 1] - self.cuts[r]      # sub-range q of MY rows
                    for k in range(1, G):
                        j, i = (r + k) % G, (r - k) % G
                        if mhi > mlo:
                            sends.append((block_in[j, mlo:mhi], j))
                        if bounds[i][q + 1] > bounds[i][q]:
                            recvs.append((slab[bounds[i][q]:bounds[i][q + 1]], i))
                    queued.append(fl.add(self.tp.p2p(sends, recvs)))
                assert len(queued) == len(parts)
            else:
                if block_in is not None:         # one rank: the row block IS the slab
                    slab = block_in[0]
                parts = [whole]
                vals = [self.val_t if transposed else self.val]
                queued = [None]
            if side_slab is None:
                side_slab = torch.empty((g.n, dg), dtype=dtype, device=dev)
            live = [b for b, p in enumerate(parts) if p.nnz > 0] or [0]
            first = True

            def arrived(b):
                if queued[b] is not None:
                    queued[b].wait()
                    queued[b] = None

            for b in range(len(parts)):
                arrived(b)
                if b not in live:
                    continue
                part, val_p = parts[b], vals[b]
                if b == live[-1]:                # the last pass: x itself is added here (plus_self), when ALL of it has arrived
                    for later in range(b + 1, len(parts)):
                        arrived(later)
                add_self = slab if (plus_self and b == live[-1]) else None
                if b == live[-1] and exchange_out and G > 1:
                    out = self._ranges_to_row_block(fl, part, val_p, slab, side_slab, first, add_self, out, pieces)
                else:
                    self.spmm(part.rowptr, part.col, val_p, slab, g.n, out=side_slab, long_rows=part.long_rows(),
                              add_self=add_self, add2=None if first else side_slab)
                first = False
        if not exchange_out:
            return side_slab
        if G == 1:
            if out is None:
                out = torch.empty((1, self.my_rows, dg), dtype=dtype, device=dev)
            out[0].copy_(side_slab)
        return side_slab, out

    def _ranges_to_row_block(self, fl: InFlight, part, val_p, slab, side_slab, first, add_self, out, pieces):
        """The last SpMM pass of ``exchange_aggregate`` fused with the slab -> row block exchange: the pass runs owner
        range by owner range, and as soon as a piece of the rows owned by rank j is done it leaves for rank j
        (point-to-point, the collective library's stream, one xGMI link per peer) while the next piece is being aggregated.
        Every peer's range is cut into `pieces` pieces visited peer-major inside a piece index, so all links are busy from
        the first round on and only the LAST piece of one peer (1 / (G * pieces) of the traffic) is exposed after the
        SpMM ends.  Piece p of round k computes rows of rank (rank + k) % G and receives from rank (rank - k) % G.

        On the GPU the row-range launches alternate between two side streams: back to back on ONE stream every extra
        launch costs ~25 us of tail and gap (5 M rows x 100 M entries x 32 columns: 1 launch 2.16 ms, 8: 2.30,
        32: 2.94), on two alternating streams the tail of one overlaps the start of the next (32 launches: 2.25 ms)
        and each piece still completes -- and leaves -- in order.  The streams and the transfers are ``fl``'s: the caller's
        ``InFlight`` joins and waits them, on the normal path and on an exception alike."""
        r, G = self.rank, self.world
        if pieces is None:
            pieces = 4
        dg = int(slab.shape[1])
        if out is None:
            out = torch.empty((G, self.my_rows, dg), dtype=slab.dtype, device=slab.device)

        def piece(lo, hi, p):      # p-th of `pieces` sub-ranges of [lo, hi)
            n = hi - lo
            return lo + n * p // pieces, lo + n * (p + 1) // pieces

        streams = []
        if slab.is_cuda:
            if not hasattr(self, "_side_streams"):
                self._side_streams = [torch.cuda.Stream(device=slab.device) for _ in range(2)]
            streams = fl.fork(self._side_streams)
        step = 0

        def launch(lo, hi):
            nonlocal step
            st = streams[step % len(streams)] if streams else None
            step += 1
            with fl.on(st):
                if hi > lo:
                    self.spmm(part.rowptr[lo:hi + 1], part.col, val_p, slab, hi - lo, out=side_slab[lo:hi],
                              long_rows=part.long_rows(lo, hi), add_self=add_self[lo:hi] if add_self is not None else None,
                              add2=None if first else side_slab[lo:hi])
            return st

        # The OTHER ranks' rows first, in `pieces` rounds: round p computes piece p of every other rank's range and then hands
        # the round's G - 1 sends and receives to the collective library in ONE group (all links busy; 4 groups per pass
        # instead of 28: the host has to stay ahead of 2.3 ms of launches).  This rank's own rows LAST, in one launch: they need
        # no transfer, so the last round (1 / pieces of a block per link) travels behind 1 / G of the pass instead of nothing.
        for p in range(pieces):
            sends, recvs = [], []
            for k in range(1, G):
                j, i = (r + k) % G, (r - k) % G
                lo, hi = piece(self.cuts[j], self.cuts[j + 1], p)
                launch(lo, hi)
                rlo, rhi = piece(0, self.my_rows, p)              # the matching piece of MY rows, arriving from rank i
                if hi > lo:
                    sends.append((side_slab[lo:hi], j))
                if rhi > rlo:
                    recvs.append((out[i, rlo:rhi], i))
            if not (sends or recvs):
                continue
            if streams:                       # the round's pieces sit on both compute streams: the group goes behind both
                streams[0].wait_stream(streams[1])
            with fl.on(streams[0] if streams else None):   # the library orders its transfers behind the CURRENT stream: this round
                fl.add(self.tp.p2p(sends, recvs))
        last = launch(self.cuts[r], self.cuts[r + 1])
        with fl.on(last):
            out[r].copy_(side_slab[self.cuts[r]:self.cuts[r + 1]])
        fl.join()
        return out

    def forward_to_row_block(self, slab: torch.Tensor, side_slab: Optional[torch.Tensor] = None,
                             out: Optional[torch.Tensor] = None, pieces: Optional[int] = None):
        """forward() fused with to_row_block(): (side_slab [N, D/G], row block [G, rows_r, D/G]); the slab -> row block
        exchange hidden behind the SpMM (``_ranges_to_row_block``)."""
        if self.world == 1:
            side_slab = self.forward(slab, out=side_slab)
            return side_slab, self.to_row_block(side_slab, out=out)
        return self.exchange_aggregate(False, slab=slab, exchange_out=True, pieces=pieces, side_slab=side_slab, out=out)

    def backward_from_row_block(self, block: torch.Tensor, out: Optional[torch.Tensor] = None,
                                pieces: Optional[int] = None) -> torch.Tensor:
        """to_column_slab() fused with backward(): grad_ego[:, my columns] from the gradient row block [G, rows_r, D/G].
        Slabs of >= 64 columns go back in COLUMN pieces (every column of A^T g is independent): all exchanges queued at
        once on the collective's stream, the transpose SpMM of piece p runs while piece p + 1 is still on the links.  A
        piece is never narrower than 32 columns (128-byte gathers; narrower rows cost the same time per entry), so
        narrower slabs (N = 8: D/G = 32) go back in head-range batches instead (``backward_in_head_parts``)."""
        g = self.graph
        if pieces is None:
            pieces = max(1, min(4, self.dg // 32))
        while pieces > 1 and self.dg % pieces:
            pieces -= 1
        if out is None:
            out = torch.empty((g.n, self.dg), dtype=block.dtype, device=block.device)
        if self.world == 1:
            return self.backward(self.to_column_slab(block), out=out)
        if pieces == 1:
            return self.backward_in_head_parts(block, out=out)
        w = self.dg // pieces
        with InFlight(block.device) as fl:
            queue = []
            for p in range(pieces):              # every piece's exchange is queued now; piece p's SpMM runs behind piece p + 1's
                src = block[:, :, p * w:(p + 1) * w].contiguous().view(self.world * self.my_rows, w)
                dst = torch.empty((g.n, w), dtype=block.dtype, device=block.device)
                queue.append((fl.add(self.tp.all_to_all(dst, src, self.rows, [self.my_rows] * self.world)), src))
            for p, (pend, _src) in enumerate(queue):
                dst = pend.wait()
                self.spmm(g.t_rowptr, g.t_col, self.val_t, dst, g.n, out=out[:, p * w:(p + 1) * w],
                          long_rows=g.long_rows(True))
        return out

    def backward_in_head_parts(self, block: torch.Tensor, out: Optional[torch.Tensor] = None,
                               n_batches: Optional[int] = None) -> torch.Tensor:
        """to_column_slab() fused with backward() when the slab is too narrow to cut by columns (N = 8: D/G = 32).
        A^T g = sum_p A_p^T g: part 0 = the entries whose HEAD lies in this rank's own rows (the SpMM starts at once),
        part q + 1 = those whose head lies in row sub-range q of the other ranks' blocks; the blocks arrive sub-range by
        sub-range (every batch uses all links; all batches queued at once on the collective's stream) and the transpose
        SpMM of part q + 1 -- a launch of the same kernel over the sub-CSC of the part, accumulating onto the sum of the
        parts before it (``add2``) -- runs while batch q + 1 is on the links.  Exposed: what the own-block part does not
        cover of the first batch (1 / n_chunks of the exchange); price: one more read + write of the N x D/G result per
        extra part.  Returns grad_ego[:, my columns]."""
        return self.exchange_aggregate(True, block_in=block, exchange_out=False, n_batches=n_batches, side_slab=out)

    # ------------------------------------------------------------------ the plain exchanges
    def to_row_block(self, slab: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """N x D/G column slab -> this rank's rows as G column panels, shape [G, rows_r, D/G]
        (panel i = columns of rank i).  One all-to-all."""
        if out is None:
            out = torch.empty((self.world, self.my_rows, self.dg), dtype=slab.dtype, device=slab.device)
        if self.world == 1:
            out[0].copy_(slab)
            return out
        self.tp.all_to_all(out.view(self.world * self.my_rows, self.dg), slab, [self.my_rows] * self.world, self.rows).wait()
        return out

    def to_column_slab(self, block: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[G, rows_r, D/G] panels of this rank's rows -> the N x D/G slab of this rank's columns."""
        if out is None:
            out = torch.empty((self.graph.n, self.dg), dtype=block.dtype, device=block.device)
        if self.world == 1:
            out.copy_(block[0])
            return out
        self.tp.all_to_all(out, block.view(self.world * self.my_rows, self.dg), self.rows, [self.my_rows] * self.world).wait()
        return out
