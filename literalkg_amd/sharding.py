"""Row-range ("edge-range") sharding of the aggregation hot path across the GPUs of one node
(SURVEY.md 8e).  One process per GPU; collectives go through torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; the same code runs under "gloo" for the CPU rehearsal tests).

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
