"""Device-side KG batch sampler: DataLoader.generate_kg_batch + sample_pos_triples_for_head +
sample_neg_triples_for_head (dataloader.py:249-330) on the device structure.  With the encoder at a few
milliseconds per step, the reference's Python rejection sampler (and its per-iteration
``list(data.training_tails)``, main_pretraining.py:101) would be the step-time floor."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _native as N
from . import ops
from .graph import KGStructure


class KGBatchSampler:
    def __init__(self, graph: KGStructure, neg_rate: int):
        if graph.nnz == 0:
            raise ValueError("cannot sample from an empty graph")
        self.graph, self.neg_rate = graph, int(neg_rate)
        deg = graph.rowptr[1:] - graph.rowptr[:-1]
        self.heads = torch.nonzero(deg > 0, as_tuple=True)[0]        # entities with at least one triple (kg_dict keys)

    def sample(self, batch_size: int, heads: Optional[torch.Tensor] = None, seed: Optional[int] = None
               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """(h, r, pos_t, neg_t) int64 device tensors of batch_size // neg_rate groups x neg_rate entries, the
        layout generate_kg_batch returns.  ``heads`` restricts the draw to an epoch's sampled heads
        (epoch_sampling_data_dict, main_pretraining.py:93-96)."""
        ops._need_gpu(self.graph.rowptr)
        pool = self.heads if heads is None else heads.long()
        if heads is not None and pool.numel():
            # caller-supplied heads: the reference's kg_dict[h] raises KeyError for an entity without triples
            inside = (pool >= 0) & (pool < self.graph.n)
            deg = self.graph.rowptr[1:] - self.graph.rowptr[:-1]
            ok = inside & (deg[pool.clamp(0, self.graph.n - 1)] > 0)
            if not bool(ok.all()):
                raise KeyError(f"sample(heads=...): {int((~ok).sum())} head id(s) outside [0, {self.graph.n}) or "
                               f"without a triple, e.g. {int(pool[~ok][0])}")
        groups = int(batch_size / self.neg_rate)                     # dataloader.py:285
        if groups <= pool.numel():                                   # random.sample: without replacement
            chosen = pool[torch.randperm(pool.numel(), device=pool.device)[:groups]]
        else:                                                        # random.choice per slot
            chosen = pool[torch.randint(pool.numel(), (groups,), device=pool.device)]
        chosen = chosen.contiguous()
        seed = ops.new_seed() if seed is None else seed
        g = self.graph
        out = torch.empty((4, groups * self.neg_rate), dtype=torch.int64, device=chosen.device)
        N.call("lkg_sample_kg_batch", groups, self.neg_rate, int(seed), N.ptr(chosen), g.n, N.ptr(g.rowptr), N.ptr(g.col),
               N.ptr(g.eptr), N.ptr(g.rel), g.nnz, g.n_raw, N.ptr(out[0]), N.ptr(out[1]), N.ptr(out[2]),
               N.ptr(out[3]), ops._stream())
        return out[0], out[1], out[2], out[3]
