"""Graph ingestion: "h r t" text -> triples -> KG structure -> the loader's initial A_in.

Replaces, for the hot path's inputs only, DataLoader.load_graph / construct_data / create_adjacency_dict /
create_laplacian_dict (dataloader.py:186-190, 369-424, 449-495): pandas(engine='python') + iterrows() +
per-relation scipy matrices become one mmap'd C++ parse, an index sort and a pass over the sorted CSR.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from . import _native as N
from .graph import KGStructure


def load_triples(path: str, drop_duplicates: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(h, r, t) int64 arrays in file order, duplicate rows dropped (first kept) like dataloader.py:189."""
    cnt = np.zeros(1, np.int64)
    N.call("lkg_triples_count", path.encode(), N.ptr(cnt))
    n = int(cnt[0])
    h, r, t = (np.empty(n, np.int64) for _ in range(3))
    got = np.zeros(1, np.int64)
    N.call("lkg_triples_read", path.encode(), n, N.ptr(h), N.ptr(r), N.ptr(t), N.ptr(got))
    assert int(got[0]) == n
    if drop_duplicates and n:
        keep = np.empty(n, np.int64)
        nk = np.zeros(1, np.int64)
        N.call("lkg_triples_dedup", n, N.ptr(h), N.ptr(r), N.ptr(t), N.ptr(keep), N.ptr(nk))
        keep = keep[:int(nk[0])]
        if len(keep) != n:
            h, r, t = h[keep], r[keep], t[keep]
    return h, r, t


def laplacian_values(g: KGStructure, kind: str = "random-walk") -> torch.Tensor:
    """fp32[nnz] initial attention values in the structure's entry order, on the structure's device: a structure in HBM is
    processed there (lkg_laplacian_device_f32: two wave-per-row passes over the arrays the device build left), a host
    structure by the host form; the two agree bit for bit."""
    kinds = {"random-walk": 0, "symmetric": 1}
    if kind not in kinds:
        raise NotImplementedError(kind)          # dataloader.py:484
    if g.device.type == "cuda":
        val = torch.zeros(g.nnz, dtype=torch.float32, device=g.device)
        if g.nnz == 0:
            return val
        n_rel = int(g.rel.max()) + 1
        deg = torch.empty(g.n * n_rel, dtype=torch.int32, device=g.device)
        with torch.cuda.device(g.device):
            N.call("lkg_laplacian_device_f32", g.n, g.nnz, n_rel, N.ptr(g.rowptr), N.ptr(g.col), N.ptr(g.eptr), N.ptr(g.rel),
                   kinds[kind], N.ptr(deg), N.ptr(val), torch.cuda.current_stream(g.device).cuda_stream)
        return val
    val = np.zeros(g.nnz, np.float32)
    N.call("lkg_laplacian_f32", g.n, g.n_raw, g.nnz, N.ptr(g.host("rowptr")), N.ptr(g.host("col")),
           N.ptr(g.host("eptr")), N.ptr(g.host("rel")), kinds[kind], N.ptr(val))
    return torch.from_numpy(val)


def initial_a_in(n_entities: int, h, t, r, kind: str = "random-walk", device=None) -> torch.Tensor:
    """The sparse COO N x N tensor DataLoader.A_in holds (dataloader.py:494-495): coalesced, int64 indices.  ``device``: build
    the structure and the values there (a GPU: radix-sort build + device Laplacian, milliseconds at 10 M triples) and return
    the tensor on that device; default: on the host, like the reference's loader."""
    g = KGStructure.from_triples(n_entities, h, t, r, device=device, with_transpose=False)
    idx, val = g.coo_indices(), laplacian_values(g, kind)
    if kind == "symmetric":
        # a tail without out-edges under relation r has d_r(t)^-1/2 = inf -> 0 (dataloader.py:467-468) and
        # scipy's diagonal products do not store those entries: the reference's pattern omits them
        keep = val != 0
        idx, val = idx[:, keep], val[keep]
    return torch.sparse_coo_tensor(idx, val, (n_entities, n_entities), is_coalesced=True)
