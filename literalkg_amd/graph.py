"""KG structure in HBM: (head, tail)-sorted CSR with merged duplicate pairs + its CSC.

Built once on the host by the C ABI (``lkg_csr_build`` / ``lkg_csr_transpose``) and kept on the
device as int32 arrays.  It replaces the sparse COO ``A_in`` *pattern* of the reference
(model.py:257-261, 462-468); the attention VALUES stay a flat fp32 array aligned with ``col``.

Layout (E_raw raw triples, nnz <= E_raw stored entries, N entities):
    rowptr  int32[N+1]   entries of head row h are rowptr[h]..rowptr[h+1]
    col     int32[nnz]   tail of each entry, ascending inside a row  (== coalesced COO order)
    eptr    int32[nnz+1] entry j covers sorted raw edges eptr[j]..eptr[j+1]   (None when nnz == E_raw)
    rel     int32[E_raw] relation of each sorted raw edge
    rel_first int32[nnz] relation of the first raw edge of each entry (only with eptr)
    t_rowptr/t_col/t_perm  the CSC: for tail t the heads pointing at it, and the CSR entry id of each
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _native as N


LONG_ROW_THRESHOLD = 256   # rows above this get a whole workgroup in the SpMM (lkg_spmm_csr_f32)


class KGStructure:
    def __init__(self):
        self.n = 0
        self.nnz = 0
        self.n_raw = 0
        self.rowptr = self.col = self.eptr = self.rel = self.rel_first = self.dup_entries = self.dup_rows = None
        self.t_rowptr = self.t_col = self.t_perm = None
        self.order = None           # int64 host: sorted raw edge k is input edge order[k]
        self.device = torch.device("cpu")
        self._coo = None
        self._long = {}

    # ------------------------------------------------------------------ build
    @classmethod
    def from_triples(cls, n_entities: int, h, t, r=None, device=None, with_transpose: bool = True) -> "KGStructure":
        """h, t, r: 1-D integer tensors / arrays of equal length (any device); r=None -> relation 0."""
        def host(x):
            if x is None:
                return None
            if isinstance(x, torch.Tensor):
                x = x.detach().cpu().numpy()
            return np.ascontiguousarray(x, dtype=np.int64)
        hh, tt, rr = host(h), host(t), host(r)
        e = int(hh.shape[0])
        if tt.shape[0] != e or (rr is not None and rr.shape[0] != e):
            raise ValueError("h, t, r must have equal lengths")
        n = int(n_entities)
        rowptr = np.empty(n + 1, np.int32)
        col = np.empty(max(e, 1), np.int32)
        eptr = np.empty(e + 1, np.int32)
        rel = np.empty(max(e, 1), np.int32)
        order = np.empty(max(e, 1), np.int64)
        nnz = np.zeros(1, np.int64)
        N.call("lkg_csr_build", n, e, N.ptr(hh), N.ptr(tt), N.ptr(rr), N.ptr(rowptr), N.ptr(col), N.ptr(eptr),
               N.ptr(rel), N.ptr(order), N.ptr(nnz))
        g = cls()
        g.n, g.nnz, g.n_raw = n, int(nnz[0]), e
        g.order = order[:e]
        dups = g.nnz != e
        g._host = dict(rowptr=rowptr, col=col[:g.nnz], rel=rel[:e], eptr=None, rel_first=None, dup_entries=None,
                       dup_rows=None)
        if dups:   # stored entries covering several raw edges (the same (h,t) under several relations)
            ep = eptr[:g.nnz + 1]
            de = np.flatnonzero(np.diff(ep) > 1).astype(np.int32)
            g._host.update(eptr=ep, rel_first=rel[:e][ep[:-1]], dup_entries=de,
                           dup_rows=(np.searchsorted(rowptr, de, side="right") - 1).astype(np.int32))
        if with_transpose:
            t_rowptr = np.empty(n + 1, np.int32)
            t_col = np.empty(max(g.nnz, 1), np.int32)
            t_perm = np.empty(max(g.nnz, 1), np.int32)
            N.call("lkg_csr_transpose", n, n, g.nnz, N.ptr(rowptr), N.ptr(col), N.ptr(t_rowptr), N.ptr(t_col),
                   N.ptr(t_perm))
            g._host.update(t_rowptr=t_rowptr, t_col=t_col[:g.nnz], t_perm=t_perm[:g.nnz])
        g.to(device if device is not None else "cpu")
        return g

    @classmethod
    def from_coo(cls, a_in: torch.Tensor, device=None) -> "KGStructure":
        """From a sparse COO N x N matrix (the loader's Laplacian A_in, dataloader.py:494-495).
        The matrix is coalesced first; its value order then equals this structure's entry order."""
        a = a_in if a_in.is_coalesced() else a_in.coalesce()
        idx = a.indices()
        g = cls.from_triples(a.shape[0], idx[0], idx[1], None, device)
        if g.nnz != idx.shape[1] or not np.array_equal(g.order, np.arange(g.nnz)):
            raise AssertionError("coalesced COO is expected to be sorted by (row, col) without duplicates")
        return g

    def to(self, device) -> "KGStructure":
        device = torch.device(device)
        for k, v in self._host.items():
            setattr(self, k, None if v is None else torch.from_numpy(v).to(device))
        if "t_rowptr" not in self._host:
            self.t_rowptr = self.t_col = self.t_perm = None
        self.device = device
        self._coo = None
        self._long = {}
        return self

    def long_rows(self, transposed: bool = False, lo: int = 0, hi: Optional[int] = None):
        """Device int32 list of the rows in [lo, hi) (relative to lo) with more than LONG_ROW_THRESHOLD
        entries, for the CSR (transposed=False) or the CSC; None when there are none."""
        hi = self.n if hi is None else hi
        key = (transposed, lo, hi)
        if key not in self._long:
            rp = self._host["t_rowptr" if transposed else "rowptr"]
            rows = np.flatnonzero(np.diff(rp[lo:hi + 1]) > LONG_ROW_THRESHOLD).astype(np.int32)
            self._long[key] = torch.from_numpy(rows).to(self.device) if len(rows) else None
        return self._long[key]

    # ------------------------------------------------------------------ views
    @property
    def has_dups(self) -> bool:
        return self.eptr is not None

    def entry_rows(self) -> torch.Tensor:
        """int64[nnz] head of every stored entry."""
        counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
        return torch.repeat_interleave(torch.arange(self.n, device=self.device), counts)

    def coo_indices(self) -> torch.Tensor:
        """int64[2, nnz], sorted by (row, col): the indices of the reference's coalesced A_in."""
        if self._coo is None:
            self._coo = torch.stack([self.entry_rows(), self.col.long()])
        return self._coo

    def row_cuts(self, n_parts: int) -> np.ndarray:
        """nnz-balanced contiguous head-row ranges."""
        cuts = np.empty(n_parts + 1, np.int64)
        N.call("lkg_row_partition", self.n, N.ptr(self._host["rowptr"]), int(n_parts), N.ptr(cuts))
        return cuts

    def host(self, name: str):
        return self._host[name]
