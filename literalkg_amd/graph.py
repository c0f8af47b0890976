"""KG structure in HBM: (head, tail)-sorted CSR with merged duplicate pairs + its CSC.

Built once per edge list -- on the DEVICE when the structure lives on a GPU (``lkg_csr_build_device`` /
``lkg_csr_transpose_device``: a hand-written radix sort, milliseconds at 10 M triples), on the host otherwise
(``lkg_csr_build`` / ``lkg_csr_transpose``; CPU tests, gloo rehearsals); both give the same arrays bit for bit.
It replaces the sparse COO ``A_in`` *pattern* of the reference (model.py:257-261, 462-468); the attention VALUES stay
a flat fp32 array aligned with ``col``.

Layout (E_raw raw triples, nnz <= E_raw stored entries, N entities), all int32:
    rowptr  [N+1]   entries of head row h are rowptr[h]..rowptr[h+1]
    col     [nnz]   tail of each entry, ascending inside a row  (== coalesced COO order)
    eptr    [nnz+1] entry j covers sorted raw edges eptr[j]..eptr[j+1]   (None when nnz == E_raw)
    rel     [E_raw] relation of each sorted raw edge
    rel_first [nnz] relation of the first raw edge of each entry (only with eptr)
    t_rowptr/t_col/t_perm  the CSC: for tail t the heads pointing at it, and the CSR entry id of each
Host mirrors (numpy) of any array are made on demand by ``host(name)`` and cached.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _native as N


LONG_ROW_THRESHOLD = 256   # rows above this get a whole workgroup in the SpMM (lkg_spmm_csr_f32)

_ARRAYS = ("rowptr", "col", "eptr", "rel", "rel_first", "dup_entries", "dup_rows", "t_rowptr", "t_col", "t_perm")


class StructurePart:
    """A subset of the entries of a CSR (or of the CSC) as a CSR of its own over the same rows
    (KGStructure.structure_parts): ``perm`` = id of each entry in the structure's CSR value array."""

    def __init__(self, rowptr, col, perm, n_rows):
        self.rowptr, self.col, self.perm, self.n = rowptr, col, perm, int(n_rows)
        self.nnz = int(col.numel())
        self._long = {}

    def long_rows(self, lo: int = 0, hi: Optional[int] = None):
        """Rows of [lo, hi) (relative to lo) with more than LONG_ROW_THRESHOLD entries, or None."""
        hi = self.n if hi is None else hi
        if (lo, hi) not in self._long:
            rows = torch.nonzero((self.rowptr[lo + 1:hi + 1] - self.rowptr[lo:hi]) > LONG_ROW_THRESHOLD, as_tuple=True)[0].int()
            self._long[(lo, hi)] = rows if rows.numel() else None
        return self._long[(lo, hi)]


class KGStructure:
    def __init__(self):
        self.n = 0
        self.nnz = 0
        self.n_raw = 0
        for k in _ARRAYS:
            setattr(self, k, None)
        self._order = None          # sorted raw edge k is input edge order[k] (device int32 or host int64)
        self.device = torch.device("cpu")
        self._host = {}
        self._coo = None
        self._long = {}

    # ------------------------------------------------------------------ build
    @classmethod
    def from_triples(cls, n_entities: int, h, t, r=None, device=None, with_transpose: bool = True) -> "KGStructure":
        """h, t, r: 1-D integer tensors / arrays of equal length (any device); r=None -> relation 0."""
        device = torch.device(device if device is not None else "cpu")
        e = int(h.shape[0])
        if int(t.shape[0]) != e or (r is not None and int(r.shape[0]) != e):
            raise ValueError("h, t, r must have equal lengths")
        if device.type == "cuda":
            return cls._build_device(int(n_entities), h, t, r, device, with_transpose)
        return cls._build_host(int(n_entities), h, t, r, device, with_transpose)

    @classmethod
    def _build_host(cls, n, h, t, r, device, with_transpose):
        def host(x):
            if x is None:
                return None
            if isinstance(x, torch.Tensor):
                x = x.detach().cpu().numpy()
            return np.ascontiguousarray(x, dtype=np.int64)
        hh, tt, rr = host(h), host(t), host(r)
        e = int(hh.shape[0])
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
        g._order = order[:e]
        host_arrays = dict(rowptr=rowptr, col=col[:g.nnz], rel=rel[:e])
        if g.nnz != e:   # stored entries covering several raw edges (the same (h,t) under several relations)
            ep = eptr[:g.nnz + 1]
            de = np.flatnonzero(np.diff(ep) > 1).astype(np.int32)
            host_arrays.update(eptr=ep, rel_first=rel[:e][ep[:-1]], dup_entries=de,
                               dup_rows=(np.searchsorted(rowptr, de, side="right") - 1).astype(np.int32))
        if with_transpose:
            t_rowptr = np.empty(n + 1, np.int32)
            t_col = np.empty(max(g.nnz, 1), np.int32)
            t_perm = np.empty(max(g.nnz, 1), np.int32)
            N.call("lkg_csr_transpose", n, n, g.nnz, N.ptr(rowptr), N.ptr(col), N.ptr(t_rowptr), N.ptr(t_col),
                   N.ptr(t_perm))
            host_arrays.update(t_rowptr=t_rowptr, t_col=t_col[:g.nnz], t_perm=t_perm[:g.nnz])
        g._host = host_arrays
        for k, v in host_arrays.items():
            setattr(g, k, torch.from_numpy(v).to(device))
        g.device = device
        return g

    @classmethod
    def _build_device(cls, n, h, t, r, device, with_transpose):
        def dev(x):
            if x is None:
                return None
            x = torch.as_tensor(x)
            return x.to(device=device, dtype=torch.int64).contiguous()
        hh, tt, rr = dev(h), dev(t), dev(r)
        e = int(hh.shape[0])
        stream = torch.cuda.current_stream(device).cuda_stream
        i32 = dict(dtype=torch.int32, device=device)
        rowptr = torch.empty(n + 1, **i32)
        col = torch.empty(max(e, 1), **i32)
        eptr = torch.empty(e + 1, **i32)
        rel = torch.empty(max(e, 1), **i32)
        order = torch.empty(max(e, 1), **i32)
        counts = torch.empty(2, dtype=torch.int64, device=device)
        ws_bytes = N.load().lkg_csr_build_device_workspace(n, e)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            N.call("lkg_csr_build_device", n, e, N.ptr(hh), N.ptr(tt), N.ptr(rr), N.ptr(rowptr), N.ptr(col),
                   N.ptr(eptr), N.ptr(rel), N.ptr(order), N.ptr(counts), N.ptr(ws), ws_bytes, stream)
            nnz, n_bad = (int(x) for x in counts.tolist())          # the one host sync of a build
            if n_bad:
                raise N.LkgError(f"lkg_csr_build_device: {n_bad} triple(s) with an entity id outside [0, {n}) or a "
                                 f"relation id outside int32")
            g = cls()
            g.n, g.nnz, g.n_raw, g.device = n, nnz, e, device
            g._order = order[:e]
            g.rowptr, g.col, g.rel = rowptr, col[:nnz], rel[:e]
            if nnz != e:
                ep = eptr[:nnz + 1]
                de = torch.nonzero((ep[1:] - ep[:-1]) > 1, as_tuple=True)[0].int()
                g.eptr, g.rel_first, g.dup_entries = ep, rel[ep[:-1].long()], de
                g.dup_rows = (torch.searchsorted(rowptr, de, right=True) - 1).int()
            if with_transpose:
                t_rowptr = torch.empty(n + 1, **i32)
                t_col = torch.empty(max(nnz, 1), **i32)
                t_perm = torch.empty(max(nnz, 1), **i32)
                ws_t = N.load().lkg_csr_transpose_device_workspace(n, nnz)
                if ws_t > ws_bytes:
                    ws = torch.empty(ws_t, dtype=torch.uint8, device=device)
                N.call("lkg_csr_transpose_device", n, n, nnz, N.ptr(rowptr), N.ptr(col), N.ptr(t_rowptr),
                       N.ptr(t_col), N.ptr(t_perm), N.ptr(ws), ws.numel(), stream)
                g.t_rowptr, g.t_col, g.t_perm = t_rowptr, t_col[:nnz], t_perm[:nnz]
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
        for k in _ARRAYS:
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, v.to(device))
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
            rp = self.t_rowptr if transposed else self.rowptr
            rows = torch.nonzero((rp[lo + 1:hi + 1] - rp[lo:hi]) > LONG_ROW_THRESHOLD, as_tuple=True)[0].int()
            self._long[key] = rows if rows.numel() else None
        return self._long[key]

    EMPTY_ROWS_LISTED_FROM = 0.5      # a structure with at least this share of empty rows is aggregated row list by row list

    def row_lists(self, transposed: bool = False):
        """(rows with entries, rows without) as device int32 lists when at least EMPTY_ROWS_LISTED_FROM of the CSR's
        (transposed: the CSC's) rows are empty, else None -- the reference's id spaces are sparse (data/Small: 765 957 entity
        rows for 125 422 used ids, dataloader.py:405-418) and one wave per EMPTY row leaves the SpMM bound by the rate at which
        workgroups start.  Computed once per structure."""
        key = ("rows", bool(transposed))
        if key not in self._long:
            rp = self.t_rowptr if transposed else self.rowptr
            lists = None
            if rp is not None and rp.is_cuda and self.n >= 4096:
                has = rp[1:] > rp[:-1]
                n_with = int(has.sum())
                if self.n - n_with >= self.EMPTY_ROWS_LISTED_FROM * self.n:
                    lists = (torch.nonzero(has, as_tuple=True)[0].int(), torch.nonzero(~has, as_tuple=True)[0].int())
            self._long[key] = lists
        return self._long[key]

    def structure_parts(self, transposed: bool, cuts, part_of_block) -> list:
        """The CSR (or, transposed, the CSC) cut by the SOURCE of every entry -- the row it gathers from: its tail in the
        CSR, its head in the CSC -- into sub-structures: an entry goes to part ``part_of_block[b]`` where b is the block
        [cuts[b], cuts[b+1]) that holds its source.  Every part is a CSR of its own over all N rows (``StructurePart``:
        entries of a row in their original order, ``perm`` = CSR entry id for the values), so that  A x = sum_p A_p x  can
        run part by part, each part as soon as the rows of x it gathers from have arrived
        (sharding.FeatureShardedAggregation).  Built once per structure with torch index ops on the structure's device
        (index bookkeeping, like the frontier lists)."""
        if transposed and self.t_rowptr is None:
            raise RuntimeError("KGStructure was built without its transpose")
        rowptr, col = (self.t_rowptr, self.t_col) if transposed else (self.rowptr, self.col)
        dev = col.device
        cuts = torch.as_tensor([int(c) for c in cuts], dtype=torch.int64, device=dev)
        lookup = torch.as_tensor([int(x) for x in part_of_block], dtype=torch.int64, device=dev)
        if cuts.numel() != lookup.numel() + 1 or int(cuts[0]) != 0 or int(cuts[-1]) != self.n:
            raise ValueError("structure_parts: cuts must run from 0 to n with one part id per block")
        n_parts = int(lookup.max()) + 1 if lookup.numel() else 0
        part = lookup[torch.bucketize(col, cuts[1:-1].to(col.dtype), right=True)]
        counts = (rowptr[1:] - rowptr[:-1]).long()
        row = torch.repeat_interleave(torch.arange(self.n, device=dev, dtype=torch.int32), counts)
        parts = []
        for p in range(n_parts):
            idx = torch.nonzero(part == p, as_tuple=True)[0]
            rp = torch.zeros(self.n + 1, dtype=torch.int64, device=dev)
            if idx.numel():
                rp[1:] = torch.cumsum(torch.bincount(row[idx].long(), minlength=self.n), 0)
            perm = self.t_perm[idx].contiguous() if transposed else idx.int()
            parts.append(StructurePart(rp.int(), col[idx].contiguous(), perm, self.n))
        return parts

    # ------------------------------------------------------------------ views
    @property
    def has_dups(self) -> bool:
        return self.eptr is not None

    @property
    def order(self) -> np.ndarray:
        """int64 host array: sorted raw edge k is input edge order[k]."""
        if isinstance(self._order, torch.Tensor):
            self._order = self._order.cpu().numpy().astype(np.int64)
        return self._order

    def entry_rows(self) -> torch.Tensor:
        """int64[nnz] head of every stored entry."""
        counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
        return torch.repeat_interleave(torch.arange(self.n, device=self.device), counts)

    def coo_indices(self) -> torch.Tensor:
        """int64[2, nnz], sorted by (row, col): the indices of the reference's coalesced A_in."""
        if self._coo is None:
            if self.rowptr.is_cuda:           # one library kernel (no first-use loading of torch's repeat_interleave / stack)
                out = torch.empty((2, self.nnz), dtype=torch.int64, device=self.device)
                with torch.cuda.device(self.device):
                    N.call("lkg_csr_coo_indices_i64", self.n, self.nnz, N.ptr(self.rowptr), N.ptr(self.col), N.ptr(out),
                           torch.cuda.current_stream(self.device).cuda_stream)
                self._coo = out
            else:
                self._coo = torch.stack([self.entry_rows(), self.col.long()])
        return self._coo

    def row_cuts(self, n_parts: int) -> np.ndarray:
        """nnz-balanced contiguous head-row ranges."""
        cuts = np.empty(n_parts + 1, np.int64)
        N.call("lkg_row_partition", self.n, N.ptr(self.host("rowptr")), int(n_parts), N.ptr(cuts))
        return cuts

    def host(self, name: str):
        """numpy mirror of one of the arrays (None when the structure has no such array); copied once."""
        if name not in self._host:
            v = getattr(self, name)
            self._host[name] = None if v is None else np.ascontiguousarray(v.cpu().numpy())
        return self._host[name]
