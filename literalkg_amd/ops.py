"""torch.autograd.Function wrappers over the C ABI (include/literalkg_hip.h).

Every op launches on ``torch.cuda.current_stream()`` and REQUIRES device tensors: there is no CPU
path here (the CPU restatement lives under ``oracle/`` and is test infrastructure only).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import _native as N
from .graph import KGStructure, LONG_ROW_THRESHOLD

LEAKY_SLOPE = 0.01
LN_EPS = 1e-5
NORMALIZE_EPS = 1e-12


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream (the raw getter is ~10x cheaper than building a Stream object)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "literalkg_amd ops run only on an MI355X device tensor (got a CPU tensor); there is no CPU "
                "fallback -- move the model and its inputs to the GPU (model.to('cuda')).")


def _f32_rows(t: torch.Tensor) -> torch.Tensor:
    """fp32 2-D tensor with unit column stride (row stride arbitrary)."""
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError(f"expected a 2-D tensor, got shape {tuple(t.shape)}")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def _i64(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.int64:
        t = t.long()
    return t.contiguous()


# ----------------------------------------------------------------------------- deferred device-side checks
class _Deferred:
    """Error conditions a kernel detects on the device (an id out of range) without stalling the step on a host
    sync: the kernel writes a counter, the counter is copied to pinned host memory behind it, and the NEXT op that
    polls (or ``check_deferred_errors()``, which waits) raises -- the way an asynchronous device-side assert of the
    reference's ``gat_trans_M[r]`` would surface one call late."""
    pending = []
    _lock = None

    @classmethod
    def _guard(cls):
        if cls._lock is None:
            import threading
            cls._lock = threading.Lock()
        return cls._lock

    @classmethod
    def watch(cls, counter: torch.Tensor, exc, message: str):
        host = torch.empty(counter.shape, dtype=counter.dtype, pin_memory=True)
        host.copy_(counter, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        with cls._guard():
            cls.pending.append((ev, host, exc, message))

    @classmethod
    def poll(cls, wait: bool = False):
        with cls._guard():                    # (replica threads share the list: each takes what is there and puts back the rest)
            mine, cls.pending = cls.pending, []
        keep, err = [], None
        for ev, host, exc, message in mine:
            if wait:
                ev.synchronize()
            if not ev.query():
                keep.append((ev, host, exc, message))
            elif int(host.sum()) != 0 and err is None:
                err = exc(message.format(n=int(host.sum())))
        if keep:
            with cls._guard():
                cls.pending = keep + cls.pending
        if err is not None:
            raise err


def check_deferred_errors():
    """Wait for the device and raise any pending deferred error (call before trusting a step's results)."""
    _Deferred.poll(wait=True)


def checked_ids(n_rows: int, *id_lists: torch.Tensor, what: str = "entity"):
    """The id lists a caller hands in (a batch's h / pos_t / neg_t, head / tail ids of the scoring heads) as int64 tensors
    that are SAFE to gather and scatter through: an id outside [0, n_rows) is replaced by row 0 on the device and
    counted; the count surfaces as an IndexError at the next poll (``check_deferred_errors()`` waits for it) -- the
    reference's embedding lookup raises for such an id, a kernel of this library would read or add out of bounds."""
    _need_gpu(*id_lists)
    shapes = [tuple(i.shape) for i in id_lists]
    flat = [_i64(i.reshape(-1)) for i in id_lists]
    src = flat[0] if len(flat) == 1 else torch.cat(flat)
    out = torch.empty_like(src)
    bad = torch.empty(1, dtype=torch.int32, device=src.device)
    N.call("lkg_sanitize_ids_i64", src.numel(), N.ptr(src), 0, int(n_rows), N.ptr(out), N.ptr(bad), _stream())
    _Deferred.poll()
    _Deferred.watch(bad, IndexError, "{n} " + what + f" id(s) outside [0, {int(n_rows)}) in the ids handed to the model "
                    "(replaced by row 0 on the device; the results of that call are meaningless)")
    res, o = [], 0
    for sh, f in zip(shapes, flat):
        res.append(out[o:o + f.numel()].view(sh))
        o += f.numel()
    return res


def count_ids_outside(n_rows: int, ids: torch.Tensor) -> int:
    """How many of ``ids`` lie outside [0, n_rows)?  (One pass of the id sanitiser; the answer is read on the host.)"""
    _need_gpu(ids)
    src = _i64(ids.reshape(-1))
    if src.numel() == 0:
        return 0
    out = torch.empty_like(src)
    bad = torch.empty(1, dtype=torch.int32, device=src.device)
    N.call("lkg_sanitize_ids_i64", src.numel(), N.ptr(src), 0, int(n_rows), N.ptr(out), N.ptr(bad), _stream())
    return int(bad.item())


# ----------------------------------------------------------------------------- raw launches
import os as _os

CHECK_STRUCTURES = _os.environ.get("LKG_CHECK_STRUCTURES", "0") not in ("", "0")     # validate every raw SpMM's structure first


def check_csr(rowptr: torch.Tensor, col: torch.Tensor, n_rows: int, n_cols: int, col_offset: int = 0,
              what: str = "structure") -> None:
    """Raise unless an SpMM over the CSR view ``rowptr[0 .. n_rows]`` / ``col`` stays inside ``col`` (and ``val`` of the same
    length) and inside a source table of ``n_cols`` rows handed over with row offset ``col_offset``: offsets ascending and
    within [0, col.numel()], column ids within the table (one streaming device pass + one host read: a hardening aid, run
    where a structure is built or -- LKG_CHECK_STRUCTURES=1 -- before every raw launch)."""
    _need_gpu(rowptr, col)
    if rowptr.dtype != torch.int32 or col.dtype != torch.int32 or not rowptr.is_contiguous() or not col.is_contiguous():
        raise TypeError(f"{what}: rowptr and col must be contiguous int32 tensors")
    if rowptr.numel() < n_rows + 1:
        raise ValueError(f"{what}: {rowptr.numel()} offsets for {n_rows} rows")
    bad = torch.empty(1, dtype=torch.int32, device=col.device)
    N.call("lkg_csr_check_i32", int(n_rows), N.ptr(rowptr), col.numel(), N.ptr(col), int(col_offset), int(n_cols),
           N.ptr(bad), _stream())
    n_bad = int(bad.item())
    if n_bad:
        raise N.LkgError(f"{what}: {n_bad} offset(s) / column id(s) would make the SpMM read outside its operands "
                         f"({n_rows} rows, {col.numel()} stored entries, a source table of {n_cols} rows at offset {col_offset})")


def spmm_raw(rowptr, col, val, x, n_rows: int, out: Optional[torch.Tensor] = None,
             x_row_offset: int = 0, long_rows: Optional[torch.Tensor] = None,
             add_self: Optional[torch.Tensor] = None, add2: Optional[torch.Tensor] = None,
             copy: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
             rowmax: Optional[torch.Tensor] = None, add2_rows: Optional[torch.Tensor] = None,
             x_rows: Optional[torch.Tensor] = None, self_rows: Optional[torch.Tensor] = None,
             out_rows: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
             row_lists: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """out[i,:] = (add_self[i,:] +) (add2[i,:] +) sum_j val[j] * x[col[j] - x_row_offset, :] for the n_rows rows
    described by rowptr (a view into a longer rowptr is fine: its values index col/val directly).  x_row_offset lets
    a row-range shard hand over only ITS rows of x while col keeps global ids.  copy = (src, dst): the kernel's
    epilogue also copies src[i,:] to dst[i,:] (a row copy riding along instead of a pass of its own).  add2_rows /
    x_rows / self_rows (uint8 per row of add2 / x / add_self): the operand is zero outside the flagged rows and is read
    there only (x_rows is indexed like x: by col - x_row_offset).  out_rows (uint8 per output row, with x_rows): receives
    1 where a row got a contribution; the other rows of ``out`` are NOT written (the caller keeps them zero).
    bias (float32[d], instead of add2): the same row added to every output row.  row_lists = (rows with entries, rows
    without), int32, of a structure whose rows are mostly empty (KGStructure.row_lists): one wave per listed row, the empty
    rows in one streaming pass; ignored with x_rows."""
    _need_gpu(x, val, rowptr, col)
    x = _f32_rows(x)
    d = x.shape[1]
    if out is None:
        out = torch.empty((n_rows, d), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (n_rows, d) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError(f"spmm_raw: out must be a float32 {n_rows} x {d} tensor with unit column stride (got {tuple(out.shape)})")
    if rowptr.numel() < n_rows + 1 or col.numel() != val.numel():
        raise ValueError("spmm_raw: rowptr needs n_rows + 1 offsets, col and val one element per stored entry")
    if rowptr.dtype != torch.int32 or col.dtype != torch.int32 or val.dtype != torch.float32:
        raise TypeError("spmm_raw: int32 offsets and column ids, float32 values")
    if CHECK_STRUCTURES:
        check_csr(rowptr, col, n_rows, x.shape[0], x_row_offset, what="spmm_raw")
    if bias is not None:
        if add2 is not None or bias.numel() != d or bias.dtype != torch.float32 or not bias.is_contiguous():
            raise ValueError(f"spmm_raw: bias must be a contiguous float32 vector of {d} elements and excludes add2")
    for name_, t_ in (("add_self", add_self), ("add2", add2)):
        if t_ is not None and tuple(t_.shape) != (n_rows, d):
            raise ValueError(f"spmm_raw: {name_} must be {n_rows} x {d} (got {tuple(t_.shape)})")
    if add_self is not None:
        add_self = _f32_rows(add_self)
    if add2 is not None:
        add2 = _f32_rows(add2)
    csrc, cdst = (_f32_rows(copy[0]), copy[1]) if copy is not None else (None, None)
    if cdst is not None and (cdst.stride(1) != 1 or cdst.shape != (n_rows, d)):
        raise ValueError("spmm_raw: copy destination must be an n_rows x d view with unit column stride")
    N.call("lkg_spmm_csr_fused_f32", n_rows, d, N.ptr(rowptr), N.ptr(col), N.ptr(val),
           N.ptr(x) - 4 * x_row_offset * _ld(x), _ld(x), N.ptr(out), _ld(out), N.ptr(add_self),
           _ld(add_self) if add_self is not None else 0, N.ptr(add2 if bias is None else bias),
           _ld(add2) if add2 is not None else 0, N.ptr(add2_rows if add2 is not None else None), N.ptr(csrc), _ld(csrc) if csrc is not None else 0, N.ptr(cdst), _ld(cdst) if cdst is not None else 0,
           N.ptr(rowmax), (x_rows.data_ptr() - x_row_offset) if x_rows is not None else None,
           N.ptr(self_rows if add_self is not None else None), N.ptr(out_rows), N.ptr(long_rows),
           0 if long_rows is None else long_rows.numel(), LONG_ROW_THRESHOLD,
           *((N.ptr(row_lists[0]), row_lists[0].numel(), N.ptr(row_lists[1]), row_lists[1].numel())
             if (row_lists is not None and x_rows is None and row_lists[0].numel() + row_lists[1].numel() == n_rows)
             else (None, 0, None, 0)), _stream())
    return out


def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False, alpha: float = 1.0,
         beta: float = 0.0, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = alpha * op(a) @ op(b) + beta * out (+ bias).  a, b: row-major 2-D fp32 (row stride free)."""
    a, b = _f32_rows(a), _f32_rows(b)
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    kb, n = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if k != kb:
        raise ValueError(f"gemm: inner dimensions differ ({k} vs {kb})")
    if out is None:
        if beta != 0.0:
            raise ValueError("gemm: beta != 0 needs out")
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    elif tuple(out.shape) != (m, n) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError(f"gemm: out must be a float32 {m} x {n} tensor with unit column stride (got {tuple(out.shape)})")
    if bias is not None and bias.numel() != n:
        raise ValueError(f"gemm: bias of {bias.numel()} elements for {n} output columns")
    if (not trans_a and alpha == 1.0 and _ENGINE != "f32" and (k * n <= 2048 or m < TALL_MIN_ROWS)
            and N.load().lkg_gemm_skinny_ok(m, n, k, N.ptr(a), _ld(a), N.ptr(out), _ld(out))):
        # narrow in AND out (the 32 x 32 products of narrow layers): exact f32 on the VALU at streaming rate
        N.call("lkg_gemm_skinny_f32", m, n, k, N.ptr(a), _ld(a), N.ptr(b), _ld(b), int(trans_b), float(beta), N.ptr(out),
               _ld(out), N.ptr(bias), _stream())
        return out
    if not trans_a and k > 0 and tall_ok(m, n, (k,), single_panel_too=tagged_rowmax(a) is not None):
        return gemm_tall((a,), ((b,),), bool(trans_b), bias, alpha, beta, out)
    if (trans_a and not trans_b and alpha == 1.0 and beta == 0.0 and bias is None and _WGRAD_ENGINE == "longk"
            and N.load().lkg_gemm_smallm_ok(m, n, k, N.ptr(a), _ld(a), N.ptr(b), _ld(b))):
        # a narrow dY (conv_dim 32 / 64) over many rows: exact f32 on the VALU instead of a mostly empty matrix-core tile
        N.call("lkg_gemm_smallm_f32", m, n, k, N.ptr(a), _ld(a), N.ptr(b), _ld(b), N.ptr(out), _ld(out), _stream())
        return out
    if (trans_a and not trans_b and alpha == 1.0 and beta == 0.0 and bias is None and _WGRAD_ENGINE == "longk"
            and N.load().lkg_gemm_longk_ok(m, n, k, N.ptr(a), _ld(a), N.ptr(b), _ld(b))):
        # weight gradients (a^T @ b over millions of rows): the 256 x 128 engine with three tiles in flight
        N.call("lkg_gemm_longk_f32", m, n, k, N.ptr(a), _ld(a), N.ptr(b), _ld(b), N.ptr(out), _ld(out), _stream())
        return out
    need = int(N.load().lkg_gemm_workspace(int(trans_a), m, n, k)) if _ENGINE != "f32" else 0
    ws = _workspace(need, out.device) if need else None
    N.call("lkg_gemm_f32", int(trans_a), int(trans_b), m, n, k, float(alpha), N.ptr(a), _ld(a), N.ptr(b), _ld(b),
           float(beta), N.ptr(out), _ld(out), N.ptr(bias), N.ptr(ws), ws.numel() if ws is not None else 0, _stream())
    return out



def colsum(x: torch.Tensor) -> torch.Tensor:
    x = _f32_rows(x)
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    N.call("lkg_colsum_f32", x.shape[0], x.shape[1], N.ptr(x), _ld(x), N.ptr(out), _stream())
    return out


def col_absmax(x: torch.Tensor) -> torch.Tensor:
    """out[c] = max_r |x[r, c]|: the column scales of lkg_gemm_wgrad_f32 for an operand nobody produced them for."""
    _need_gpu(x)
    x = _f32_rows(x)
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    N.call("lkg_col_absmax_f32", x.shape[0], x.shape[1], N.ptr(x), _ld(x), N.ptr(out), _stream())
    return out


def tag_colmax(t: torch.Tensor, cm: Optional[torch.Tensor]) -> torch.Tensor:
    """Remember max |t[:, j]| on the tensor OBJECT (as tag_rowmax does for rows)."""
    if cm is not None:
        t._lkg_colmax = (t._version, cm)
    return t


def tagged_colmax(t: torch.Tensor, cache: bool = False) -> Optional[torch.Tensor]:
    """The column maxima a producer left on t; with ``cache`` (constant tables: the literals) computed once and kept."""
    tag = getattr(t, "_lkg_colmax", None)
    if tag is not None and tag[0] == t._version and tag[1].shape[0] == t.shape[1] and tag[1].device == t.device:
        return tag[1]
    if not cache:
        return None
    cm = col_absmax(t)
    try:
        t._lkg_colmax = (t._version, cm)
    except AttributeError:
        pass
    return cm


def gemm_wgrad(a: torch.Tensor, b: torch.Tensor, a_colmax: torch.Tensor, b_colmax: torch.Tensor) -> torch.Tensor:
    """a^T @ b over the rows (a: k x m, b: k x n) on the fp16 matrix cores (lkg_gemm_wgrad_f32); the column maxima of
    both operands are inputs."""
    _need_gpu(a, b, a_colmax, b_colmax)
    a, b = _f32_rows(a), _f32_rows(b)
    if a.shape[0] != b.shape[0]:
        raise ValueError(f"gemm_wgrad: the operands have {a.shape[0]} and {b.shape[0]} rows")
    out = torch.empty((a.shape[1], b.shape[1]), dtype=torch.float32, device=a.device)
    N.call("lkg_gemm_wgrad_f32", a.shape[1], b.shape[1], a.shape[0], N.ptr(a), _ld(a), N.ptr(a_colmax), N.ptr(b), _ld(b),
           N.ptr(b_colmax), N.ptr(out), b.shape[1], _stream())
    return out


FRONTIER_GROWTH = 24           # a row set is followed through one more transpose SpMM while it is this many times smaller than N
NARROW_PANEL = 8               # input panels this narrow take their weight gradient from lkg_colsum_weighted_f32


def narrow_weight_grad(gy: torch.Tensor, panel: torch.Tensor, want_sum: bool):
    """(gy^T @ panel  [d x n_w], column sums of gy or None) in ONE pass over gy, for a panel of <= NARROW_PANEL columns."""
    gy, panel = _f32_rows(gy), _f32_rows(panel)
    d, n_w = gy.shape[1], panel.shape[1]
    gw = torch.empty((d, n_w), dtype=torch.float32, device=gy.device)
    gs = torch.empty(d, dtype=torch.float32, device=gy.device) if want_sum else None
    N.call("lkg_colsum_weighted_f32", gy.shape[0], d, N.ptr(gy), _ld(gy), N.ptr(panel), _ld(panel), n_w, N.ptr(gs),
           N.ptr(gw), n_w, _stream())
    return gw, gs


def weight_grads(gy: torch.Tensor, panels: Sequence[torch.Tensor], needed: Sequence[bool], want_sum: bool):
    """([gy^T @ p for p in panels] (None where not needed), column sums of gy or None): the long-k engine for the wide
    panels, the narrow ones riding on the bias gradient's pass over gy."""
    gws = [None] * len(panels)
    gs = None
    for i, p in enumerate(panels):
        if needed[i] and p.shape[1] <= NARROW_PANEL and gy.shape[0] >= 2048:
            gws[i], s_ = narrow_weight_grad(gy, p, want_sum and gs is None)
            gs = s_ if s_ is not None else gs
    if want_sum and gs is None:
        gs = colsum(gy)
    for i, p in enumerate(panels):
        if needed[i] and gws[i] is None:
            gws[i] = gemm(gy, p, trans_a=True)
    return gws, gs


# ----------------------------------------------------------------------------- tall GEMM (f16 x 2 engine)
import ctypes as _C
import os as _os

TALL_MIN_ROWS = 16384          # below this a product is bound by its launch overhead: the f32-MFMA engine serves it
_ENGINE = _os.environ.get("LKG_GEMM_ENGINE", "f16x2")      # f16x2 | bf16x3 (round-1 split engines) | f32 via lkg_gemm_f32
_WGRAD_ENGINE = _os.environ.get("LKG_WGRAD_ENGINE", "longk")  # longk (lkg_gemm_longk_f32) | bf16x3 (lkg_gemm_f32's long-k engine)
_workspaces = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Per (device, stream) scratch for B's fp16 planes: kernels of one stream run in order, so it is reused."""
    key = (device, _stream())
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = w
    return w


def _wants_rowmax(n_rows: int) -> bool:
    """Should a producer of an n_rows x d tensor also emit its row maxima (the consumer is likely a tall GEMM)?"""
    return _ENGINE in ("f16x2", "f16x2-all") and n_rows >= TALL_MIN_ROWS


def tag_rowmax(t: torch.Tensor, rm: Optional[torch.Tensor]) -> torch.Tensor:
    """Remember max |t[i,:]| on the tensor OBJECT (it dies with it; an in-place change of t invalidates it)."""
    if rm is not None:
        t._lkg_rowmax = (t._version, rm)
    return t


def tagged_rowmax(t: torch.Tensor) -> Optional[torch.Tensor]:
    tag = getattr(t, "_lkg_rowmax", None)
    if tag is None or tag[0] != t._version or tag[1].shape[0] != t.shape[0] or tag[1].device != t.device:
        return None
    return tag[1]


class RowSet:
    """The rows of an N-row table that may be non-zero: the <= 3B rows a loss's gradient reaches.  ``flags`` (uint8 per
    row) is what the kernels look at; ``compact_ids()`` lists every such row exactly once (no host sync: the list has the
    length of the id lists it came from, duplicates replaced by -1, which the row kernels skip)."""

    def __init__(self, flags: torch.Tensor, id_lists: Sequence[torch.Tensor], unique: bool = False):
        self.flags = flags
        self.id_lists = [i.reshape(-1) for i in id_lists]
        self.n_max = sum(i.numel() for i in self.id_lists)
        self._ids = self.id_lists[0] if (unique and len(self.id_lists) == 1) else None

    def compact_ids(self) -> torch.Tensor:
        if self._ids is None:       # (index bookkeeping over <= 3B ids)
            s_ = torch.sort(torch.cat(self.id_lists)).values
            dup = torch.zeros_like(s_, dtype=torch.bool)
            dup[1:] = s_[1:] == s_[:-1]
            self._ids = torch.where(dup, torch.full_like(s_, -1), s_)
        return self._ids


def union_rows(a: Optional[RowSet], b: Optional[RowSet]) -> Optional[RowSet]:
    """The rows of a or b (either may be None = no such gradient).  The flags of ``a`` are extended in place: they are
    the per-backward array a frontier SpMM produced, nobody else looks at them afterwards."""
    if a is None or b is None:
        return a if b is None else b
    if a is b:
        return a
    for ids in b.id_lists:
        N.call("lkg_fill_rows_f32", ids.numel(), 0, N.ptr(ids), None, 0, 0.0, N.ptr(a.flags), 1, _stream())
    return RowSet(a.flags, [a.compact_ids(), *b.id_lists])


def tag_rows(t: torch.Tensor, rows: Optional[RowSet]) -> torch.Tensor:
    """Remember on the tensor OBJECT that t is zero outside ``rows`` (a loss's row-sparse gradient and what the last
    layer's backward derives from it).  t itself stays dense and correct: a consumer that does not look at the tag just
    reads the zeros."""
    if rows is not None:
        t._lkg_rows = (t._version, rows)
    return t


def tagged_rows(t: Optional[torch.Tensor]) -> Optional[RowSet]:
    tag = getattr(t, "_lkg_rows", None) if t is not None else None
    if tag is None or tag[0] != t._version or tag[1].flags.shape[0] != t.shape[0] or tag[1].flags.device != t.device:
        return None
    return tag[1]


def _flags(rows: Optional[RowSet]) -> Optional[torch.Tensor]:
    return rows.flags if rows is not None else None


def rows_worth_compacting(rows: Optional[RowSet], n: int) -> bool:
    """Is the row set small enough that work on the listed rows beats a pass over all n?"""
    return rows is not None and n >= TALL_MIN_ROWS and rows.n_max * 8 <= n


_HAS_USE_COUNT = hasattr(torch._C, "_storage_Use_Count")


def _storage_users(t: torch.Tensor) -> int:
    """How many tensors / views share t's storage, or -1 when this torch build cannot tell (the counter torch's own
    multiprocessing reductions read; without it a kept-zero table is never RE-used -- see _RowScratch.acquire)."""
    if not _HAS_USE_COUNT:
        return -1
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


import threading as _threading


class _RowScratch:
    """The gradient a loss returns for the N x C entity table touches <= 3B of its rows, yet autograd wants it dense:
    an N x C zero fill per step (2 GB at 1 M x 512) plus the consumers' reads of those zeros.  Here ONE table per shape,
    purpose AND STREAM is kept all-zero BETWEEN steps: a backward resets the rows the previous one touched
    (lkg_fill_rows_f32), scatters its own, and tags the result with the RowSet so that the aware consumers skip the zero
    rows: act_ln backward, the transpose SpMM, and -- for the LAST layer, whose only gradient is that table's slice --
    the Linear's whole backward, which then works on the listed rows alone.

    Ownership.  A table is handed out as a fresh view; it may be handed out AGAIN only when nothing but this registry
    holds its storage any more -- no view of it alive in a graph, in a hook, as a parameter's .grad (autograd adopts a
    gradient tensor it is given when it can).  That is read off the storage's use count; a torch build without that
    counter never re-uses a table (every acquire allocates a fresh zero table: correct, one fill slower).  The registry is
    keyed by (device, rows, columns, purpose, stream) -- two models training on two streams never share a table, whose
    reset / scatter launches are ordered by ONE stream only -- and guarded by a lock (nn.DataParallel's replica threads,
    main_pretraining.py:69-71, run their backward passes concurrently)."""

    _tables: dict = {}
    _lock = _threading.Lock()
    MAX_TABLES = 16

    def __init__(self, shape, device, with_flags: bool):
        self.buf = torch.zeros(tuple(shape), dtype=torch.float32, device=device)
        self.flags = torch.zeros(shape[0], dtype=torch.uint8, device=device) if with_flags else None
        self.users = _storage_users(self.buf)
        self.dirty: list = []
        self.unknown = False          # rows were written that ``dirty`` does not list yet (a backward died in between)

    @classmethod
    def acquire(cls, n: int, c: int, device, pool: str = "loss") -> "_RowScratch":
        key = (device, n, c, pool, _stream())
        with cls._lock:
            ent = cls._tables.pop(key, None)
        if ent is None or ent.users < 0 or _storage_users(ent.buf) != ent.users:
            ent = _RowScratch((n, c), device, pool == "loss")     # (somebody still holds the old one: it is theirs now)
        with cls._lock:
            cls._tables[key] = ent                   # (most recently used last)
            while len(cls._tables) > cls.MAX_TABLES:     # another model / shape / stream: let the oldest table go
                cls._tables.pop(next(iter(cls._tables)))
        if ent.unknown:               # (only after an exception between a kernel's writes and their book-keeping)
            ent.buf.zero_()
            if ent.flags is not None:
                ent.flags.zero_()
            ent.unknown = False
        else:
            for ids in ent.dirty:
                N.call("lkg_fill_rows_f32", ids.numel(), ent.buf.shape[1], N.ptr(ids), N.ptr(ent.buf), _ld(ent.buf), 0.0,
                       N.ptr(ent.flags), 0, _stream())
        ent.dirty = []
        return ent

    def mark(self, *id_lists: torch.Tensor):
        for ids in id_lists:
            if self.flags is not None:
                N.call("lkg_fill_rows_f32", ids.numel(), 0, N.ptr(ids), None, 0, 0.0, N.ptr(self.flags), 1, _stream())
            self.dirty.append(ids)

    def table(self, rows: RowSet) -> torch.Tensor:
        """A fresh view per hand-out: while autograd (or anybody) holds it, the storage count shows it."""
        return tag_rows(self.buf.view(self.buf.shape), rows)


def _loss_grad_table(emb: torch.Tensor, sparse_rows: bool, *id_lists: torch.Tensor) -> torch.Tensor:
    """The all-zero N x C table a loss backward scatters its rows ``id_lists`` (int64, contiguous) into."""
    if not sparse_rows:
        return torch.zeros_like(emb, memory_format=torch.contiguous_format)
    ent = _RowScratch.acquire(emb.shape[0], emb.shape[1], emb.device)
    ent.mark(*id_lists)
    return ent.table(RowSet(ent.flags, id_lists))


def gather_rows_range(table: torch.Tensor, ids: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
    """out[i, :] = table[ids[i] - lo, :] for ids in [lo, hi), zeros for the others (-1 padding, rows of other shards)."""
    table = _f32_rows(table)
    out = torch.empty((ids.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    N.call("lkg_gather_rows_range_f32", ids.numel(), table.shape[1], N.ptr(table), _ld(table), N.ptr(ids), int(lo), int(hi),
           N.ptr(out), table.shape[1], _stream())
    return out


def zero_table_for(rows: RowSet, n: int, c: int, device, pool: str) -> torch.Tensor:
    """An n x c table that is zero everywhere and that the caller fills in ``rows`` only (tagged accordingly)."""
    ent = _RowScratch.acquire(n, c, device, pool)
    ent.mark(rows.compact_ids())
    return ent.table(rows)


def row_absmax(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[i] = max_j |x[i, j]| (accumulate: max with what out holds): the row scale of the tall GEMM's A operand."""
    _need_gpu(x)
    x = _f32_rows(x)
    if out is None:
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        accumulate = False
    N.call("lkg_row_absmax_f32", x.shape[0], x.shape[1], N.ptr(x), _ld(x), N.ptr(out), int(accumulate), _stream())
    return out


def rows_absmax(panels: Sequence[torch.Tensor]) -> torch.Tensor:
    """max over the panels of their row maxima; a panel whose producer tagged it costs no pass."""
    out = None
    for p in panels:
        rm = tagged_rowmax(p)
        if rm is not None and out is None and len(panels) == 1:
            return rm
        if rm is not None:
            out = rm.clone() if out is None else torch.maximum(out, rm, out=out)
        else:
            out = row_absmax(p, out, accumulate=out is not None)
    return out


def tall_ok(m: int, n: int, ks: Sequence[int], single_panel_too: bool = False) -> bool:
    """The tall engine takes the products whose inputs it reads ONCE where the round-1 engines need several
    accumulating launches: multi-panel Linears and the fused gate.  A plain one-panel product stays on the bf16 x 3
    engine of lkg_gemm_f32 (measured at 1 M x 256 x 256: 0.84 ms there, 0.72 ms + a 0.19 ms row-scale pass here) unless
    LKG_GEMM_ENGINE=f16x2-all asks for it."""
    if _ENGINE not in ("f16x2", "f16x2-all") or m < TALL_MIN_ROWS or not 1 <= len(ks) <= 3 or min(ks) <= 0:
        return False
    if len(ks) == 1 and not (single_panel_too or _ENGINE == "f16x2-all"):
        return False
    return n * sum(ks) <= (1 << 22)


TALL_VARIANTS = {"256x2": 0, "128x1": 1, "256x1": 2, "256x1w": 3, "ws": 4, "256r": 5}      # lkg_gemm_tall_f32: bits 8-15 of `epilogue` = id + 1
DEFAULT_TALL_VARIANT: Optional[str] = None      # None = the library's default; tests / tools set a key to steer whole modules


def gemm_tall(a_panels: Sequence[torch.Tensor], b_blocks: Sequence[Sequence[torch.Tensor]], trans_b: bool,
              bias: Optional[torch.Tensor] = None, alpha: float = 1.0, beta: float = 0.0,
              out: Optional[torch.Tensor] = None, rowmax: Optional[torch.Tensor] = None, gate_x=None, keep=None,
              variant: Optional[str] = None):
    """C = sum_p a_panels[p] @ op(B_p) (+ bias) for a tall A (lkg_gemm_tall_f32).  b_blocks[g][p]: block of B for row
    group g (1 group; 2 = the gate's stacked g / z projections with the blend epilogue on gate_x) and panel p; trans_b:
    blocks stored [n, k_p] (nn.Linear weights) else [k_p, n].  keep = (g_out, z_out) for the gate's backward.
    variant: None = the library's default tiling, or a key of TALL_VARIANTS (tests and tools run them side by side)."""
    if rowmax is None:
        rowmax = rows_absmax(a_panels)          # (before any .contiguous(): the producers' tags live on these objects)
    a_panels = [_f32_rows(a) for a in a_panels]
    m = a_panels[0].shape[0]
    if any(a.shape[0] != m for a in a_panels):
        raise ValueError(f"gemm_tall: the K-panels have {[a.shape[0] for a in a_panels]} rows")
    if rowmax.numel() != m:
        raise ValueError(f"gemm_tall: {rowmax.numel()} row maxima for {m} rows")
    ks = [a.shape[1] for a in a_panels]
    n_groups = len(b_blocks)
    gate = gate_x is not None
    if gate != (n_groups == 2):
        raise ValueError("gemm_tall: two weight groups go with the gate epilogue (and only with it)")
    blocks = [[_f32_rows(b) for b in grp] for grp in b_blocks]
    rows = blocks[0][0].shape[0] if trans_b else blocks[0][0].shape[1]
    for grp in blocks:
        if len(grp) != len(ks):
            raise ValueError("gemm_tall: one B block per panel and group")
        for b, k in zip(grp, ks):
            if tuple(b.shape) != ((rows, k) if trans_b else (k, rows)):
                raise ValueError(f"gemm_tall: B block of shape {tuple(b.shape)} does not match (rows {rows}, k {k})")
    n = rows * n_groups
    if out is None:
        if beta != 0.0:
            raise ValueError("gemm_tall: beta != 0 needs out")
        out = torch.empty((m, rows), dtype=torch.float32, device=a_panels[0].device)
    elif tuple(out.shape) != (m, rows) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError(f"gemm_tall: out must be a float32 {m} x {rows} tensor with unit column stride (got {tuple(out.shape)})")
    if bias is not None and bias.numel() != n:
        raise ValueError(f"gemm_tall: bias of {bias.numel()} elements for {n} (stacked) output columns")
    if gate and tuple(gate_x.shape) != (m, rows):
        raise ValueError(f"gemm_tall: the gate's x must be {m} x {rows} (got {tuple(gate_x.shape)})")
    if keep is not None and any(t_ is not None and (tuple(t_.shape) != (m, rows) or t_.stride(1) != 1) for t_ in keep):
        raise ValueError("gemm_tall: the kept tanh(g) / sigmoid(z) buffers must match the output's shape")
    np_ = len(ks)
    a_ptr = (_C.c_void_p * np_)(*[a.data_ptr() for a in a_panels])
    a_ld = (_C.c_int64 * np_)(*[_ld(a) for a in a_panels])
    a_k = (_C.c_int32 * np_)(*ks)
    flat = [b for grp in blocks for b in grp]
    b_ptr = (_C.c_void_p * len(flat))(*[b.data_ptr() for b in flat])
    b_ld = (_C.c_int64 * len(flat))(*[_ld(b) for b in flat])
    variant = variant if variant is not None else DEFAULT_TALL_VARIANT
    epi = (1 if gate else 0) | (0 if variant is None else (TALL_VARIANTS[variant] + 1) << 8)
    need = N.load().lkg_gemm_tall_workspace(n, np_, a_k, epi)
    ws = _workspace(int(need), out.device)
    gx = _f32_rows(gate_x) if gate else None
    g_out, z_out = keep if keep is not None else (None, None)
    N.call("lkg_gemm_tall_f32", m, n, np_, a_ptr, a_ld, a_k, N.ptr(rowmax), n_groups, b_ptr, b_ld, int(trans_b),
           float(alpha), float(beta), N.ptr(out), _ld(out), N.ptr(bias), epi, N.ptr(gx), _ld(gx) if gate else 0,
           N.ptr(g_out), _ld(g_out) if g_out is not None else 0, N.ptr(z_out), _ld(z_out) if z_out is not None else 0,
           N.ptr(ws), ws.numel(), _stream())
    return out


def fused_layer_ok(m: int, n: int, ks: Sequence[int]) -> bool:
    """Can lkg_linear_act_layernorm_fwd_f32 take this Linear + LeakyReLU + LayerNorm?  (whole rows in one 256-column tile)"""
    return (_ENGINE in ("f16x2", "f16x2-all") and m >= TALL_MIN_ROWS and 1 <= len(ks) <= 2 and min(ks) > 0 and 1 <= n <= 256
            and n * sum(ks) <= (1 << 22))


def linear_act_layernorm_fwd(a_panels: Sequence[torch.Tensor], w_blocks: Sequence[torch.Tensor], bias: Optional[torch.Tensor],
                             gamma: torch.Tensor, beta: torch.Tensor, slope: float, eps: float, norm_eps: float,
                             drop_p: float, seed: int, want_y: bool = True, want_norm: bool = True,
                             yn_out: Optional[torch.Tensor] = None, rowmax: Optional[torch.Tensor] = None):
    """(y, yn, mean, rstd) of  Dropout(LayerNorm(LeakyReLU(sum_p a_p @ w_p^T + bias)))  in ONE launch (K5 as surveyed:
    lkg_linear_act_layernorm_fwd_f32); no autograd here -- ``linear_act_layernorm`` wraps it."""
    if rowmax is None:
        rowmax = rows_absmax(a_panels)
    a_panels = [_f32_rows(a) for a in a_panels]
    ws_ = [_f32_rows(w) for w in w_blocks]
    _need_gpu(*a_panels, *ws_, gamma, beta)
    m, n = a_panels[0].shape[0], ws_[0].shape[0]
    ks = [a.shape[1] for a in a_panels]
    if any(a.shape[0] != m for a in a_panels) or any(tuple(w.shape) != (n, k) for w, k in zip(ws_, ks)) or len(ws_) != len(ks):
        raise ValueError("linear_act_layernorm: panels / weight blocks do not match")
    if not (want_y or want_norm):
        raise ValueError("linear_act_layernorm: neither output wanted")
    if rowmax.numel() != m or gamma.numel() != n or beta.numel() != n or (bias is not None and bias.numel() != n):
        raise ValueError("linear_act_layernorm: row maxima / gamma / beta / bias do not match the product's shape")
    dev = a_panels[0].device
    y = torch.empty((m, n), dtype=torch.float32, device=dev) if want_y else None
    yn = None
    if want_norm:
        yn = yn_out if yn_out is not None else torch.empty((m, n), dtype=torch.float32, device=dev)
        if tuple(yn.shape) != (m, n) or yn.stride(1) != 1 or yn.dtype != torch.float32:
            raise ValueError("linear_act_layernorm: the normalised copy's destination must be an m x n float32 view with unit column stride")
    mean = torch.empty(m, dtype=torch.float32, device=dev)
    rstd = torch.empty(m, dtype=torch.float32, device=dev)
    np_ = len(ks)
    a_ptr = (_C.c_void_p * np_)(*[a.data_ptr() for a in a_panels])
    a_ld = (_C.c_int64 * np_)(*[_ld(a) for a in a_panels])
    a_k = (_C.c_int32 * np_)(*ks)
    w_ptr = (_C.c_void_p * np_)(*[w.data_ptr() for w in ws_])
    w_ld = (_C.c_int64 * np_)(*[_ld(w) for w in ws_])
    need = N.load().lkg_linear_act_layernorm_workspace(n, np_, a_k)
    ws = _workspace(int(need), dev)
    N.call("lkg_linear_act_layernorm_fwd_f32", m, n, np_, a_ptr, a_ld, a_k, N.ptr(rowmax), w_ptr, w_ld, N.ptr(bias),
           float(slope), N.ptr(gamma), N.ptr(beta), float(eps), N.ptr(y), _ld(y) if y is not None else 0, N.ptr(yn),
           _ld(yn) if yn is not None else 0, float(norm_eps), N.ptr(mean), N.ptr(rstd), float(drop_p), int(seed), N.ptr(ws),
           ws.numel(), _stream())
    return y, yn, mean, rstd


# ----------------------------------------------------------------------------- K1+K2 attention refresh
@torch.no_grad()
def edge_softmax(g: KGStructure, ent: torch.Tensor, relemb: torch.Tensor, want_logits: bool = False,
                 row_lo: int = 0, row_hi: Optional[int] = None, out: Optional[torch.Tensor] = None
                 ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Attention values (CSR entry order) of head rows [row_lo, row_hi) -- model.py:430-471.
    Returns (values float32[nnz], merged logits or None); rows outside the range are left untouched."""
    _need_gpu(ent, relemb, g.rowptr)
    ent, relemb = _f32_rows(ent), _f32_rows(relemb)
    if ent.shape[1] != relemb.shape[1]:
        raise ValueError("update_att adds entity and relation embeddings: embed_dim must equal relation_dim "
                         "(model.py:441)")
    row_hi = g.n if row_hi is None else row_hi
    if ent.shape[0] != g.n or not 0 <= row_lo <= row_hi <= g.n:
        raise ValueError(f"edge_softmax: entity table of {ent.shape[0]} rows / row range [{row_lo}, {row_hi}) for a structure "
                         f"of {g.n} entities")
    if out is not None and (out.numel() != g.nnz or out.dtype != torch.float32 or not out.is_contiguous()):
        raise ValueError(f"edge_softmax: out must be a contiguous float32 array of {g.nnz} values")
    long_rows = g.long_rows(False, row_lo, row_hi)
    val = out if out is not None else torch.empty(g.nnz, dtype=torch.float32, device=ent.device)
    logits = torch.empty(g.nnz, dtype=torch.float32, device=ent.device) if want_logits else None
    if g.nnz == 0:          # no stored entry: nothing to refresh (the reference returns an empty A_in)
        return val, logits
    dup = (None, None, 0)
    if g.has_dups:          # entries with several raw edges inside this row range (rows relative to row_lo)
        de, dr = g.dup_entries, g.dup_rows
        if row_lo != 0 or row_hi != g.n:
            keep = (dr >= row_lo) & (dr < row_hi)
            de, dr = de[keep].contiguous(), (dr[keep] - row_lo).contiguous()
        dup = (N.ptr(de), N.ptr(dr), de.numel())
    if row_lo == 0 and row_hi == g.n:
        e_lo, e_hi = 0, g.nnz
    else:
        rp = g.host("rowptr")
        e_lo, e_hi = int(rp[row_lo]), int(rp[row_hi])
    N.call("lkg_edge_softmax_f32", row_hi - row_lo, row_lo, ent.shape[1], g.rowptr.data_ptr() + 4 * row_lo,
           N.ptr(g.col), N.ptr(g.eptr), N.ptr(g.rel), N.ptr(g.rel_first), *dup, e_lo, e_hi,
           N.ptr(ent), _ld(ent), N.ptr(relemb), _ld(relemb), N.ptr(val),
           N.ptr(logits), N.ptr(long_rows), 0 if long_rows is None else long_rows.numel(), LONG_ROW_THRESHOLD,
           relemb.shape[0], _stream())
    return val, logits


def permute_values(val: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """out[i] = val[perm[i]] for an int32 index list of any length (the CSC's values, a part of them); the caller
    guarantees 0 <= perm[i] < val.numel() -- the lists come from the structure build."""
    _need_gpu(val, perm)
    if perm.dtype != torch.int32 or val.dtype != torch.float32 or not perm.is_contiguous() or not val.is_contiguous():
        raise TypeError("permute_values: contiguous float32 values and a contiguous int32 index list")
    out = torch.empty(perm.numel(), dtype=val.dtype, device=val.device)
    N.call("lkg_permute_f32", perm.numel(), N.ptr(perm), val.numel(), N.ptr(val), N.ptr(out), _stream())
    return out


# ----------------------------------------------------------------------------- K3/K4 aggregation
def _check_table(ego: torch.Tensor, g: KGStructure, val: torch.Tensor):
    """The kernels gather rows of ego through g's column ids: a table with fewer rows than the structure has entities
    would be read out of bounds."""
    if ego.dim() != 2 or ego.shape[0] != g.n:
        raise ValueError(f"aggregate: the table has shape {tuple(ego.shape)}, the structure {g.n} entities")
    if val.numel() != g.nnz:
        raise ValueError(f"aggregate: {val.numel()} attention values for {g.nnz} stored entries")


class _Aggregate(Function):
    """side = A @ ego over the CSR; backward A^T @ grad over the CSC (A carries no gradient,
    model.py:261)."""

    @staticmethod
    def forward(ctx, ego, g: KGStructure, val, val_t, plus_self, bias=None):
        _need_gpu(ego, val, bias)
        _check_table(ego, g, val)
        ctx.g = g
        ctx.val_t = val_t
        ctx.plus_self = plus_self
        ctx.has_bias = bias is not None
        rm = torch.empty(g.n, dtype=torch.float32, device=ego.device) if _wants_rowmax(g.n) else None
        return tag_rowmax(spmm_raw(g.rowptr, g.col, val, ego, g.n, long_rows=g.long_rows(False),
                                   add_self=ego if plus_self else None, rowmax=rm,
                                   bias=bias.detach().contiguous() if bias is not None else None,
                                   row_lists=g.row_lists(False)), rm)

    @staticmethod
    def backward(ctx, grad):
        g_ego, *rest = _Aggregate._backward_ego(ctx, grad)
        g_bias = None
        if ctx.has_bias and ctx.needs_input_grad[5]:        # the bias reaches every row: its gradient is the column sum
            rs = tagged_rows(grad)
            g_bias = (colsum(gather_rows_range(grad, rs.compact_ids(), 0, grad.shape[0]))
                      if rows_worth_compacting(rs, grad.shape[0]) else colsum(grad))
        return (g_ego, *rest, g_bias)

    @staticmethod
    def _backward_ego(ctx, grad):
        g = ctx.g
        if g.t_rowptr is None:
            raise RuntimeError("KGStructure was built without its transpose; backward needs the CSC")
        rs = tagged_rows(grad)                # the last layer's gradient: all but <= 3B rows are zero and are not gathered
        rows = _flags(rs) if rows_worth_compacting(rs, g.n) else None      # (a dense-ish row set: the plain kernel is faster)
        grad = _f32_rows(grad)
        d = grad.shape[1]
        if (rows_worth_compacting(rs, g.n) and rs.n_max * FRONTIER_GROWTH <= g.n and d % 4 == 0 and d <= 1024
                and grad.data_ptr() % 16 == 0 and _ld(grad) % 4 == 0):
            # ... and the rows it reaches in turn (the gradient's frontier) are few as well: the result goes into a table kept
            # all-zero elsewhere, the kernel flags the rows it wrote, and the layer below works on those (one host sync: their
            # number decides the shapes of its products)
            ent = _RowScratch.acquire(g.n, d, grad.device, "g_agg")
            out = ent.buf.view(ent.buf.shape)
            reached = torch.empty(g.n, dtype=torch.uint8, device=grad.device)
            ent.unknown = True            # (until the rows the kernel writes are on record)
            spmm_raw(g.t_rowptr, g.t_col, ctx.val_t, grad, g.n, out=out, long_rows=g.long_rows(True),
                     add_self=grad if ctx.plus_self else None, x_rows=rows, self_rows=rows, out_rows=reached)
            ids = torch.nonzero(reached).flatten()
            ent.dirty.append(ids)
            ent.unknown = False
            return tag_rows(out, RowSet(reached, [ids], unique=True)), None, None, None, None
        return spmm_raw(g.t_rowptr, g.t_col, ctx.val_t, grad, g.n, long_rows=g.long_rows(True),
                        add_self=grad if ctx.plus_self else None, x_rows=rows, self_rows=rows,
                        row_lists=g.row_lists(True)), None, None, None, None


def aggregate(ego: torch.Tensor, g: KGStructure, val: torch.Tensor, val_t: torch.Tensor,
              plus_self: bool = False, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """side = A @ ego, or ego + side in one pass when plus_self; bias: a row added to every output row."""
    return _Aggregate.apply(ego, g, val, val_t, plus_self, bias)


class _AggregateKeep(Function):
    """The first aggregation layer together with the OTHER consumer of its input: gat_embeddings keeps the layer
    input itself as column slot 0 of the concatenated table (model.py:300-309).  Returns (side, kept): kept is ego --
    copied into ``keep_dst`` by the SpMM's epilogue when ego is not stored there already (the raw entity table, no
    gate).  Backward: A^T g_side (+ g_side) + g_kept in ONE launch (autograd would add the two contributions to ego
    in a separate N x D pass)."""

    @staticmethod
    def forward(ctx, ego, g: KGStructure, val, val_t, plus_self, keep_dst):
        _need_gpu(ego, val)
        _check_table(ego, g, val)
        ctx.g, ctx.val_t, ctx.plus_self = g, val_t, plus_self
        # (a leaf input -- the raw entity table of a model without a gate -- takes its gradient dense: its consumer is the
        # optimizer, nobody would read a row set; the frontier form pays off when a gate's backward follows)
        ctx.follow_frontier = not ego.is_leaf
        ctx.set_materialize_grads(False)
        copy = None
        kept = ego
        if keep_dst is not None and not _same_view(keep_dst, ego):
            copy, kept = (ego, keep_dst), keep_dst
        rm = torch.empty(g.n, dtype=torch.float32, device=ego.device) if _wants_rowmax(g.n) else None
        side = spmm_raw(g.rowptr, g.col, val, ego, g.n, long_rows=g.long_rows(False),
                        add_self=ego if plus_self else None, copy=copy, rowmax=rm, row_lists=g.row_lists(False))
        return tag_rowmax(side, rm), kept          # kept may be the input itself: autograd aliases it as this node's output

    @staticmethod
    def backward(ctx, g_side, g_kept):
        g = ctx.g
        if g_side is None:
            return g_kept, None, None, None, None, None
        if g.t_rowptr is None:
            raise RuntimeError("KGStructure was built without its transpose; backward needs the CSC")
        rs = tagged_rows(g_side)
        rows = _flags(rs) if rows_worth_compacting(rs, g.n) else None
        g_side = _f32_rows(g_side)
        rk = tagged_rows(g_kept)
        d = g_side.shape[1]
        if (ctx.follow_frontier and rows is not None and (g_kept is None or rk is not None) and rs.n_max * FRONTIER_GROWTH <= g.n and d % 4 == 0
                and d <= 1024 and g_side.data_ptr() % 16 == 0 and _ld(g_side) % 4 == 0
                and (g_kept is None or (g_kept.data_ptr() % 16 == 0 and _ld(g_kept) % 4 == 0))):
            # both gradients are zero outside a few rows (a one-layer model behind a gate: the loss's rows and what one
            # transpose SpMM reaches from them): the sum goes into a table kept all-zero elsewhere, the kernel flags the rows
            # it wrote, and the gate's backward works on those
            ent = _RowScratch.acquire(g.n, d, g_side.device, "g_agg_keep")
            out = ent.buf.view(ent.buf.shape)
            reached = torch.empty(g.n, dtype=torch.uint8, device=g_side.device)
            ent.unknown = True
            spmm_raw(g.t_rowptr, g.t_col, ctx.val_t, g_side, g.n, out=out, long_rows=g.long_rows(True),
                     add_self=g_side if ctx.plus_self else None, add2=g_kept, add2_rows=_flags(rk), x_rows=rows, self_rows=rows,
                     out_rows=reached)
            ids = torch.nonzero(reached).flatten()
            ent.dirty.append(ids)
            ent.unknown = False
            return tag_rows(out, RowSet(reached, [ids], unique=True)), None, None, None, None, None
        return spmm_raw(g.t_rowptr, g.t_col, ctx.val_t, g_side, g.n, long_rows=g.long_rows(True),
                        add_self=g_side if ctx.plus_self else None, add2=g_kept,
                        add2_rows=_flags(rk), x_rows=rows, self_rows=rows, row_lists=g.row_lists(True)), None, None, None, None, None


def aggregate_keep(ego, g: KGStructure, val, val_t, plus_self: bool = False, keep_dst: Optional[torch.Tensor] = None):
    return _AggregateKeep.apply(ego, g, val, val_t, plus_self, keep_dst)


class _Fanout(Function):
    """One tensor, two consumers, and a backward that keeps what autograd's own sum of the two gradients loses: when both
    are zero outside a few rows (the loss's rows of column slot 0 of the concatenated table, and the rows a layer's backward
    reached) the sum is built on the union of those rows, in a table kept all-zero elsewhere that says which rows they are --
    autograd would add two N x D tables (a pass over both) and hand the layer below a dense, untagged one."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        ra, rb = tagged_rows(ga), tagged_rows(gb)
        n, d = ga.shape
        if ra is not None and rb is not None and rows_worth_compacting(ra, n) and rows_worth_compacting(rb, n):
            flags = ra.flags.clone()
            rows = union_rows(RowSet(flags, ra.id_lists), rb)
            if rows_worth_compacting(rows, n):
                out = zero_table_for(rows, n, d, ga.device, "g_fanout")
                for g_, r_ in ((ga, ra), (gb, rb)):
                    ids = r_.compact_ids()
                    part = gather_rows_range(g_, ids, 0, n)
                    N.call("lkg_scatter_add_rows_range_f32", ids.numel(), d, N.ptr(part), d, N.ptr(ids), 0, n, N.ptr(out), _ld(out),
                           _stream())
                return out
        return _elt(0, ga, gb, 1.0, 1.0)


def fanout(x: torch.Tensor):
    """(x, x) for two consumers whose gradients may both be row-sparse (see _Fanout)."""
    return _Fanout.apply(x)


# ----------------------------------------------------------------------------- dense layers on the MFMA GEMM
class _MultiLinear(Function):
    """y = sum_i x_i @ w_i^T + bias  (nn.Linear on a column-concatenated input without the cat:
    gate.py:22-25; plain nn.Linear is the one-term case)."""

    @staticmethod
    def forward(ctx, bias, n_terms, *xw):
        xs, ws = xw[:n_terms], xw[n_terms:]
        _need_gpu(*xs, *ws)
        y = None
        if n_terms == 1 and _skinny(xs[0], ws[0].shape[0]):
            y = gemm(xs[0], ws[0], trans_b=True, bias=bias)            # (narrow in and out: the streaming VALU kernel)
        elif tall_ok(xs[0].shape[0], ws[0].shape[0], [x.shape[1] for x in xs],
                     single_panel_too=tagged_rowmax(xs[0]) is not None):
            # every panel in ONE launch: the accumulators stay in registers, the inputs are read once
            y = gemm_tall(xs, (ws,), True, bias)
        else:
            for i, (x, w) in enumerate(zip(xs, ws)):
                y = gemm(x, w, trans_b=True, beta=0.0 if i == 0 else 1.0, out=y, bias=bias if i == 0 else None)
        ctx.save_for_backward(*xs, *ws)
        ctx.n_terms = n_terms
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        n = ctx.n_terms
        saved = ctx.saved_tensors
        xs, ws = saved[:n], saved[n:]
        rows = tagged_rows(gy)
        if rows_worth_compacting(rows, gy.shape[0]):
            return _MultiLinear._backward_on_rows(ctx, gy, rows, xs, ws)
        gy = _f32_rows(gy)
        gxs = []
        rm = None
        for i in range(n):
            if not ctx.needs_input_grad[2 + i]:
                gxs.append(None)
            elif _skinny(gy, ws[i].shape[1]):
                gxs.append(gemm(gy, ws[i]))
            elif tall_ok(gy.shape[0], ws[i].shape[1], (gy.shape[1],), single_panel_too=tagged_rowmax(gy) is not None):
                rm = rows_absmax((gy,)) if rm is None else rm    # one scale (tagged by the producer, or one pass) for all
                gxs.append(gemm_tall((gy,), ((ws[i],),), False, rowmax=rm))
            else:
                gxs.append(gemm(gy, ws[i]))
        gws, gb = weight_grads(gy, xs, ctx.needs_input_grad[2 + n:2 + 2 * n], ctx.has_bias and ctx.needs_input_grad[0])
        return (gb, None, *gxs, *gws)

    @staticmethod
    def _backward_on_rows(ctx, gy, rows: RowSet, xs, ws):
        """gy is zero outside ``rows`` (the last layer under a loss on <= 3B rows): every product of the backward is the
        same product over the listed rows -- gathered once, a few thousand of them -- and the data gradient is their
        scatter into a table that is kept all-zero elsewhere (and tells ITS consumer which rows those are)."""
        n = ctx.n_terms
        ids = rows.compact_ids()                                   # every row once, -1 padding (gathers as zeros)
        n_rows = gy.shape[0]
        gc = gather_rows_range(gy, ids, 0, n_rows)
        gb = colsum(gc) if (ctx.has_bias and ctx.needs_input_grad[0]) else None
        gxs, gws = [], []
        for i in range(n):
            if not ctx.needs_input_grad[2 + i]:
                gxs.append(None)
                continue
            gx = zero_table_for(rows, n_rows, ws[i].shape[1], gy.device, f"g_x{i}")
            N.call("lkg_scatter_add_rows_range_f32", ids.numel(), ws[i].shape[1], N.ptr(gemm(gc, ws[i])), ws[i].shape[1],
                   N.ptr(ids), 0, n_rows, N.ptr(gx), _ld(gx), _stream())
            gxs.append(gx)
        for i in range(n):
            gws.append(gemm(gc, gather_rows_range(xs[i], ids, 0, n_rows), trans_a=True)
                       if ctx.needs_input_grad[2 + n + i] else None)
        return (gb, None, *gxs, *gws)


def _skinny(x: torch.Tensor, n_out: int) -> bool:
    """Is x @ W^T (n_out columns) a product for lkg_gemm_skinny_f32 (many rows, <= 64 columns in and out)?"""
    # (k * n <= 2048: measured at 766 k x 64 x 64 the VALU kernel takes 157 us, the 128-column matrix-core tile 131 us)
    return (x.dim() == 2 and x.shape[0] >= 4096 and 4 <= x.shape[1] <= 64 and 4 <= n_out <= 64 and x.shape[1] % 4 == 0
            and x.shape[1] * n_out <= 2048
            and n_out % 4 == 0 and _ENGINE != "f32" and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def linear(x, w, b=None):
    return _MultiLinear.apply(b, 1, x, w)


def multi_linear(xs: Sequence[torch.Tensor], ws: Sequence[torch.Tensor], b=None):
    return _MultiLinear.apply(b, len(xs), *xs, *ws)


class _StackedLinear(Function):
    """ys[l] = x @ ws[l]^T + bs[l] for several Linears over the SAME input, as ONE product (the GCNII-style residual projects
    the layer-0 table h0 once per layer, model.py:93: eight Linears over the same N x embed_dim input at the reference's
    main.py defaults -- and bi-interaction asks for each of them twice).  Forward: one GEMM over the stacked weights, the
    outputs are its column slices.  Backward: the slices' gradients side by side in one buffer, ONE data-gradient product
    (K = the stacked width) instead of one N x embed_dim table per Linear for autograd to add up, one weight-gradient
    product, one column sum."""

    @staticmethod
    def forward(ctx, x, n_lin, *wb):
        ws, bs = wb[:n_lin], wb[n_lin:]
        _need_gpu(x, *ws, *bs)
        x = _f32_rows(x)
        w_all = torch.cat([w.detach() for w in ws], 0).contiguous()
        b_all = torch.cat([b.detach() for b in bs]).contiguous()
        m, k = x.shape
        n = w_all.shape[0]
        if tall_ok(m, n, (k,), single_panel_too=True):
            y = gemm_tall((x,), ((w_all,),), True, b_all)
        else:
            y = gemm(x, w_all, trans_b=True, bias=b_all)
        ctx.save_for_backward(x, w_all)
        ctx.widths = [int(w.shape[0]) for w in ws]
        ctx.set_materialize_grads(False)
        outs, o = [], 0
        for wd in ctx.widths:
            outs.append(y[:, o:o + wd])
            o += wd
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        x, w_all = ctx.saved_tensors
        m, n = x.shape[0], w_all.shape[0]
        g_all = torch.empty((m, n), dtype=torch.float32, device=x.device)
        o = 0
        for wd, g in zip(ctx.widths, gs):
            if g is None:
                g_all[:, o:o + wd].zero_()
            else:
                fill_slot(g_all, o, g)
            o += wd
        n_lin = len(ctx.widths)
        need = ctx.needs_input_grad
        gx = gemm(g_all, w_all) if need[0] else None
        gws, gbs = [None] * n_lin, [None] * n_lin
        if any(need[2:2 + n_lin]):
            gw_all = gemm(g_all, x, trans_a=True)
            o = 0
            for i, wd in enumerate(ctx.widths):
                gws[i] = gw_all[o:o + wd] if need[2 + i] else None
                o += wd
        if any(need[2 + n_lin:]):
            gb_all = colsum(g_all)
            o = 0
            for i, wd in enumerate(ctx.widths):
                gbs[i] = gb_all[o:o + wd] if need[2 + n_lin + i] else None
                o += wd
        return (gx, None, *gws, *gbs)


def stacked_linear(x: torch.Tensor, ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor]):
    """[x @ w^T + b for w, b in zip(ws, bs)] as one product, forward and backward."""
    return _StackedLinear.apply(x, len(ws), *ws, *bs)


class _MatMul(Function):
    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        ctx.save_for_backward(a, b)
        return gemm(a, b)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = _f32_rows(gc)
        ga = gemm(gc, b, trans_b=True) if ctx.needs_input_grad[0] else None
        gb = gemm(a, gc, trans_a=True) if ctx.needs_input_grad[1] else None
        return ga, gb


def matmul(a, b):
    return _MatMul.apply(a, b)


# ----------------------------------------------------------------------------- small element-wise steps
def gemm_f64acc(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False) -> torch.Tensor:
    """op(a) @ op(b) with float64 accumulation, rounded to float32 once (lkg_gemm_f64acc_f32: small products only)."""
    a, b = _f32_rows(a), _f32_rows(b)
    _need_gpu(a, b)
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    kb, n = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if k != kb:
        raise ValueError(f"gemm_f64acc: inner dimensions {k} and {kb} differ")
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    N.call("lkg_gemm_f64acc_f32", int(trans_a), int(trans_b), m, n, k, N.ptr(a), _ld(a), N.ptr(b), _ld(b), N.ptr(out), _ld(out),
           _stream())
    return out


class _FoldNT(Function):
    """a @ b^T for two SMALL matrices, float64 accumulation (forward and both gradients): the residual's weight fold."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return gemm_f64acc(a, b, trans_b=True)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = gemm_f64acc(g, b) if ctx.needs_input_grad[0] else None
        gb = gemm_f64acc(g, a, trans_a=True) if ctx.needs_input_grad[1] else None
        return ga, gb


FOLD_F64 = _os.environ.get("LKG_FOLD_F32", "0") in ("", "0")      # (LKG_FOLD_F32=1: the fp32 fold of rounds 2-3, kept for
#                                                                     before / after measurements of the residual sweeps)


def fold_nt(a, b):
    """a @ b^T, float64 accumulation (small matrices)"""
    return _FoldNT.apply(a, b) if FOLD_F64 else matmul(a, b.t())


def _elt(op, a, b=None, alpha=1.0, beta=0.0):
    a = _f32_rows(a)
    b = _f32_rows(b) if b is not None else None
    if b is not None and b.shape != a.shape:
        raise ValueError(f"element-wise operands differ in shape: {tuple(a.shape)} vs {tuple(b.shape)}")
    out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    N.call("lkg_eltwise_f32", op, a.shape[0], a.shape[1], N.ptr(a), _ld(a), N.ptr(b), _ld(b) if b is not None else 0,
           float(alpha), float(beta), N.ptr(out), _ld(out), _stream())
    return out


class _Axpby(Function):
    """alpha * a + beta * b  (b None: alpha * a + beta)."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        _need_gpu(a, b)
        ctx.ab = (alpha, beta, b is not None)
        return _elt(0, a, b, alpha, beta)

    @staticmethod
    def backward(ctx, g):
        alpha, beta, has_b = ctx.ab
        rows = tagged_rows(g)            # (element-wise: a gradient that is zero outside a row set stays zero outside it)
        ga = tag_rows(_elt(0, g, None, alpha, 0.0), rows) if ctx.needs_input_grad[0] else None
        gb = tag_rows(_elt(0, g, None, beta, 0.0), rows) if has_b and ctx.needs_input_grad[1] else None
        return ga, gb, None, None


def axpby(a, b, alpha=1.0, beta=1.0):
    return _Axpby.apply(a, b, alpha, beta)


class _Mul(Function):
    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        ctx.save_for_backward(a, b)
        return _elt(1, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        rows = tagged_rows(g)
        return (tag_rows(_elt(1, g, b), rows) if ctx.needs_input_grad[0] else None,
                tag_rows(_elt(1, g, a), rows) if ctx.needs_input_grad[1] else None)


def mul(a, b):
    return _Mul.apply(a, b)


class _BiMix(Function):
    """(c (ego + side) + alpha h0p, c (ego * side) + alpha h0p) with c = 1 - alpha, or the plain (sum, product) without h0p:
    bi-interaction's two branch inputs (model.py:123-128) and the residual's mix (model.py:94) in ONE pass, one pass back."""

    @staticmethod
    def forward(ctx, ego, side, h0p, alpha):
        _need_gpu(ego, side, h0p)
        ego, side = _f32_rows(ego), _f32_rows(side)
        h = _f32_rows(h0p) if h0p is not None else None
        if side.shape != ego.shape or (h is not None and h.shape != ego.shape):
            raise ValueError("bi_mix: ego, side (and h0p) must have one shape")
        n, d = ego.shape
        out_sum = torch.empty((n, d), dtype=torch.float32, device=ego.device)
        out_prod = torch.empty((n, d), dtype=torch.float32, device=ego.device)
        N.call("lkg_bi_mix_fwd_f32", n, d, N.ptr(ego), _ld(ego), N.ptr(side), _ld(side), N.ptr(h), _ld(h) if h is not None else 0,
               float(alpha), N.ptr(out_sum), d, N.ptr(out_prod), d, _stream())
        ctx.save_for_backward(ego, side)
        ctx.cfg = (h is not None, float(alpha))
        ctx.set_materialize_grads(False)
        return out_sum, out_prod

    @staticmethod
    def backward(ctx, g_sum, g_prod):
        ego, side = ctx.saved_tensors
        has_h0, alpha = ctx.cfg
        if g_sum is None and g_prod is None:
            return None, None, None, None
        rs, rp = tagged_rows(g_sum), tagged_rows(g_prod)
        rows = rs if (rs is rp or g_prod is None) else (rp if g_sum is None else None)     # (one row set: it travels on)
        g_sum = _f32_rows(g_sum) if g_sum is not None else torch.zeros_like(ego)
        g_prod = _f32_rows(g_prod) if g_prod is not None else torch.zeros_like(ego)
        n, d = ego.shape
        g_ego = torch.empty((n, d), dtype=torch.float32, device=ego.device)
        g_side = torch.empty((n, d), dtype=torch.float32, device=ego.device)
        g_h = torch.empty((n, d), dtype=torch.float32, device=ego.device) if has_h0 else None
        N.call("lkg_bi_mix_bwd_f32", n, d, N.ptr(ego), _ld(ego), N.ptr(side), _ld(side), N.ptr(g_sum), _ld(g_sum), N.ptr(g_prod),
               _ld(g_prod), int(has_h0), float(alpha), N.ptr(g_ego), N.ptr(g_side), N.ptr(g_h), _stream())
        return tag_rows(g_ego, rows), tag_rows(g_side, rows), (tag_rows(g_h, rows) if has_h0 else None), None


def bi_mix(ego, side, h0p=None, alpha: float = 0.0):
    return _BiMix.apply(ego, side, h0p, alpha)


class _LeakySum(Function):
    """leaky_relu(a) + leaky_relu(b)  (b None: leaky_relu(a))."""

    @staticmethod
    def forward(ctx, a, b, slope):
        _need_gpu(a, b)
        ctx.save_for_backward(a, b) if b is not None else ctx.save_for_backward(a)
        ctx.slope = slope
        return _elt(2, a, b, slope)

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        # the LeakyReLU behind linear_gat (model.py:310) sits between the loss and the whole encoder: the loss's gradient is
        # zero outside <= 3B rows and so is this one -- the row set travels on, or everything below would run dense
        rows = tagged_rows(g)
        if rows_worth_compacting(rows, g.shape[0]):
            # ... and only the listed rows are computed (gathered, passed through LeakyReLU', scattered into a table that is
            # kept all-zero elsewhere) instead of a pass over all N rows
            ids = rows.compact_ids()
            n_rows, d = g.shape
            gc = gather_rows_range(g, ids, 0, n_rows)
            outs = []
            for i in range(len(saved)):
                if not ctx.needs_input_grad[i]:
                    outs.append(None)
                    continue
                part = _elt(3, gc, gather_rows_range(saved[i], ids, 0, n_rows), ctx.slope)
                dst = zero_table_for(rows, n_rows, d, g.device, f"g_leaky{i}")
                N.call("lkg_scatter_add_rows_range_f32", ids.numel(), d, N.ptr(part), d, N.ptr(ids), 0, n_rows, N.ptr(dst),
                       _ld(dst), _stream())
                outs.append(dst)
            return (outs[0], outs[1] if len(outs) > 1 else None, None)
        ga = tag_rows(_elt(3, g, saved[0], ctx.slope), rows) if ctx.needs_input_grad[0] else None
        gb = tag_rows(_elt(3, g, saved[1], ctx.slope), rows) if len(saved) > 1 and ctx.needs_input_grad[1] else None
        return ga, gb, None


def leaky_relu(a, slope=LEAKY_SLOPE):
    return _LeakySum.apply(a, None, slope)


def leaky_relu_sum(a, b, slope=LEAKY_SLOPE):
    return _LeakySum.apply(a, b, slope)


# ----------------------------------------------------------------------------- K5 epilogue
class _ActLayerNorm(Function):
    """y = Dropout(LayerNorm(LeakyReLU(z))); yn = y / max(|y|_2, eps)  (model.py:111, 161, 305).
    The dropout mask is counter-based (seed, element index) and regenerated in the backward."""

    @staticmethod
    def forward(ctx, z, gamma, beta, want_norm, slope, eps, norm_eps, drop_p, seed, yn_out, want_y):
        _need_gpu(z, gamma, beta)
        z = _f32_rows(z)
        n, d = z.shape
        if not want_y and not want_norm:
            raise ValueError("act_layernorm: neither output wanted")
        # (the LAST layer's y is read by nobody -- only its normalised copy is kept: 4 n d bytes not written; the
        # backward recomputes the few rows of y it needs)
        y = torch.empty((n, d), dtype=torch.float32, device=z.device) if want_y else None
        yn = None
        if want_norm:   # yn_out: a column slice of the concat buffer (CatBuffer), written in place
            yn = yn_out if yn_out is not None else torch.empty((n, d), dtype=torch.float32, device=z.device)
        mean = torch.empty(n, dtype=torch.float32, device=z.device)
        rstd = torch.empty(n, dtype=torch.float32, device=z.device)
        N.call("lkg_act_layernorm_fwd_f32", n, d, N.ptr(z), _ld(z), float(slope), N.ptr(gamma), N.ptr(beta),
               float(eps), N.ptr(y), _ld(y) if y is not None else 0, N.ptr(yn), _ld(yn) if yn is not None else 0,
               float(norm_eps), N.ptr(mean), N.ptr(rstd), float(drop_p), int(seed), _stream())
        ctx.save_for_backward(z, gamma, beta, mean, rstd, *([y] if y is not None else []))
        ctx.cfg = (slope, norm_eps, drop_p, seed)
        ctx.set_materialize_grads(False)
        if want_norm:
            return y, yn
        return y, None

    @staticmethod
    def backward(ctx, gy, gyn):
        z, gamma, beta, mean, rstd, *kept_y = ctx.saved_tensors
        y = kept_y[0] if kept_y else None
        slope, norm_eps, drop_p, seed = ctx.cfg
        n, d = z.shape
        none = (None,) * 11
        if gy is None and gyn is None:
            return none
        rows_n = tagged_rows(gyn)        # a loss's row-sparse gradient: the kernel skips the zero rows
        rows_y = tagged_rows(gy)         # the frontier a transpose SpMM over row-sparse input reached
        gy = _f32_rows(gy) if gy is not None else None
        gyn = _f32_rows(gyn) if gyn is not None else None
        gg = torch.zeros(d, dtype=torch.float32, device=z.device)
        gb = torch.zeros(d, dtype=torch.float32, device=z.device)
        # no gradient outside a few rows (the LAST layer: g_yn's rows, no g_y; the layer below: + the frontier g_y reaches):
        # g_z is zero outside them too -- it goes into a table kept all-zero between steps (only the listed rows are visited)
        # and carries the row set on to the Linear's backward and the transpose SpMM
        rows = None
        if (gy is None or rows_y is not None) and (gyn is None or rows_n is not None):
            rows = union_rows(rows_y, rows_n)
        sparse_out = rows_worth_compacting(rows, n)
        ids = rows.compact_ids() if sparse_out else None
        if sparse_out:
            gz, rm = zero_table_for(rows, n, d, z.device, "g_z"), None
        else:
            gz = torch.empty((n, d), dtype=torch.float32, device=z.device)
            rm = torch.empty(n, dtype=torch.float32, device=z.device) if _wants_rowmax(n) else None   # for the Linear's data gradient
        N.call("lkg_act_layernorm_bwd_f32", n, d, N.ptr(z), _ld(z), float(slope), N.ptr(gamma), N.ptr(beta), N.ptr(y),
               _ld(y) if y is not None else 0, N.ptr(mean), N.ptr(rstd), N.ptr(gy), _ld(gy) if gy is not None else 0, N.ptr(gyn),
               _ld(gyn) if gyn is not None else 0, float(norm_eps), N.ptr(gz), _ld(gz), N.ptr(gg), N.ptr(gb),
               float(drop_p), int(seed), N.ptr(rm), N.ptr(_flags(rows_n)), int(sparse_out),
               N.ptr(ids) if sparse_out else None, ids.numel() if sparse_out else 0, _stream())
        return (tag_rowmax(gz, rm), gg, gb) + none[3:]


def new_seed() -> int:
    """63-bit seed drawn from torch's CPU generator (follows torch.manual_seed, no device sync)."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


def act_layernorm(z, gamma, beta, want_norm=True, slope=LEAKY_SLOPE, eps=LN_EPS, norm_eps=NORMALIZE_EPS,
                  drop_p: float = 0.0, seed: Optional[int] = None, yn_out: Optional[torch.Tensor] = None,
                  want_y: bool = True):
    """(y, yn); want_y = False: y is not produced (returned as None) -- the last layer, whose output only its normalised
    copy survives."""
    if drop_p > 0 and seed is None:
        seed = new_seed()
    return _ActLayerNorm.apply(z, gamma, beta, want_norm, slope, eps, norm_eps, drop_p, seed or 0, yn_out, want_y)


# ----------------------------------------------------------------------------- K5 in one launch (opt-in)
def _fused_layer_mode():
    v = _os.environ.get("LKG_FUSED_LAYER", "auto").strip().lower()
    return {"": "auto", "auto": "auto", "0": False, "off": False, "1": True, "on": True}.get(v, "auto")


FUSED_LAYER = _fused_layer_mode()
"""Route Linear + LeakyReLU + LayerNorm (+ dropout + normalised copy) of an aggregation layer through ONE launch
(lkg_linear_act_layernorm_fwd_f32).  "auto" (the default): where it is the faster way -- rows wider than 128 columns (the
launch's tile is 256 columns wide whatever the layer's width) and NO gradient to come, i.e. evaluation / link scoring:
measured on MI355X (profiles/r04_tall_after_epilogue_fix.log, 1 M x 256 x 256) 1.04 ms against 1.14 ms for the tall GEMM
followed by the row-wise kernel.  In training the pair stays: the backward pass needs z = x W^T + b, which the fused launch does not
write, and recomputing it costs a second GEMM.  True: wherever the shape allows (tests); False: never.  Same bits every way."""


def fused_layer_wanted(xs: Sequence[torch.Tensor], ws: Sequence[torch.Tensor], others: Sequence[Optional[torch.Tensor]] = ()) -> bool:
    if not FUSED_LAYER or not xs[0].is_cuda or not fused_layer_ok(xs[0].shape[0], ws[0].shape[0], [x.shape[1] for x in xs]):
        return False
    if FUSED_LAYER is True:
        return True
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (*xs, *ws, *others))
    return ws[0].shape[0] > 128 and not needs_grad


class _ShimCtx:
    """What the backward passes of _ActLayerNorm / _MultiLinear read from their ctx, for reuse by _FusedLayer."""

    def __init__(self, saved, **kw):
        self.saved_tensors = tuple(saved)
        self.__dict__.update(kw)


class _FusedLayer(Function):
    """(y, yn) = Dropout(LayerNorm(LeakyReLU(sum_i x_i @ w_i^T + bias))), normalised copy: model.py:108-111, 161, 305 in ONE
    launch.  z is not kept: the backward recomputes it -- on the listed rows when the incoming gradients are row-sparse (the
    last layers under a loss on <= 3B rows: a few thousand rows through the small-product engine), else with one tall GEMM --
    and then runs the unfused pair's own backward passes."""

    @staticmethod
    def forward(ctx, bias, gamma, beta, n_terms, slope, eps, norm_eps, drop_p, seed, yn_out, want_y, want_norm, *xw):
        xs, ws = xw[:n_terms], xw[n_terms:]
        y, yn, mean, rstd = linear_act_layernorm_fwd(xs, ws, bias, gamma, beta, slope, eps, norm_eps, drop_p, seed,
                                                     want_y=want_y, want_norm=want_norm, yn_out=yn_out)
        ctx.save_for_backward(*xs, *ws, gamma, beta, mean, rstd, *([bias] if bias is not None else []),
                              *([y] if y is not None else []))
        ctx.meta = (n_terms, bias is not None, y is not None, (slope, norm_eps, drop_p, seed))
        ctx.set_materialize_grads(False)
        return y, yn

    @staticmethod
    def backward(ctx, gy, gyn):
        n_terms, has_bias, has_y, cfg = ctx.meta
        saved = list(ctx.saved_tensors)
        xs, ws = saved[:n_terms], saved[n_terms:2 * n_terms]
        gamma, beta, mean, rstd = saved[2 * n_terms:2 * n_terms + 4]
        rest = saved[2 * n_terms + 4:]
        bias = rest.pop(0) if has_bias else None
        y = rest.pop(0) if has_y else None
        none = (None,) * 12
        if gy is None and gyn is None:
            return none + (None,) * (2 * n_terms)
        n, d = xs[0].shape[0], ws[0].shape[0]
        # ---- z again: on the rows the gradients can be non-zero on, or everywhere
        rows_n, rows_y = tagged_rows(gyn), tagged_rows(gy)
        rows = union_rows(rows_y, rows_n) if (gy is None or rows_y is not None) and (gyn is None or rows_n is not None) else None
        if rows_worth_compacting(rows, n):
            ids = rows.compact_ids()
            zc = None
            for x, w in zip(xs, ws):
                zc = gemm(gather_rows_range(x, ids, 0, n), w, trans_b=True, beta=0.0 if zc is None else 1.0, out=zc,
                          bias=bias if zc is None else None)
            keep = ids >= 0
            z = torch.empty((n, d), dtype=torch.float32, device=zc.device)      # (only the listed rows are ever read)
            z.index_copy_(0, ids[keep], zc[keep])
        elif tall_ok(n, d, [x.shape[1] for x in xs], single_panel_too=True):
            z = gemm_tall(xs, (ws,), True, bias)
        else:
            z = None
            for x, w in zip(xs, ws):
                z = gemm(x, w, trans_b=True, beta=0.0 if z is None else 1.0, out=z, bias=bias if z is None else None)
        # ---- the unfused pair's own backward passes
        gz, gg, gb = _ActLayerNorm.backward(_ShimCtx((z, gamma, beta, mean, rstd, *([y] if y is not None else [])), cfg=cfg),
                                            gy, gyn)[:3]
        need = ctx.needs_input_grad
        lin = _MultiLinear.backward(_ShimCtx((*xs, *ws), n_terms=n_terms, has_bias=has_bias,
                                             needs_input_grad=(need[0], False, *need[12:12 + 2 * n_terms])), gz)
        return (lin[0], gg, gb) + none[3:] + tuple(lin[2:])


def linear_act_layernorm(xs: Sequence[torch.Tensor], ws: Sequence[torch.Tensor], bias, gamma, beta, want_norm=True,
                         slope=LEAKY_SLOPE, eps=LN_EPS, norm_eps=NORMALIZE_EPS, drop_p: float = 0.0,
                         seed: Optional[int] = None, yn_out: Optional[torch.Tensor] = None, want_y: bool = True):
    """(y, yn) of a whole aggregation layer's dense part in one launch; see _FusedLayer / FUSED_LAYER."""
    if drop_p > 0 and seed is None:
        seed = new_seed()
    return _FusedLayer.apply(bias, gamma, beta, len(xs), float(slope), float(eps), float(norm_eps), float(drop_p), seed or 0,
                             yn_out, want_y, want_norm, *xs, *ws)


# ----------------------------------------------------------------------------- a narrow layer's dense backward in one launch
NARROW_LAYER = _os.environ.get("LKG_NARROW_LAYER", "1") not in ("", "0")
"""Run the DENSE backward of a 32 -> 32 aggregation layer's Linear + LeakyReLU + LayerNorm (+ dropout + normalised copy) as one
launch (lkg_narrow_layer_bwd_f32) instead of four (row-wise backward, data gradient, weight gradient, bias column sum).  The
forward launches, and the backward under row-sparse gradients, are the unfused pair's own."""


def narrow_layer_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (NARROW_LAYER and x.is_cuda and x.dim() == 2 and x.shape[1] == 32 and tuple(w.shape) == (32, 32) and x.shape[0] >= 4096
            and _skinny(x, 32))


class _NarrowLayer(Function):
    """(y, yn) = Dropout(LayerNorm(LeakyReLU(x @ w^T + bias))) and its normalised copy (model.py:108-111, 161, 305) for the
    reference's default conv_dim of 32.  Forward: the streaming f32 product and the row-wise kernel, as unfused.  Backward:
    ONE launch when the incoming gradients are dense, else the unfused pair's own backward passes (row-sparse gradients)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, want_norm, slope, eps, norm_eps, drop_p, seed, yn_out, want_y):
        _need_gpu(x, w, gamma, beta)
        x = _f32_rows(x)
        z = gemm(x, w, trans_b=True, bias=bias)
        n, d = z.shape
        if not want_y and not want_norm:
            raise ValueError("narrow_layer: neither output wanted")
        y = torch.empty((n, d), dtype=torch.float32, device=z.device) if want_y else None
        yn = None
        if want_norm:
            yn = yn_out if yn_out is not None else torch.empty((n, d), dtype=torch.float32, device=z.device)
        mean = torch.empty(n, dtype=torch.float32, device=z.device)
        rstd = torch.empty(n, dtype=torch.float32, device=z.device)
        N.call("lkg_act_layernorm_fwd_f32", n, d, N.ptr(z), _ld(z), float(slope), N.ptr(gamma), N.ptr(beta),
               float(eps), N.ptr(y), _ld(y) if y is not None else 0, N.ptr(yn), _ld(yn) if yn is not None else 0,
               float(norm_eps), N.ptr(mean), N.ptr(rstd), float(drop_p), int(seed), _stream())
        ctx.save_for_backward(x, w, z, gamma, beta, mean, rstd, *([y] if y is not None else []))
        ctx.cfg = (slope, norm_eps, drop_p, seed)
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)
        return y, yn

    @staticmethod
    def backward(ctx, gy, gyn):
        x, w, z, gamma, beta, mean, rstd, *kept_y = ctx.saved_tensors
        y = kept_y[0] if kept_y else None
        slope, norm_eps, drop_p, seed = ctx.cfg
        none = (None,) * 13
        if gy is None and gyn is None:
            return none
        n, d = z.shape
        need = ctx.needs_input_grad
        rows_n, rows_y = tagged_rows(gyn), tagged_rows(gy)
        rows = union_rows(rows_y, rows_n) if (gy is None or rows_y is not None) and (gyn is None or rows_n is not None) else None
        gy_ = _f32_rows(gy) if gy is not None else None
        gyn_ = _f32_rows(gyn) if gyn is not None else None
        fused = (not rows_worth_compacting(rows, n) and need[0] and need[1] and gamma.data_ptr() % 16 == 0
                 and N.load().lkg_narrow_layer_bwd_ok(n, x.shape[1], d, N.ptr(x), _ld(x), N.ptr(z), _ld(z), N.ptr(y),
                                                      _ld(y) if y is not None else 0, N.ptr(gy_), _ld(gy_) if gy_ is not None else 0,
                                                      N.ptr(gyn_), _ld(gyn_) if gyn_ is not None else 0))
        if not fused:       # the unfused pair's own backward passes (they carry the row sets on)
            gz, gg, gb = _ActLayerNorm.backward(_ShimCtx((z, gamma, beta, mean, rstd, *kept_y), cfg=ctx.cfg), gy, gyn)[:3]
            lin = _MultiLinear.backward(_ShimCtx((x, w), n_terms=1, has_bias=ctx.has_bias,
                                                 needs_input_grad=(need[2], False, need[0], need[1])), gz)
            return (lin[2], lin[3], lin[0], gg, gb) + none[5:]
        gx = torch.empty((n, x.shape[1]), dtype=torch.float32, device=z.device)
        sums = torch.empty(32 * 32 + 3 * 32, dtype=torch.float32, device=z.device)     # g_w | g_bias | g_gamma | g_beta
        gw, gbias, gg, gb = sums[:1024].view(32, 32), sums[1024:1056], sums[1056:1088], sums[1088:1120]
        wc = w if w.stride(1) == 1 else w.contiguous()
        ws_floats = int(N.load().lkg_narrow_layer_bwd_workspace(n))
        ws_ = _workspace(4 * ws_floats, z.device)
        N.call("lkg_narrow_layer_bwd_f32", n, x.shape[1], d, N.ptr(x), _ld(x), N.ptr(wc), _ld(wc), N.ptr(z), _ld(z), float(slope),
               N.ptr(gamma), N.ptr(y), _ld(y) if y is not None else 0, N.ptr(mean), N.ptr(rstd), N.ptr(gy_),
               _ld(gy_) if gy_ is not None else 0, N.ptr(gyn_), _ld(gyn_) if gyn_ is not None else 0, float(norm_eps), float(drop_p),
               int(seed), N.ptr(_flags(rows_n)) if gyn_ is not None else None, N.ptr(gx), _ld(gx), N.ptr(gw),
               N.ptr(gbias) if (ctx.has_bias and need[2]) else None, N.ptr(gg), N.ptr(gb), N.ptr(ws_), ws_floats, _stream())
        return (gx, gw, gbias if (ctx.has_bias and need[2]) else None, gg, gb) + none[5:]


def narrow_layer(x, w, bias, gamma, beta, want_norm=True, slope=LEAKY_SLOPE, eps=LN_EPS, norm_eps=NORMALIZE_EPS,
                 drop_p: float = 0.0, seed: Optional[int] = None, yn_out: Optional[torch.Tensor] = None, want_y: bool = True):
    """(y, yn) of a 32 -> 32 aggregation layer's dense part; see _NarrowLayer / NARROW_LAYER."""
    if drop_p > 0 and seed is None:
        seed = new_seed()
    return _NarrowLayer.apply(x, w, bias, gamma, beta, want_norm, float(slope), float(eps), float(norm_eps), float(drop_p),
                              seed or 0, yn_out, want_y)


# ----------------------------------------------------------------------------- concat without the copy
class CatBuffer:
    """The N x (sum of widths) table that torch.cat(all_embed, dim=1) would build (model.py:309/314), allocated up
    front so that producers write their column slice directly (ld = total width) and no concat pass runs."""

    def __init__(self, n: int, widths: Sequence[int], device):
        self.widths = list(widths)
        self.offsets = [0]
        for w in self.widths:
            self.offsets.append(self.offsets[-1] + w)
        self.buf = torch.empty((n, self.offsets[-1]), dtype=torch.float32, device=device)

    def slot(self, k: int) -> torch.Tensor:
        return self.buf[:, self.offsets[k]:self.offsets[k + 1]]


def _same_view(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Do a and b address the same elements?  (A dimension of size 1 may carry any stride: a one-row view of a wider buffer
    reports the buffer's row stride, the same row produced in place may report its own width.)"""
    return (a.data_ptr() == b.data_ptr() and a.shape == b.shape
            and all(sa == sb for sa, sb, n_ in zip(a.stride(), b.stride(), a.shape) if n_ > 1))


class _AssembleCat(Function):
    @staticmethod
    def forward(ctx, holder: CatBuffer, pending, *parts):
        for k, p in enumerate(parts):
            s = holder.slot(k)
            if k in pending:                     # left unwritten on purpose: the caller completes it on demand (fill_slot)
                continue
            if not _same_view(p, s):
                s.copy_(p)                       # a part that was not produced in place (e.g. the raw entity table)
        ctx.offsets = holder.offsets
        # An ALIAS of the buffer, not the buffer itself: the parts are views of holder.buf, and a consumer below may have
        # saved one for its backward (a Linear applied to slot 0, the gate's output).  With buf itself as this node's
        # output, buf -> grad_fn -> ... -> saved view -> its base buf would be a cycle of C++ references that only a
        # backward pass breaks: every forward WITHOUT one (an evaluation run with grad enabled) kept its whole graph.
        return holder.buf.detach()

    @staticmethod
    def backward(ctx, g):
        o = ctx.offsets
        rows = tagged_rows(g)            # the column slices of a row-sparse gradient are row-sparse
        return (None, None, *[tag_rows(g[:, o[k]:o[k + 1]], rows) for k in range(len(o) - 1)])


def assemble_cat(holder: CatBuffer, parts: Sequence[torch.Tensor], pending: Sequence[int] = ()) -> torch.Tensor:
    """The concatenated table; ``pending`` slots are NOT written (their columns hold garbage until fill_slot)."""
    return _AssembleCat.apply(holder, tuple(pending), *parts)


def fill_slot(table: torch.Tensor, col0: int, src: torch.Tensor):
    """table[:, col0 : col0 + src.shape[1]] = src by a kernel of this library (no autograd version bump: the table may
    already be saved for a backward that does not read these columns)."""
    src = _f32_rows(src.detach())
    n, w = src.shape
    N.call("lkg_gather_rows_f32", n, w, N.ptr(src), _ld(src), None, None, table.data_ptr() + 4 * col0, _ld(table), _stream())


# ----------------------------------------------------------------------------- K6 gate blend
class _GateBlend(Function):
    @staticmethod
    def forward(ctx, x, gpre, zpre, out):
        _need_gpu(x, gpre, zpre)
        x, gpre, zpre = _f32_rows(x), _f32_rows(gpre), _f32_rows(zpre)
        n, d = x.shape
        if out is None:
            out = torch.empty((n, d), dtype=torch.float32, device=x.device)
        N.call("lkg_gate_blend_fwd_f32", n, d, N.ptr(x), _ld(x), N.ptr(gpre), _ld(gpre), N.ptr(zpre), _ld(zpre),
               N.ptr(out), _ld(out), _stream())
        ctx.save_for_backward(x, gpre, zpre)
        return out

    @staticmethod
    def backward(ctx, go):
        x, gpre, zpre = ctx.saved_tensors
        go = _f32_rows(go)
        n, d = x.shape
        gx, gg, gz = (torch.empty((n, d), dtype=torch.float32, device=x.device) for _ in range(3))
        N.call("lkg_gate_blend_bwd_f32", n, d, N.ptr(x), _ld(x), N.ptr(gpre), _ld(gpre), N.ptr(zpre), _ld(zpre),
               N.ptr(go), _ld(go), N.ptr(gx), _ld(gx), N.ptr(gg), _ld(gg), N.ptr(gz), _ld(gz), 0, None, _stream())
        return gx, gg, gz, None


def gate_blend(x, gpre, zpre, out: Optional[torch.Tensor] = None):
    return _GateBlend.apply(x, gpre, zpre, out)


class _FusedGate(Function):
    """The whole literal gate (gate.py:22-28 / 45-51) in ONE launch: the two projections g([x | lits]) and
    z(x, lits) as one stacked tall GEMM over the K-panels (x, literal panels), tanh / sigmoid / blend in its epilogue,
    the result written straight into ``out`` (a column slot of the concatenated table).  Inputs are read once; for the
    backward only tanh(g) and sigmoid(z) are kept.
    Backward: one blend kernel (g_x direct term, g_gpre, g_zpre side by side), ONE tall GEMM for the data gradient
    [g_gpre | g_zpre] . [W_g[:, :d] ; W_e] accumulated onto the direct term, the weight gradients as long-k products."""

    @staticmethod
    def forward(ctx, out, bg, bz, n_lit, grad_mode, x, *rest):
        lits = rest[:n_lit]
        wgs = rest[n_lit:2 * n_lit + 1]           # g.weight column panels: [x part, literal parts ...]
        wzs = rest[2 * n_lit + 1:]                # gate_ent.weight, gate_*_lit.weight ...
        _need_gpu(x, *lits, *wgs, *wzs)
        x = _f32_rows(x)
        n, d = x.shape
        panels = (x,) + tuple(_f32_rows(l) for l in lits)
        if out is None:
            out = torch.empty((n, d), dtype=torch.float32, device=x.device)
        training = grad_mode and any(ctx.needs_input_grad)     # (needs_input_grad ignores no_grad; forward runs with grad off)
        keep = None
        if training:
            keep = (torch.empty((n, d), dtype=torch.float32, device=x.device),
                    torch.empty((n, d), dtype=torch.float32, device=x.device))
        bias = torch.cat([bg, bz]) if bg is not None else None
        # the literals are constants of the model: their row maxima are computed once and cached on the tensor
        rm = row_absmax(x)
        for l in panels[1:]:
            cached = getattr(l, "_lkg_rowmax", None)
            if cached is None or cached[0] != l._version or cached[1].device != l.device:
                cached = (l._version, row_absmax(l))
                try:
                    l._lkg_rowmax = cached
                except AttributeError:
                    pass
            torch.maximum(rm, cached[1], out=rm)             # (n floats: bookkeeping, not a pass over the table)
        gemm_tall(panels, (wgs, wzs), True, bias, out=out, rowmax=rm, gate_x=x, keep=keep)
        ctx.n_lit = n_lit
        ctx.has_bias = bg is not None
        if training:
            ctx.save_for_backward(x, keep[0], keep[1], *panels[1:], *wgs, *wzs)
            # column maxima of the (constant) wide literal tables, once per table: column scales of their weight gradient
            ctx.lit_colmax = [tagged_colmax(l, cache=True) if (l.shape[1] > 4 and n >= TALL_MIN_ROWS) else None
                              for l in panels[1:]]
        return out

    @staticmethod
    def backward(ctx, go):
        nl = ctx.n_lit
        sv = ctx.saved_tensors
        x, g, z = sv[0], sv[1], sv[2]
        lits, wgs, wzs = sv[3:3 + nl], sv[3 + nl:4 + 2 * nl], sv[4 + 2 * nl:]
        rows = tagged_rows(go)
        go = _f32_rows(go)
        n, d = x.shape
        if rows_worth_compacting(rows, n):
            return _FusedGate._backward_on_rows(ctx, go, rows, x, g, z, lits, wgs, wzs)
        gx = torch.empty((n, d), dtype=torch.float32, device=x.device)
        gpz = torch.empty((n, 2 * d), dtype=torch.float32, device=x.device)      # [g_gpre | g_zpre]
        ggp, gzp = gpz[:, :d], gpz[:, d:]
        need = ctx.needs_input_grad           # (out, bg, bz, n_lit, grad_mode, x, lits..., wgs..., wzs...)
        tall = need[5] and tall_ok(n, d, (d, d))
        rm = torch.empty(n, dtype=torch.float32, device=x.device) if tall else None   # row scale of the data-gradient GEMM
        panels = (x,) + tuple(lits)
        base = 6 + nl
        want = [need[base + i] or need[base + nl + 1 + i] for i in range(nl + 1)]
        want_bias = ctx.has_bias and (need[1] or need[2])
        # the blend kernel can emit, while it writes [g_gpre | g_zpre], everything the weight / bias gradients need beside the
        # wide products: the bias sums, ONE narrow literal panel's weight gradient (the numeric literals), and the column
        # maxima of [g_gpre | g_zpre] and of x -- the column scales of the fp16 long-k product (lkg_gemm_wgrad_f32)
        narrow = [i for i in range(1, nl + 1) if want[i] and panels[i].shape[1] <= 4]
        stats = None
        if (_WGRAD_ENGINE == "longk" and n >= TALL_MIN_ROWS and d % 4 == 0 and d <= 1024 and len(narrow) <= 1
                and all(t_.data_ptr() % 16 == 0 and _ld(t_) % 4 == 0 for t_ in (x, g, z, go))):
            wn = _f32_rows(panels[narrow[0]]) if narrow else None
            n_w = wn.shape[1] if wn is not None else 0
            n_stats = (5 + 2 * n_w) * d
            ws = _workspace(4 * 1024 * n_stats, x.device).view(torch.float32)
            stats = torch.empty(n_stats, dtype=torch.float32, device=x.device)
            N.call("lkg_gate_blend_bwd_stats_f32", n, d, N.ptr(x), _ld(x), N.ptr(g), _ld(g), N.ptr(z), _ld(z), N.ptr(go),
                   _ld(go), N.ptr(gx), _ld(gx), N.ptr(ggp), _ld(ggp), N.ptr(gzp), _ld(gzp), 1, N.ptr(rm), N.ptr(wn),
                   _ld(wn) if wn is not None else 0, n_w, N.ptr(ws), ws.numel(), N.ptr(stats), _stream())
        else:
            N.call("lkg_gate_blend_bwd_f32", n, d, N.ptr(x), _ld(x), N.ptr(g), _ld(g), N.ptr(z), _ld(z), N.ptr(go),
                   _ld(go), N.ptr(gx), _ld(gx), N.ptr(ggp), _ld(ggp), N.ptr(gzp), _ld(gzp), 1, N.ptr(rm), _stream())
        g_x = None
        if need[5]:
            if tall:
                g_x = gemm_tall((ggp, gzp), ((wgs[0], wzs[0]),), False, beta=1.0, out=gx, rowmax=rm)
            else:
                g_x = gemm(ggp, wgs[0], beta=1.0, out=gx)
                g_x = gemm(gzp, wzs[0], beta=1.0, out=g_x)
        # weight and bias gradients of both projections from the side-by-side buffer: [g_gpre | g_zpre]^T @ panel is
        # [2d x k]: rows :d are g's, rows d: gate_*'s (one long-k product per panel reads the panel once)
        if stats is not None:
            gb = stats[:2 * d] if want_bias else None
            cm_g, cm_x = stats[2 * d:4 * d], stats[4 * d:5 * d]
            gws = [None] * (nl + 1)
            for i in range(nl + 1):
                if not want[i]:
                    continue
                if narrow and i == narrow[0]:
                    gws[i] = stats[5 * d:].view(n_w, 2 * d).t()          # [2d x n_w]
                    continue
                cm_p = cm_x if i == 0 else ctx.lit_colmax[i - 1]        # (the literal tables are constants: cached in forward)
                gws[i] = gemm_wgrad(gpz, panels[i], cm_g, cm_p) if cm_p is not None else gemm(gpz, panels[i], trans_a=True)
        else:
            gws, gb = weight_grads(gpz, panels, want, want_bias)
        g_wg = [gws[i][:d] if need[base + i] else None for i in range(nl + 1)]
        g_wz = [gws[i][d:] if need[base + nl + 1 + i] else None for i in range(nl + 1)]
        gb_g = gb[:d] if (gb is not None and need[1]) else None
        gb_z = gb[d:] if (gb is not None and need[2]) else None
        return (None, gb_g, gb_z, None, None, g_x, *([None] * nl), *g_wg, *g_wz)


def _fused_gate_backward_on_rows(ctx, go, rows: RowSet, x, g, z, lits, wgs, wzs):
    """go is zero outside ``rows`` (a one-layer model: the loss's rows and the rows one transpose SpMM reaches from them, a few
    percent of the entities): every product of the gate's backward is the same product over the listed rows -- x, tanh(g),
    sigmoid(z), the literals and go gathered once -- and the gradient of x is their scatter into a table kept all-zero elsewhere.
    Only zero addends are dropped."""
    nl = ctx.n_lit
    n, d = x.shape
    need = ctx.needs_input_grad           # (out, bg, bz, n_lit, grad_mode, x, lits..., wgs..., wzs...)
    ids = rows.compact_ids()              # every row once, -1 padding (gathers as zeros: go = 0 there, so every gradient is 0)
    nc = ids.numel()
    goc = gather_rows_range(go, ids, 0, n)
    xc, gc, zc = (gather_rows_range(t_, ids, 0, n) for t_ in (x, g, z))
    gxc = torch.empty((nc, d), dtype=torch.float32, device=x.device)
    gpz = torch.empty((nc, 2 * d), dtype=torch.float32, device=x.device)      # [g_gpre | g_zpre]
    ggp, gzp = gpz[:, :d], gpz[:, d:]
    N.call("lkg_gate_blend_bwd_f32", nc, d, N.ptr(xc), d, N.ptr(gc), d, N.ptr(zc), d, N.ptr(goc), d, N.ptr(gxc), d, N.ptr(ggp),
           _ld(ggp), N.ptr(gzp), _ld(gzp), 1, None, _stream())
    g_x = None
    if need[5]:
        gemm(ggp, wgs[0], beta=1.0, out=gxc)
        gemm(gzp, wzs[0], beta=1.0, out=gxc)
        g_x = zero_table_for(rows, n, d, x.device, "g_gate_x")
        N.call("lkg_scatter_add_rows_range_f32", nc, d, N.ptr(gxc), d, N.ptr(ids), 0, n, N.ptr(g_x), _ld(g_x), _stream())
    base = 6 + nl
    want = [need[base + i] or need[base + nl + 1 + i] for i in range(nl + 1)]
    want_bias = ctx.has_bias and (need[1] or need[2])
    panels = [xc] + [gather_rows_range(_f32_rows(l), ids, 0, n) if want[i + 1] else None for i, l in enumerate(lits)]
    gws = [gemm(gpz, panels[i], trans_a=True) if want[i] else None for i in range(nl + 1)]
    gb = colsum(gpz) if want_bias else None
    g_wg = [gws[i][:d] if need[base + i] else None for i in range(nl + 1)]
    g_wz = [gws[i][d:] if need[base + nl + 1 + i] else None for i in range(nl + 1)]
    gb_g = gb[:d] if (gb is not None and need[1]) else None
    gb_z = gb[d:] if (gb is not None and need[2]) else None
    return (None, gb_g, gb_z, None, None, g_x, *([None] * nl), *g_wg, *g_wz)


_FusedGate._backward_on_rows = staticmethod(_fused_gate_backward_on_rows)


def fused_gate(x, lits: Sequence[torch.Tensor], wg_panels: Sequence[torch.Tensor], wz_panels: Sequence[torch.Tensor],
               bias_g, bias_z, out: Optional[torch.Tensor] = None):
    """out = (1 - sigmoid(z)) x + sigmoid(z) tanh(g),  g = [x | lits] wg^T + bias_g,  z = x wz_0^T + sum lits wz_i^T + bias_z"""
    return _FusedGate.apply(out, bias_g, bias_z, len(lits), torch.is_grad_enabled(), x, *lits, *wg_panels, *wz_panels)


def gate_fusable(x, lits, d) -> bool:
    return tall_ok(x.shape[0], 2 * d, [x.shape[1]] + [l.shape[1] for l in lits])


# ----------------------------------------------------------------------------- K8 TransE scoring
class _TransELoss(Function):
    """model_bce.py:329-368 on table rows."""

    @staticmethod
    def forward(ctx, emb, relemb, h, r, pt, nt, lam, keep, sparse_rows):
        _need_gpu(emb, relemb, h, r, pt, nt)
        emb, relemb = _f32_rows(emb), _f32_rows(relemb)
        ctx.sparse_rows = bool(sparse_rows)
        if emb.shape[1] != relemb.shape[1]:
            raise ValueError(f"TransE scoring needs entity-side width == relation_dim ({emb.shape[1]} vs "
                             f"{relemb.shape[1]})")
        h, r, pt, nt = _i64(h), _i64(r), _i64(pt), _i64(nt)
        b = h.numel()
        buf = torch.empty((4, b), dtype=torch.float32, device=emb.device)
        loss = torch.empty((), dtype=torch.float32, device=emb.device)
        N.call("lkg_transe_score_fwd_f32", b, emb.shape[1], N.ptr(emb), _ld(emb), N.ptr(relemb), _ld(relemb),
               N.ptr(h), N.ptr(r), N.ptr(pt), N.ptr(nt), N.ptr(buf[0]), N.ptr(buf[1]), N.ptr(buf[2]), N.ptr(buf[3]),
               _stream())
        N.call("lkg_loss_reduce_f32", b, N.ptr(buf[3]), N.ptr(buf[2]), float(lam), N.ptr(loss), _stream())
        ctx.save_for_backward(emb, relemb, h, r, pt, nt, buf)
        ctx.lam = lam
        if keep is not None:
            keep["pos"], keep["neg"] = buf[0], buf[1]
        return loss

    @staticmethod
    def backward(ctx, gl):
        emb, relemb, h, r, pt, nt, buf = ctx.saved_tensors
        gl = gl.contiguous().float()
        g_emb = _loss_grad_table(emb, ctx.sparse_rows, h, pt, nt)
        g_rel = torch.zeros_like(relemb, memory_format=torch.contiguous_format)
        N.call("lkg_transe_score_bwd_f32", h.numel(), emb.shape[1], N.ptr(emb), _ld(emb), N.ptr(relemb), _ld(relemb),
               N.ptr(h), N.ptr(r), N.ptr(pt), N.ptr(nt), N.ptr(buf[0]), N.ptr(buf[1]), float(ctx.lam), N.ptr(gl),
               N.ptr(g_emb), _ld(g_emb), N.ptr(g_rel), _ld(g_rel), _stream())
        return g_emb, g_rel, None, None, None, None, None, None, None


def transe_loss(emb, relemb, h, r, pos_t, neg_t, lam, keep=None, sparse_rows: bool = False):
    """sparse_rows: the gradient of ``emb`` is consumed inside the backward pass by this package's Functions only
    (the model's concatenated table), so it may be the shared all-zero table of _RowScratch instead of a fresh fill."""
    return _TransELoss.apply(emb, relemb, h, r, pos_t, neg_t, lam, keep, sparse_rows)


# ----------------------------------------------------------------------------- K7+K8 TransR scoring
def _grouped(mode, seg, max_len, a, b, out, m, n, k, trans_a, trans_b, beta, stride_b=0, stride_c=0, b_period=0):
    N.call("lkg_grouped_gemm_f32", mode, seg.numel() - 1, N.ptr(seg), max_len, int(trans_a), int(trans_b), m, n, k,
           1.0, N.ptr(a), _ld(a), N.ptr(b), b.stride(-2), stride_b, int(b_period), float(beta), N.ptr(out),
           out.stride(-2), stride_c, _stream())


class GroupedCheck:
    """Is the batch whole groups of ``group_size`` consecutive rows with identical (h, r, t+) -- the layout of
    DataLoader.generate_kg_batch (dataloader.py:318-330)?  The answer decides tensor shapes, so the host has to wait for
    it -- but not where it asks: the check kernel and the copy of its verdict to pinned memory are queued when the object
    is made (BEFORE the encoder), ``result()`` waits for that event only (just before the loss: the encoder's launches
    are queued by then and the device never idles behind the wait)."""

    def __init__(self, h, r, pos_t, group_size: int):
        self.answer = None
        b = h.numel()
        if group_size <= 1 or b == 0 or b % group_size:
            self.answer = False
            return
        _need_gpu(h, r, pos_t)
        h, r, pos_t = _i64(h), _i64(r), _i64(pos_t)
        bad = torch.empty(1, dtype=torch.int32, device=h.device)
        N.call("lkg_check_grouped_i64", b, int(group_size), N.ptr(h), N.ptr(r), N.ptr(pos_t), N.ptr(bad), _stream())
        self.host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        self.host.copy_(bad, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def result(self) -> bool:
        if self.answer is None:
            self.event.synchronize()
            self.answer = int(self.host[0]) == 0
        return self.answer


def is_grouped_batch(h, r, pos_t, group_size: int) -> bool:
    """GroupedCheck, asked and answered on the spot (one host sync)."""
    return GroupedCheck(h, r, pos_t, group_size).result()


class _TransRLoss(Function):
    """model.py:364-428.  The batch is grouped by relation so W_r = gat_trans_M[r] is applied by ONE
    grouped MFMA GEMM per operand; the B x C x D gather of the reference never exists.  With ``group`` = K > 1 (a batch
    of B/K groups of K rows sharing (h, r, t+), checked by the caller) the head and the positive tail are gathered and
    projected ONCE per group: 2 (2 + K) / K B C D flops instead of 6 B C D."""

    @staticmethod
    def forward(ctx, emb, relemb, trans_m, h, r, pt, nt, lam, keep, group, sparse_rows, slot0):
        _need_gpu(emb, relemb, trans_m, h, r, pt, nt)
        ctx.sparse_rows = bool(sparse_rows)
        emb, relemb = _f32_rows(emb), _f32_rows(relemb)
        trans_m = trans_m.contiguous()
        n_rel, c, dout = trans_m.shape
        if emb.shape[1] != c or relemb.shape[1] != dout:
            raise ValueError("gat_trans_M shape does not match the embedding widths")
        b = h.numel()
        k = int(group)
        if k < 1 or b % k:
            raise ValueError(f"batch of {b} triples is not a whole number of groups of {k}")
        n_g = b // k
        # one (h, r, t+) per group; the negatives stay one per triple
        hg, rg, pg, nt = _i64(h[::k]), _i64(r[::k]), _i64(pt[::k]), _i64(nt)
        dev = emb.device
        _Deferred.poll()
        perm = torch.empty(n_g, dtype=torch.int32, device=dev)           # groups in relation order
        seg = torch.empty(n_rel + 2, dtype=torch.int32, device=dev)      # [n_rel + 1] offsets + the bad-key counter
        N.call("lkg_group_by_key_i64", n_g, n_rel, N.ptr(rg), N.ptr(perm), N.ptr(seg), seg.data_ptr() + 4 * (n_rel + 1),
               _stream())
        _Deferred.watch(seg[n_rel + 1:], IndexError,
                        "pre_training batch holds {n} relation id(s) outside [0, n_relations) (gat_trans_M[r], "
                        "model.py:372)")
        seg = seg[:n_rel + 1]
        if k > 1:    # row order of the negatives: they follow their group
            perm_n = torch.empty(b, dtype=torch.int32, device=dev)
            seg_n = torch.empty(n_rel + 1, dtype=torch.int32, device=dev)
            N.call("lkg_expand_groups_i32", n_g, k, n_rel + 1, N.ptr(perm), N.ptr(seg), N.ptr(perm_n), N.ptr(seg_n),
                   _stream())
        else:
            perm_n, seg_n = perm, seg
        rs = torch.empty_like(rg)
        N.call("lkg_gather_i64", n_g, N.ptr(rg), N.ptr(perm), N.ptr(rs), _stream())
        # gathered rows (relation order) and their projections: [h rows | t+ rows | t- rows]
        x = torch.empty((2 * n_g + b, c), dtype=torch.float32, device=dev)
        p = torch.empty((2 * n_g + b, dout), dtype=torch.float32, device=dev)
        parts = ((hg, perm, seg, 0, n_g), (pg, perm, seg, n_g, n_g), (nt, perm_n, seg_n, 2 * n_g, b))
        # slot0: the first columns of emb are not filled in (assemble_cat's pending slot): they are read from the table
        # they would be a copy of (the raw entity table) -- the copy of N rows is never made for the <= 3B rows read here
        w0 = 0
        if slot0 is not None:
            slot0 = _f32_rows(slot0)
            w0 = slot0.shape[1]
        for ids, pm, sg, off, rows in parts:
            if w0:
                N.call("lkg_gather_rows_f32", rows, w0, N.ptr(slot0), _ld(slot0), N.ptr(ids), N.ptr(pm), N.ptr(x[off:]), c,
                       _stream())
            N.call("lkg_gather_rows_f32", rows, c - w0, emb.data_ptr() + 4 * w0, _ld(emb), N.ptr(ids), N.ptr(pm),
                   x[off:].data_ptr() + 4 * w0, c, _stream())
        # the three row blocks are each sorted by relation: 3 R row ranges over the R matrices, ONE grouped launch
        seg_all = torch.cat([seg, seg[1:] + n_g, seg_n[1:] + 2 * n_g])
        _grouped(1, seg_all, max(b, n_g), x, trans_m, p, 0, dout, c, False, False, 0.0, stride_b=c * dout, b_period=n_rel)
        buf = torch.empty((4, b), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        N.call("lkg_dense_score_fwd_f32", b, k, dout, N.ptr(p), N.ptr(p[n_g:]), N.ptr(p[2 * n_g:]), dout, N.ptr(relemb),
               _ld(relemb), N.ptr(rs), N.ptr(buf[0]), N.ptr(buf[1]), N.ptr(buf[2]), N.ptr(buf[3]), _stream())
        N.call("lkg_loss_reduce_f32", b, N.ptr(buf[3]), N.ptr(buf[2]), float(lam), N.ptr(loss), _stream())
        ctx.save_for_backward(emb, relemb, trans_m, hg, pg, nt, rs, perm, seg, perm_n, seg_n, x, p, buf, seg_all)
        ctx.lam, ctx.k = lam, k
        if keep is not None:   # scores back in the caller's triple order
            inv = torch.empty(b, dtype=torch.int64, device=dev)
            inv[perm_n.long()] = torch.arange(b, device=dev)
            keep["pos"], keep["neg"] = buf[0][inv], buf[1][inv]
        return loss

    @staticmethod
    def backward(ctx, gl):
        emb, relemb, trans_m, hg, pg, nt, rs, perm, seg, perm_n, seg_n, x, p, buf, seg_all = ctx.saved_tensors
        n_rel, c, dout = trans_m.shape
        b, k = nt.numel(), ctx.k
        n_g = b // k
        dev = emb.device
        gl = gl.contiguous().float()
        gp = torch.empty_like(p)
        g_rel = torch.zeros_like(relemb, memory_format=torch.contiguous_format)
        N.call("lkg_dense_score_bwd_f32", b, k, dout, N.ptr(p), N.ptr(p[n_g:]), N.ptr(p[2 * n_g:]), dout, N.ptr(relemb),
               _ld(relemb), N.ptr(rs), N.ptr(buf[0]), N.ptr(buf[1]), float(ctx.lam), N.ptr(gl), N.ptr(gp),
               N.ptr(gp[n_g:]), N.ptr(gp[2 * n_g:]), dout, N.ptr(g_rel), _ld(g_rel), _stream())
        g_w = torch.empty_like(trans_m)
        g_emb = _loss_grad_table(emb, ctx.sparse_rows, hg, pg, nt)
        gx = torch.empty((2 * n_g + b, c), dtype=torch.float32, device=dev)
        # g_X = G W_r^T for the three row blocks in one grouped launch, scattered back to the table rows below
        _grouped(1, seg_all, max(b, n_g), gp, trans_m, gx, 0, c, dout, False, True, 0.0, stride_b=c * dout, b_period=n_rel)
        parts = ((hg, perm, seg, 0, n_g), (pg, perm, seg, n_g, n_g), (nt, perm_n, seg_n, 2 * n_g, b))
        # g_W[r] = X_h^T G_h + X_+^T G_+ + X_-^T G_- over relation r's rows.  The three terms nearly cancel when the table's rows
        # share a large common part v (G_h = -(G_+ + sum G_-) up to the regulariser): the reference adds them PER SAMPLE before it
        # sums over the batch (model.py:391-397 under autograd); block after block they are rounded relative to the uncancelled
        # partial sums -- 60 x the fp32 reference's own error with one relation and 683 groups (tests/test_gpu_fuzz.py, seed 44053).
        # So the common part is taken out first:  g_W[r] = sum_i (x_i - v)^T g_i  +  v^T (sum_i g_i),  v = the mean row, and the second
        # sum is taken per GROUP first (head + positive + its negatives: what is left of their cancellation), then over the groups.
        v = x.mean(0, keepdim=True) if x.shape[0] else torch.zeros((1, c), dtype=torch.float32, device=dev)
        xc = x - v
        for i, (ids, pm, sg, off, rows) in enumerate(parts):
            # g_W[r] (+)= (X_r - v)^T G_r
            _grouped(2, sg, rows, xc[off:off + rows], gp[off:off + rows], g_w, c, dout, 0, True, False, 0.0 if i == 0 else 1.0,
                     stride_c=c * dout)
            N.call("lkg_scatter_add_rows_f32", rows, c, N.ptr(gx[off:]), c, N.ptr(ids), N.ptr(pm), N.ptr(g_emb),
                   _ld(g_emb), _stream())
        per_group = gp[:n_g] + gp[n_g:2 * n_g] + gp[2 * n_g:].view(n_g, k, dout).sum(1)         # (rows are in relation order: rs)
        per_rel = torch.zeros((n_rel, dout), dtype=torch.float32, device=dev).index_add_(0, rs, per_group)
        g_w.addcmul_(v.view(1, c, 1), per_rel.view(n_rel, 1, dout))
        return g_emb, g_rel, g_w, None, None, None, None, None, None, None, None, None


def transr_loss(emb, relemb, trans_m, h, r, pos_t, neg_t, lam, keep=None, group: int = 1, sparse_rows: bool = False,
                slot0: Optional[torch.Tensor] = None):
    """slot0: emb's first slot0.shape[1] columns are pending (assemble_cat) and are read from this table instead."""
    return _TransRLoss.apply(emb, relemb, trans_m, h, r, pos_t, neg_t, lam, keep, group, sparse_rows,
                             slot0.detach() if slot0 is not None else None)


# ----------------------------------------------------------------------------- f1 fine-tuning head
class _DotLoss(Function):
    """model.py:316-348: dot-product BPR loss on table rows."""

    @staticmethod
    def forward(ctx, emb, h, pt, nt, lam, sparse_rows):
        _need_gpu(emb, h, pt, nt)
        emb = _f32_rows(emb)
        ctx.sparse_rows = bool(sparse_rows)
        h, pt, nt = _i64(h), _i64(pt), _i64(nt)
        b = h.numel()
        buf = torch.empty((4, b), dtype=torch.float32, device=emb.device)
        loss = torch.empty((), dtype=torch.float32, device=emb.device)
        N.call("lkg_dot_score_fwd_f32", b, emb.shape[1], N.ptr(emb), _ld(emb), N.ptr(h), N.ptr(pt), N.ptr(nt),
               N.ptr(buf[0]), N.ptr(buf[1]), N.ptr(buf[2]), N.ptr(buf[3]), _stream())
        N.call("lkg_loss_reduce_f32", b, N.ptr(buf[3]), N.ptr(buf[2]), float(lam), N.ptr(loss), _stream())
        ctx.save_for_backward(emb, h, pt, nt, buf)
        ctx.lam = lam
        return loss

    @staticmethod
    def backward(ctx, gl):
        emb, h, pt, nt, buf = ctx.saved_tensors
        gl = gl.contiguous().float()
        g_emb = _loss_grad_table(emb, ctx.sparse_rows, h, pt, nt)
        N.call("lkg_dot_score_bwd_f32", h.numel(), emb.shape[1], N.ptr(emb), _ld(emb), N.ptr(h), N.ptr(pt), N.ptr(nt),
               N.ptr(buf[0]), N.ptr(buf[1]), float(ctx.lam), N.ptr(gl), N.ptr(g_emb), _ld(g_emb), _stream())
        return g_emb, None, None, None, None, None


def dot_loss(emb, h, pos_t, neg_t, lam, sparse_rows: bool = False):
    return _DotLoss.apply(emb, h, pos_t, neg_t, lam, sparse_rows)


class _GatherPair(Function):
    """(table[ids_a], table[ids_b]) for the pair head (model.py:506-512) with ONE backward: both row gradients scattered
    into one table -- the shared all-zero table of _RowScratch when the caller says the gradient stays inside the package,
    tagged with the rows it touches, so that everything below runs on those rows (two separate index ops would hand
    autograd two dense N x C tables to add, and the sum carries no row set)."""

    @staticmethod
    def forward(ctx, table, ids_a, ids_b, sparse_rows):
        _need_gpu(table, ids_a, ids_b)
        ids_a, ids_b = _i64(ids_a), _i64(ids_b)
        ctx.save_for_backward(ids_a, ids_b)
        ctx.meta = (tuple(table.shape), table.device, bool(sparse_rows))
        ctx.set_materialize_grads(False)
        return gather_rows(table, ids_a), gather_rows(table, ids_b)

    @staticmethod
    def backward(ctx, ga, gb):
        ids_a, ids_b = ctx.saved_tensors
        shape, device, sparse = ctx.meta
        if sparse:
            ent = _RowScratch.acquire(shape[0], shape[1], device)
            ent.mark(ids_a, ids_b)
            g_tab = ent.table(RowSet(ent.flags, [ids_a, ids_b]))
        else:
            g_tab = torch.zeros(shape, dtype=torch.float32, device=device)
        for g, ids in ((ga, ids_a), (gb, ids_b)):
            if g is None:
                continue
            g = _f32_rows(g)
            N.call("lkg_scatter_add_rows_f32", ids.numel(), shape[1], N.ptr(g), _ld(g), N.ptr(ids), None, N.ptr(g_tab),
                   _ld(g_tab), _stream())
        return g_tab, None, None, None


def gather_rows_pair(table: torch.Tensor, ids_a: torch.Tensor, ids_b: torch.Tensor, sparse_rows: bool = False):
    return _GatherPair.apply(table, ids_a, ids_b, sparse_rows)


# ----------------------------------------------------------------------------- f1 MLP head
class _ReluBatchNorm(Function):
    """y = BatchNorm1d(relu(z)) (model.py:515-516); updates the running buffers in training mode."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, training, momentum, eps):
        _need_gpu(z, gamma, beta, running_mean, running_var)
        z = _f32_rows(z)
        n, d = z.shape
        y = torch.empty((n, d), dtype=torch.float32, device=z.device)
        mean = torch.empty(d, dtype=torch.float32, device=z.device)
        invstd = torch.empty(d, dtype=torch.float32, device=z.device)
        N.call("lkg_relu_batchnorm_fwd_f32", n, d, N.ptr(z), _ld(z), N.ptr(gamma), N.ptr(beta), float(eps),
               int(training), float(momentum), N.ptr(running_mean), N.ptr(running_var), N.ptr(y), _ld(y),
               N.ptr(mean), N.ptr(invstd), _stream())
        ctx.save_for_backward(z, gamma, mean, invstd)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, gy):
        z, gamma, mean, invstd = ctx.saved_tensors
        gy = _f32_rows(gy)
        n, d = z.shape
        gz = torch.empty((n, d), dtype=torch.float32, device=z.device)
        gg = torch.empty(d, dtype=torch.float32, device=z.device)
        gb = torch.empty(d, dtype=torch.float32, device=z.device)
        N.call("lkg_relu_batchnorm_bwd_f32", n, d, N.ptr(z), _ld(z), N.ptr(gamma), N.ptr(mean), N.ptr(invstd),
               int(ctx.training), N.ptr(gy), _ld(gy), N.ptr(gz), _ld(gz), N.ptr(gg), N.ptr(gb), _stream())
        return gz, gg, gb, None, None, None, None, None


def relu_batchnorm(z, bn: "torch.nn.BatchNorm1d"):
    """bn(relu(z)) with bn's parameters / buffers / mode; bumps num_batches_tracked like nn.BatchNorm1d."""
    if bn.training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _ReluBatchNorm.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, momentum, bn.eps)


# ----------------------------------------------------------------------------- f1 heads
def gather_rows(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """table[ids] without autograd (inference heads, model.py:475-476)."""
    _need_gpu(table, ids)
    table, ids = _f32_rows(table), _i64(ids)
    out = torch.empty((ids.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    N.call("lkg_gather_rows_f32", ids.numel(), table.shape[1], N.ptr(table), _ld(table), N.ptr(ids), None, N.ptr(out),
           table.shape[1], _stream())
    return out
