"""The one place that talks to ``torch.distributed`` for the multi-GPU path (the reference's only
scaling device is ``nn.DataParallel``, main_pretraining.py:69-71).

``Transport`` hides WHICH collective library moves the bytes: backend "nccl" (= RCCL over xGMI) takes device tensors as
they are; "gloo" moves host memory only, so a device tensor is staged through the host inside the transport.  Every call
returns a ``Pending`` and the caller waits where it needs the bytes -- the SAME control flow under both backends, so the
gloo rehearsals (tests/test_*_gloo.py on CPU tensors, tests/test_*_gpu.py on one GPU's device tensors) walk the code path
an RCCL run walks; no call site branches on the backend.

``InFlight`` makes a pipelined pass exception-safe: it owns the side streams and every queued ``Pending`` of the pass,
and on the way out -- normal or by exception -- joins the side streams into the caller's stream and waits every
collective still queued.  On an exception it also drains the device before the exception travels on: the tensors the
queued kernels read and write are released while the exception unwinds, and a kernel must never outlive its operands.
"""
from __future__ import annotations

import contextlib
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

ERROR_WAIT_SECONDS = 60.0      # a collective waited for on an error path: bounded (the peer may never send)


class Pending:
    """A transfer in flight.  ``wait()`` makes its result usable on the CURRENT stream (under RCCL a stream dependency:
    the host does not block) and returns ``out``."""

    def __init__(self, works: Sequence, out=None, after: Optional[Callable[[], None]] = None):
        self.works, self.out, self.after = list(works), out, after

    def wait(self, timeout: Optional[float] = None):
        works, self.works = self.works, []
        for w in works:
            if w is None:
                continue
            if timeout is not None:
                import datetime
                w.wait(datetime.timedelta(seconds=timeout))
            else:
                w.wait()
        after, self.after = self.after, None
        if after is not None:
            after()
        return self.out

    @property
    def done(self) -> bool:
        return not self.works and self.after is None


class Transport:
    def __init__(self, group=None):
        self.group = group
        self.bytes_sent = 0            # payload bytes this rank handed to the collective library

    # ---- facts
    @property
    def multi(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0

    def backend(self) -> str:
        return dist.get_backend(self.group)

    def _staged(self, t: torch.Tensor) -> bool:
        """gloo moves host memory only: a device tensor travels through a host copy (inside this class, nowhere else)"""
        return t.is_cuda and self.backend() == "gloo"

    def control_device(self, like: torch.device) -> torch.device:
        """where small integer control messages (counts) live: the data's device under RCCL, the host under gloo"""
        return like if self.backend() != "gloo" else torch.device("cpu")

    # ---- point to point
    def p2p(self, sends: Sequence[Tuple[torch.Tensor, int]], recvs: Sequence[Tuple[torch.Tensor, int]]) -> Pending:
        """One group of sends (tensor, peer) and receives (destination, peer), handed to the library together; ordered
        behind the CURRENT stream.  Destinations may be views; they hold the bytes after ``wait()``."""
        ops_, copies = [], []
        for t, peer in sends:
            self.bytes_sent += t.numel() * t.element_size()
            src = t.contiguous()
            ops_.append(dist.P2POp(dist.isend, src.cpu() if self._staged(src) else src, peer, self.group))
        for dst, peer in recvs:
            if self._staged(dst) or not dst.is_contiguous():
                buf = torch.empty(dst.shape, dtype=dst.dtype, device="cpu" if self._staged(dst) else dst.device)
                copies.append((buf, dst))
            else:
                buf = dst
            ops_.append(dist.P2POp(dist.irecv, buf, peer, self.group))
        if not ops_:
            return Pending([])

        def land():
            for buf, dst in copies:
                dst.copy_(buf)
        return Pending(dist.batch_isend_irecv(ops_), None, land if copies else None)

    # ---- collectives
    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, out_splits: List[int], in_splits: List[int],
                   count: bool = True) -> Pending:
        """rows of ``inp`` sorted by destination -> rows of ``out`` sorted by source (splits in rows)"""
        if count:
            per_row = (inp.numel() // max(inp.shape[0], 1)) * inp.element_size() if inp.shape[0] else 0
            self.bytes_sent += (sum(in_splits) - in_splits[self.rank]) * per_row
        inp = inp.contiguous()
        if self._staged(inp):
            host = torch.empty(out.shape, dtype=out.dtype)
            w = dist.all_to_all_single(host, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                                       group=self.group, async_op=True)
            return Pending([w], out, lambda: out.copy_(host))
        w = dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group,
                                   async_op=True)
        return Pending([w], out)

    def all_gather(self, x: torch.Tensor) -> Pending:
        """[R, ...] blocks -> [world * R, ...]"""
        x = x.contiguous()
        world = self.world
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        self.bytes_sent += (world - 1) * x.numel() * x.element_size()
        if self._staged(x):
            host = torch.empty(out.shape, dtype=out.dtype)
            w = dist.all_gather_into_tensor(host, x.cpu(), group=self.group, async_op=True)
            return Pending([w], out, lambda: out.copy_(host))
        return Pending([dist.all_gather_into_tensor(out, x, group=self.group, async_op=True)], out)

    def all_reduce(self, x: torch.Tensor) -> Pending:
        """sum over ranks, in place"""
        world = self.world
        self.bytes_sent += 2 * (world - 1) * x.numel() * x.element_size() // max(world, 1)
        if self._staged(x) or not x.is_contiguous():
            host = x.cpu() if x.is_cuda and self.backend() == "gloo" else x.contiguous()
            w = dist.all_reduce(host, group=self.group, async_op=True)
            return Pending([w], x, lambda: x.copy_(host))
        return Pending([dist.all_reduce(x, group=self.group, async_op=True)], x)

    def reduce_scatter(self, x: torch.Tensor) -> Pending:
        """x: [world * R, ...] partial sums -> this rank's [R, ...] block of the sum"""
        world = self.world
        out = torch.empty((x.shape[0] // world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        self.bytes_sent += (world - 1) * out.numel() * x.element_size()
        if self.backend() == "gloo":                 # gloo has no reduce-scatter: all-reduce and keep the own block
            buf = x.cpu() if x.is_cuda else x.clone()
            w = dist.all_reduce(buf, group=self.group, async_op=True)
            r = self.rank
            return Pending([w], out, lambda: out.copy_(buf[r * out.shape[0]:(r + 1) * out.shape[0]]))
        return Pending([dist.reduce_scatter_tensor(out, x.contiguous(), group=self.group, async_op=True)], out)


class InFlight:
    """Everything a pipelined pass has queued and that must not outlive it: side streams and collectives."""

    def __init__(self, device: Optional[torch.device] = None):
        self.device = device if (device is not None and device.type == "cuda") else None
        self.pending: List[Pending] = []
        self.streams: List["torch.cuda.Stream"] = []
        self.main = torch.cuda.current_stream(self.device) if self.device is not None else None

    def fork(self, streams: Sequence["torch.cuda.Stream"]):
        """the side streams start behind everything queued on the caller's stream"""
        if self.device is None:
            return []
        for st in streams:
            st.wait_stream(self.main)
            if st not in self.streams:
                self.streams.append(st)
        return list(streams)

    def on(self, stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def add(self, p: Pending) -> Pending:
        self.pending.append(p)
        return p

    def join(self):
        """side streams back into the caller's stream, every queued transfer waited for"""
        for st in self.streams:
            self.main.wait_stream(st)
        pend, self.pending = self.pending, []
        for p in pend:
            p.wait()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.join()
            return False
        # an exception is on its way out: nothing queued by this pass may still run when its tensors are released
        try:
            for st in self.streams:
                self.main.wait_stream(st)
            pend, self.pending = self.pending, []
            for p in pend:
                try:
                    p.wait(timeout=ERROR_WAIT_SECONDS)
                except Exception:          # (the first error is the one reported)
                    pass
        finally:
            if self.device is not None:
                try:
                    torch.cuda.synchronize(self.device)
                except Exception:
                    pass
        return False


def drain_on_error(device: Optional[torch.device] = None):
    """Context manager for drivers (bench.py, tools/): an exception leaves only after the device is idle."""
    return InFlight(device)


def install_drain_excepthook():
    """For scripts (tools/, examples/): an uncaught exception first drains the device, then is reported as usual -- the
    interpreter releases every tensor right after, and a kernel still queued on a side stream must not outlive them."""
    import sys
    previous = sys.excepthook

    def hook(exc_type, exc, tb):
        try:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
        except Exception:   # noqa: BLE001
            pass
        previous(exc_type, exc, tb)
    sys.excepthook = hook
