#!/usr/bin/env python3
"""Cost of cutting one SpMM into row-range launches (one stream vs alternating streams), N = 8 per-GPU shape.  GPU box only; not part of the tests."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
import literalkg_amd as L
from literalkg_amd import ops
dev = torch.device("cuda:0")
n, e, d = 5_000_000, 100_000_000, 32
rng = np.random.default_rng(5)
perm = rng.permutation(n)
h = perm[np.minimum((n * rng.random(e) ** 1.75).astype(np.int64), n - 1)]
t = rng.integers(0, n, e, dtype=np.int64)
g = L.KGStructure.from_triples(n, h, t, None, device=dev, with_transpose=False)
x = torch.rand((n, d), device=dev); val = torch.rand(g.nnz, device=dev); out = torch.empty((n, d), device=dev)
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
def run(parts):
    cuts = [n * i // parts for i in range(parts + 1)]
    lrs = [g.long_rows(False, cuts[i], cuts[i + 1]) for i in range(parts)]
    def fn():
        for i in range(parts):
            lo, hi = cuts[i], cuts[i + 1]
            ops.spmm_raw(g.rowptr[lo:hi + 1], g.col, val, x, hi - lo, out=out[lo:hi], long_rows=lrs[i])
    return fn
def run_streams(parts, n_streams):
    cuts = [n * i // parts for i in range(parts + 1)]
    lrs = [g.long_rows(False, cuts[i], cuts[i + 1]) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    def fn():
        main = torch.cuda.current_stream()
        for s_ in streams:
            s_.wait_stream(main)
        for i in range(parts):
            lo, hi = cuts[i], cuts[i + 1]
            with torch.cuda.stream(streams[i % n_streams]):
                ops.spmm_raw(g.rowptr[lo:hi + 1], g.col, val, x, hi - lo, out=out[lo:hi], long_rows=lrs[i])
        for s_ in streams:
            main.wait_stream(s_)
    return fn
for parts in (1, 8, 16, 32):
    print(f"{parts:3d} row-range launches: {timeit(run(parts)):.3f} ms   on 2 alternating streams: "
          f"{timeit(run_streams(parts, 2)):.3f} ms   on 4: {timeit(run_streams(parts, 4)):.3f} ms", flush=True)
