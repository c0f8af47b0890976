#!/usr/bin/env python3
"""Does keeping the gathered table slice inside the 256 MiB Infinity Cache pay for the narrow (D/G-column) SpMM of
the feature-sharded N=8 shape?  Emulates tail-range blocking with the existing kernel: entries are regrouped
block-major (block = tail range), block b is one launch that accumulates into `out` in place (self = out).
GPU box only; not part of the tests.   python tools/mall_probe.py [--n 5000000 --e 100000000 --dim 32]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd import ops
from literalkg_amd.graph import LONG_ROW_THRESHOLD

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=5_000_000)
ap.add_argument("--e", type=int, default=100_000_000)
ap.add_argument("--dim", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--blocks", type=str, default="1,2,3,4,6,8")
ap.add_argument("--skew", default="zipf")
args = ap.parse_args()
dev = torch.device("cuda:0")
n, e, d = args.n, args.e, args.dim
rng = np.random.default_rng(5)
perm = rng.permutation(n)
h = perm[np.minimum((n * rng.random(e) ** 1.75).astype(np.int64), n - 1)] if args.skew == "zipf" else rng.integers(0, n, e, dtype=np.int64)
t = rng.integers(0, n, e, dtype=np.int64)
g = L.KGStructure.from_triples(n, h, t, None, device=dev, with_transpose=False)
del h, t
print(f"graph {n} x {g.nnz} entries, D={d}", flush=True)
x = torch.rand((n, d), device=dev)
val = torch.rand(g.nnz, device=dev)
out = torch.empty((n, d), device=dev)
by = g.nnz * (4 * d + 8) + n * 4 * d + 4 * (n + 1)


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    return float(np.median(ms))


ref = ops.spmm_raw(g.rowptr, g.col, val, x, n, long_rows=g.long_rows(False)).clone()
rows = torch.repeat_interleave(torch.arange(n, device=dev), (g.rowptr[1:] - g.rowptr[:-1]).long())
for nb in [int(b) for b in args.blocks.split(",")]:
    if nb == 1:
        ms = timeit(lambda: ops.spmm_raw(g.rowptr, g.col, val, x, n, out=out, long_rows=g.long_rows(False)))
        print(f"blocks=1 (one launch): {ms:.3f} ms -> {by/ms/1e6:.0f} GB/s algorithmic", flush=True)
        continue
    bs = (n + nb - 1) // nb
    key = (g.col.long() // bs) * n + rows
    order = torch.argsort(key, stable=True)
    ks = key[order]
    colb, valb = g.col[order].contiguous(), val[order].contiguous()
    ptr = torch.searchsorted(ks, torch.arange(nb * n + 1, device=dev)).int()
    del key, order, ks
    rps = [ptr[b * n:(b + 1) * n + 1] for b in range(nb)]
    longs = []
    for rp in rps:
        lr = torch.nonzero((rp[1:] - rp[:-1]) > LONG_ROW_THRESHOLD).flatten().int()
        longs.append(lr if lr.numel() else None)

    def run():
        for b in range(nb):
            ops.spmm_raw(rps[b], colb, valb, x, n, out=out, long_rows=longs[b], add_self=out if b else None)
    run()
    err = float((out - ref).abs().max())
    ms = timeit(run)
    print(f"blocks={nb} ({bs * d * 4 / 2**20:.0f} MiB of table per block): {ms:.3f} ms -> {by/ms/1e6:.0f} GB/s "
          f"algorithmic  (max |diff| vs one launch {err:.2e})", flush=True)
    del colb, valb, ptr, rps
