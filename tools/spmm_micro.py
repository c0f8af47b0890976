#!/usr/bin/env python3
"""Micro-benchmark of lkg_spmm_csr_f32 on the BASELINE graph shapes (GPU box only; not part of the tests).
   python tools/spmm_micro.py [--dim 256] [--skew zipf]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd.transport import install_drain_excepthook

install_drain_excepthook()      # an uncaught exception drains the device before the interpreter releases the tensors
from literalkg_amd import ops
from literalkg_amd.synth import make_kg, xavier_table

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--e", type=int, default=10_000_000)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--device-graph", action="store_true", help="draw the graph on the device (fast; no de-duplication, no degree clip)")
ap.add_argument("--only-main", action="store_true", help="time only the library's default forward / transpose launches (PMC passes)")
args = ap.parse_args()
dev = torch.device("cuda:0")
ap_skews = ("zipf", "uniform") if args.e <= 20_000_000 else ("zipf",)
for skew in ap_skews:
    # the tables first (a model's embeddings exist before its edge lists): behind the generator's temporaries the same launches
    # run 10 % slower at 5 M x 256 (placement in physical memory)
    x = xavier_table(args.n, args.dim, dev)
    out = torch.empty((args.n, args.dim), device=dev)
    if args.device_graph:
        from literalkg_amd.synth import make_kg_device
        h, t, r = make_kg_device(args.n, args.e, skew, 2022, dev)
    else:
        h, t, r = make_kg(args.n, args.e, skew)
    g = L.KGStructure.from_triples(args.n, h, t, r, device=dev)
    del h, t, r
    if os.environ.get("LKG_MICRO_EMPTY_CACHE"):     # (does the allocator's reuse of the graph generator's blocks matter for the tables?)
        torch.cuda.empty_cache()
    d = args.dim
    val = torch.rand(g.nnz, device=dev)
    val_t = ops.permute_values(val, g.t_perm)
    by = g.nnz * (4 * d + 8) + args.n * 4 * d + 4 * (args.n + 1)
    def timeit(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms = np.array([a.elapsed_time(b) for a, b in ev])
        return np.median(ms), ms.min()
    def slabs(k):
        w = d // k
        def run():
            for i in range(k):
                ops.spmm_raw(g.rowptr, g.col, val, x[:, i * w:(i + 1) * w], args.n, out=out[:, i * w:(i + 1) * w],
                             long_rows=g.long_rows(False))
        return run
    extra = [(f"fwd in {k} column slabs   ", slabs(k)) for k in (2, 4, 8, 16) if d % (4 * k) == 0 and d // k >= 16]
    if args.only_main:
        for name, fn in [("fwd", lambda: ops.spmm_raw(g.rowptr, g.col, val, x, args.n, out=out, long_rows=g.long_rows(False))),
                         ("bwd", lambda: ops.spmm_raw(g.t_rowptr, g.t_col, val_t, x, args.n, out=out, long_rows=g.long_rows(True)))]:
            med, mn = timeit(fn)
            print(f"{skew:8s} D={d} {name} median {med:.3f} ms -> {by/med/1e6:.0f} GB/s algorithmic ({by/med/1e6/8000:.3f} of 8 TB/s)")
        continue
    for name, fn in extra + [
        ("fwd wave-per-row      ", lambda: ops.spmm_raw(g.rowptr, g.col, val, x, args.n, out=out)),
        ("fwd + long-row blocks ", lambda: ops.spmm_raw(g.rowptr, g.col, val, x, args.n, out=out, long_rows=g.long_rows(False))),
        ("bwd wave-per-row      ", lambda: ops.spmm_raw(g.t_rowptr, g.t_col, val_t, x, args.n, out=out)),
        ("bwd + long-row blocks ", lambda: ops.spmm_raw(g.t_rowptr, g.t_col, val_t, x, args.n, out=out, long_rows=g.long_rows(True))),
    ]:
        med, mn = timeit(fn)
        print(f"{skew:8s} D={d} {name} median {med:.3f} ms  min {mn:.3f} ms  -> {by/med/1e6:.0f} GB/s algorithmic "
              f"({by/med/1e6/8000:.3f} of 8 TB/s)  long rows: {0 if g.long_rows(False) is None else g.long_rows(False).numel()}")
    # (the rows-per-wave / gathers-in-flight experiments are COMPILE-TIME variants since round 4: rebuild with
    #  LKG_EXTRA_HIPCC_FLAGS="-DLKG_SPMM_GROUPED_CHUNKS=32" or "-DLKG_SPMM_U=8" and run this tool again)
    # plain copy reference for this box: read+write 2 x table
    y = torch.empty_like(x)
    med, _ = timeit(lambda: y.copy_(x))
    print(f"         torch copy of the {x.numel()*4/1e9:.2f} GB table: {med:.3f} ms -> {2*x.numel()*4/med/1e6:.0f} GB/s")
