#!/usr/bin/env python3
"""lkg_spmm_csr_fused_f32 at the BASELINE shape with its epilogue extras, back to back and behind another kernel's traffic
(GPU box only; tuning aid).   python tools/spmm_extras_micro.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd import ops
from literalkg_amd.synth import make_kg, xavier_table

dev = torch.device("cuda:0")
n, e, d = 1_000_000, 10_000_000, 256
h, t, r = make_kg(n, e, "zipf")
g = L.KGStructure.from_triples(n, h, t, r, device=dev)
x = xavier_table(n, d, dev)
wide = torch.empty((n, 3 * d), device=dev)
wide[:, :d] = x
xs = wide[:, :d]
val = torch.rand(g.nnz, device=dev)
out = torch.empty((n, d), device=dev)
rm = torch.empty(n, device=dev)
junk_a = torch.empty(256 << 20, device=dev)
junk_b = torch.empty(256 << 20, device=dev)


def timeit(fn, between=None, iters=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        if between is not None:
            between()
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


lr = g.long_rows(False)
cases = [
    ("plain                 ", lambda: ops.spmm_raw(g.rowptr, g.col, val, x, n, out=out, long_rows=lr)),
    ("+ self                ", lambda: ops.spmm_raw(g.rowptr, g.col, val, x, n, out=out, long_rows=lr, add_self=x)),
    ("+ rowmax              ", lambda: ops.spmm_raw(g.rowptr, g.col, val, x, n, out=out, long_rows=lr, rowmax=rm)),
    ("+ self + rowmax       ", lambda: ops.spmm_raw(g.rowptr, g.col, val, x, n, out=out, long_rows=lr, add_self=x, rowmax=rm)),
    ("+ self + rowmax ld=768", lambda: ops.spmm_raw(g.rowptr, g.col, val, xs, n, out=out, long_rows=lr, add_self=xs, rowmax=rm)),
]
for name, fn in cases:
    a = timeit(fn)
    b = timeit(fn, between=lambda: junk_b.copy_(junk_a))
    print(f"{name} back to back {a:.3f} ms | behind a 2 GB copy {b:.3f} ms")
