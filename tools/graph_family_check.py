#!/usr/bin/env python3
"""make_kg (host, numpy) against make_kg_device (torch on the GPU): the same graph family?  Degree statistics and the SpMM pair
on both, at one size (GPU box only).   python tools/graph_family_check.py [--n 1000000] [--e 10000000]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd import ops
from literalkg_amd.synth import make_kg, make_kg_device
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--e", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=256)
a = ap.parse_args()
dev = torch.device("cuda:0")
x = torch.randn(a.n, a.dim, device=dev) * 0.05
out = torch.empty_like(x)
def tm(fn, reps=10):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s_, e_ in ev:
        s_.record(); fn(); e_.record()
    torch.cuda.synchronize()
    return float(np.median([s_.elapsed_time(e_) for s_, e_ in ev]))
for name, (h, t, r) in (("make_kg (host)", make_kg(a.n, a.e)), ("make_kg_device", make_kg_device(a.n, a.e, "zipf", 2022, dev))):
    g = L.KGStructure.from_triples(a.n, h, t, r, device=dev)
    od = (g.rowptr[1:] - g.rowptr[:-1]).float()
    idg = (g.t_rowptr[1:] - g.t_rowptr[:-1]).float()
    q = torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)
    val = torch.rand(g.nnz, device=dev)
    val_t = ops.permute_values(val, g.t_perm)
    f = tm(lambda: ops.spmm_raw(g.rowptr, g.col, val, x, a.n, out=out, long_rows=g.long_rows(False)))
    b = tm(lambda: ops.spmm_raw(g.t_rowptr, g.t_col, val_t, x, a.n, out=out, long_rows=g.long_rows(True)))
    print(f"{name:16s} nnz {g.nnz}  out-degree: empty rows {int((od == 0).sum())} q50/90/99/99.9 {[int(v) for v in torch.quantile(od, q)]} max {int(od.max())} "
          f"rows > 256: {int((od > 256).sum())} entries in them {int(od[od > 256].sum())} | in-degree max {int(idg.max())} | fwd {f:.3f} ms  bwd {b:.3f} ms")
