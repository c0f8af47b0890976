#!/usr/bin/env python3
"""Timing of the fused attention refresh (lkg_edge_softmax_f32) on the BASELINE graph (GPU box only)."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd import ops
from literalkg_amd.synth import make_kg, xavier_table
ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--rel", type=int, default=16, help="relations (16: the synthetic graphs' own ids)")
args = ap.parse_args()
dev = torch.device("cuda:0")
n, e, d = 1_000_000, 10_000_000, args.dim
for dup in (2e-4,):
    for skew in ("zipf", "uniform"):
        h, t, r = make_kg(n, e, skew, dup_frac=dup)
        if args.rel != 16:
            r = np.random.default_rng(5).integers(0, args.rel, len(r))
        g = L.KGStructure.from_triples(n, h, t, r, device=dev)
        ent = xavier_table(n, d, dev); rel = xavier_table(args.rel, d, dev, seed=7)
        val = torch.empty(g.nnz, device=dev)
        def fn(): ops.edge_softmax(g, ent, rel, out=val)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        by = g.nnz * (4 * d + 12) + n * 4 * d + args.rel * 4 * d + 4 * (n + 1)
        print(f"D={d} {skew:8s} dups={g.has_dups!s:5s} {ms:.3f} ms  {by/ms/1e6:.0f} GB/s algorithmic ({by/ms/1e6/8000:.3f} of 8 TB/s)")
