#!/usr/bin/env python3
"""Per-rank kernel times of the feature-sharded aggregation at the EXACT per-rank shape of an N-GPU run, measured on ONE
GPU (no collective runs here: the exchange is priced separately from the link budget).
    python tools/n8_rank_shape.py [--world 8] [--n 5000000] [--e 100000000] [--dim 256]
Rank 0 of `world`: the whole structure, D / world columns.  Timed (median of --iters, HIP events):
  fwd 1 launch            lkg_spmm_csr_f32 over all head rows
  fwd pipelined           the same in world x 4 head-row-range launches on two alternating streams (the launches
                          FeatureShardedAggregation.forward_to_row_block interleaves with its sends)
  bwd 1 launch            the transpose SpMM over the whole CSC
  bwd in 1 + P parts      FeatureShardedAggregation.backward_in_head_parts without the transfers: the own-block part, then P
                          launches over the sub-CSCs of row sub-range q of the other blocks, each accumulating onto the parts
                          before it (the stage lengths of the pipeline)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ge.build()
import literalkg_amd as L
from literalkg_amd.transport import install_drain_excepthook

install_drain_excepthook()      # an uncaught exception drains the device before the interpreter releases the tensors
from literalkg_amd import ops
from literalkg_amd.sharding import FeatureShardedAggregation, shard_bounds
from literalkg_amd.synth import make_kg_device

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--n", type=int, default=5_000_000)
ap.add_argument("--e", type=int, default=100_000_000)
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--iters", type=int, default=15)
ap.add_argument("--json", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
G, n, d = args.world, args.n, args.dim
dg = d // G
slab = torch.randn((n, dg), device=dev) * 0.05          # (the tables first, like a model's: see tools/graph_family_check.py)
out = torch.empty((n, dg), device=dev)
h, t, r = make_kg_device(n, args.e, "zipf", 2022, dev)
g = L.KGStructure.from_triples(n, h, t, r, device=dev)
del h, t, r
val = torch.rand(g.nnz, device=dev)
cuts = shard_bounds(g, G)
fs = FeatureShardedAggregation(g, val, 0, G, d, cuts)
by = g.nnz * (4 * dg + 8) + n * 4 * dg + 4 * (n + 1)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def fwd_pipelined(pieces=4):
    main = torch.cuda.current_stream(dev)
    for st in streams:
        st.wait_stream(main)
    step = 0
    order = [(p, k) for p in range(pieces) for k in range(1, G)] + [(None, 0)]      # the library's order: own rows last, one launch
    for p, k in order:
        lo0, hi0 = cuts[k], cuts[k + 1]
        lo, hi = (lo0, hi0) if p is None else (lo0 + (hi0 - lo0) * p // pieces, lo0 + (hi0 - lo0) * (p + 1) // pieces)
        with torch.cuda.stream(streams[step % 2]):
            ops.spmm_raw(g.rowptr[lo:hi + 1], g.col, val, slab, hi - lo, out=out[lo:hi], long_rows=g.long_rows(False, lo, hi))
        step += 1
    for st in streams:
        main.wait_stream(st)


def bwd_parts(nb):
    n_chunks, parts, vals = fs.head_parts(nb)

    def run():
        first = True
        for part, val_p in zip(parts, vals):
            if part is None or part.nnz == 0:
                continue
            ops.spmm_raw(part.rowptr, part.col, val_p, slab, n, out=out, long_rows=part.long_rows(),
                         add2=None if first else out)
            first = False
    return run, n_chunks, [p.nnz if p is not None else 0 for p in parts]


res = {"world": G, "entities": n, "stored_entries": g.nnz, "columns_per_rank": dg, "algorithmic_bytes_per_pass": by}
res["fwd_1_launch_ms"] = timeit(lambda: fs.forward(slab, out=out))
res["fwd_pipelined_launches_ms"] = timeit(fwd_pipelined)
res["bwd_1_launch_ms"] = timeit(lambda: fs.backward(slab, out=out))
want = fs.backward(slab).clone()
for nb in (2, 3, 4):      # row sub-ranges per block: 1 + nb parts
    run, sizes, nnzs = bwd_parts(nb)
    ms = timeit(run)
    run()
    err = float((out - want).abs().max() / want.abs().max())
    res[f"bwd_{nb}_head_parts_ms"] = ms
    res[f"bwd_{nb}_head_parts"] = {"row_sub_ranges_per_block": sizes, "entries_per_part (own block first)": nnzs,
                                   "max_rel_err_vs_1_launch": err}
    # per-part launch times (the pipeline's stage lengths)
    _, parts, vals = fs.head_parts(nb)
    stage = []
    for i, (part, val_p) in enumerate(zip(parts, vals)):
        stage.append(timeit(lambda: ops.spmm_raw(part.rowptr, part.col, val_p, slab, n, out=out, long_rows=part.long_rows(),
                                                 add2=None if i == 0 else out)))
    res[f"bwd_{nb}_head_parts"]["stage_ms"] = stage
for k, v in res.items():
    print(k, v)
res["frac_of_hbm_roofline"] = {k: by / (res[k] * 1e-3) / 8e12 for k in res if k.endswith("_ms")}
print(res["frac_of_hbm_roofline"])
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
