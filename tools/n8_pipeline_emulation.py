#!/usr/bin/env python3
"""The N = 8 step of the feature-sharded aggregation on ONE GPU with the transfers EMULATED: the real per-rank launches in the
library's order (FeatureShardedAggregation.exchange_aggregate: forward = the pass by owner ranges, other ranks' rows first;
backward = the own-block part, then one part per row sub-range of the incoming blocks) and, in place of RCCL over xGMI, one
timeline that holds every message for bytes / link-rate (torch.cuda._sleep) while writing the same bytes into HBM.
What this checks on hardware: that the stream / event choreography overlaps as DESIGN.md section 6.1 assumes and what the
launches cost when they run next to incoming traffic.  What it cannot check: RCCL itself (ordering of batched point-to-point
groups, how close a link gets to its rate) -- no multi-GPU machine was available.
    python tools/n8_pipeline_emulation.py [--world 8] [--n 5000000] [--e 100000000] [--dim 256] [--json out.json]
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
ap.add_argument("--chunks", type=int, default=3, help="row sub-ranges per incoming block (the library's default: 3)")
ap.add_argument("--pieces", type=int, default=4, help="pieces per outgoing owner range (the library's default: 4)")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--n1-ms", type=float, default=3.05, help="the N = 1 step (10 M edges) the scaling is quoted against")
ap.add_argument("--json", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
G, n, d = args.world, args.n, args.dim
dg = d // G
slab = torch.randn((n, dg), device=dev) * 0.05
side = torch.empty((n, dg), device=dev)
h, t, r = make_kg_device(n, args.e, "zipf", 2022, dev)
g = L.KGStructure.from_triples(n, h, t, r, device=dev)
del h, t, r
val = torch.rand(g.nnz, device=dev)
cuts = shard_bounds(g, G)
fs = FeatureShardedAggregation(g, val, 0, G, d, cuts)
rows0 = fs.my_rows
rows_max = max(fs.rows)
block = torch.randn((G, rows_max, dg), device=dev) * 0.05       # what this rank would send / receive as panels (blocks are cut
landing = torch.empty((G, rows_max, dg), device=dev)            # by stored entries: their row counts differ a little)
landing_flat = landing.view(-1)
block_bytes = rows0 * dg * 4

# cycles per millisecond of torch.cuda._sleep
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda._sleep(1_000_000)
torch.cuda.synchronize()
ev[0].record()
torch.cuda._sleep(50_000_000)
ev[1].record()
torch.cuda.synchronize()
cyc_per_ms = 50_000_000 / ev[0].elapsed_time(ev[1])

links = [torch.cuda.Stream(device=dev) for _ in range(G)]       # links[k]: the link to / from the peer at offset k
compute = [torch.cuda.Stream(device=dev) for _ in range(2)]
main = torch.cuda.current_stream(dev)


def hold(ms):
    torch.cuda._sleep(int(ms * cyc_per_ms))


SUB = 6


def arrive(ms, nbytes):
    """`nbytes` arriving over `ms`: the time in SUB holds, after each of them that share of the bytes WRITTEN into HBM (what a
    receive does to the memory system: writes at the links' rate, spread over the transfer -- a device copy at the end would read
    AND write them in a burst at HBM speed and overstate the interference)"""
    words = int(nbytes / 4 / SUB)
    for _ in range(SUB):
        hold(ms / SUB)
        if words:
            landing_flat[:words].fill_(0.5)


def forward(link_ms_per_block):
    """the pass by owner ranges (the library's order and streams).  The links are symmetric and independent, so ONE timeline
    stands for them -- the link of the peer served LAST in every round (offset G - 1), whose pieces become ready latest: its
    piece p is held for a quarter of a block's time once the launch that produced it is done, while the bytes of the whole round
    (the pieces arriving over all G - 1 links) are written into HBM.  (An event + a hold + a copy per piece and link --
    100 host calls per pass -- made the emulation host-bound: 3.0 ms for the 2.3 ms of launches.)"""
    for st in compute + [links[G - 1]]:
        st.wait_stream(main)
    order = [(p, k) for p in range(args.pieces) for k in range(1, G)] + [(None, 0)]
    step = 0
    for p, k in order:
        lo0, hi0 = cuts[k], cuts[k + 1]
        lo, hi = (lo0, hi0) if p is None else (lo0 + (hi0 - lo0) * p // args.pieces, lo0 + (hi0 - lo0) * (p + 1) // args.pieces)
        cs = compute[step % 2]
        step += 1
        with torch.cuda.stream(cs):
            ops.spmm_raw(g.rowptr[lo:hi + 1], g.col, val, slab, hi - lo, out=side[lo:hi], long_rows=g.long_rows(False, lo, hi))
        if k != G - 1:
            continue
        links[G - 1].wait_stream(cs)                             # (the piece is done: the stream's work so far)
        with torch.cuda.stream(links[G - 1]):                    # round p's pieces from all G - 1 peers
            arrive(link_ms_per_block / args.pieces, (G - 1) * block_bytes / args.pieces)
    for st in compute + [links[G - 1]]:
        main.wait_stream(st)


def backward(link_ms_per_block):
    """the own-block part at once; sub-range q of every incoming block after (q + 1) / chunks of a block's time"""
    n_chunks, parts, vals = fs.head_parts(args.chunks)
    bounds = fs.chunk_bounds(n_chunks)
    links[1].wait_stream(main)
    arrived = []
    with torch.cuda.stream(links[1]):                            # all links run in parallel: one timeline stands for them
        for q in range(n_chunks):
            arrive(link_ms_per_block / n_chunks, (G - 1) * block_bytes / n_chunks)
            e = torch.cuda.Event()
            e.record()
            arrived.append(e)
    slab[cuts[0]:cuts[1]].copy_(block[0, :rows0])
    first = True
    for b, (part, val_p) in enumerate(zip(parts, vals)):
        if b > 0:
            main.wait_event(arrived[b - 1])
        if part.nnz:
            ops.spmm_raw(part.rowptr, part.col, val_p, slab, n, out=side, long_rows=part.long_rows(), add2=None if first else side)
            first = False


def timed(fn, x):
    for _ in range(2):
        fn(x)
    torch.cuda.synchronize()
    ms = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(x)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.median(ms))


res = {"what": "EMULATED transfers (per-link holds + the same bytes through HBM), real per-rank launches; rank 0 of %d" % G,
       "entities": n, "stored_entries": g.nnz, "columns_per_rank": dg, "block_MB_per_peer_and_pass": block_bytes / 1e6,
       "incoming_row_sub_ranges": args.chunks, "outgoing_pieces": args.pieces, "n1_step_ms": args.n1_ms, "by_link_rate": {}}
res["no_exchange"] = {"fwd_ms": timed(forward, 0.0), "bwd_ms": timed(backward, 0.0)}
for gbs in (76.8, 64.0, 48.0, 40.0, 32.0):
    x = block_bytes / (gbs * 1e6)                                # ms for one block over one link at this rate
    f, b = timed(forward, x), timed(backward, x)
    res["by_link_rate"][f"{gbs:g} GB/s per link and direction"] = {
        "block_ms_on_a_link": x, "fwd_ms": f, "bwd_ms": b, "step_ms": f + b,
        "unpipelined_step_ms": res["no_exchange"]["fwd_ms"] + res["no_exchange"]["bwd_ms"] + 2 * x,
        "scaling_vs_n1": (2 * g.nnz / (f + b)) / (2 * 9_997_896 / args.n1_ms)}
for k, v in res.items():
    print(k, v)
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
