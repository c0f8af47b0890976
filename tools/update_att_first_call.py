#!/usr/bin/env python3
"""Where the FIRST update_att of an edge list spends its time (GPU box): structure build pieces vs the refresh."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd import ops
from literalkg_amd.synth import make_kg, xavier_table
dev = torch.device("cuda:0")
n = 1_000_000
h, t, r = (torch.from_numpy(x).to(dev) for x in make_kg(n, 10_000_000))
ent, rel = xavier_table(n, 256, dev), xavier_table(16, 256, dev, seed=7)
def tm(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) * 1e3:8.2f} ms"); return out
for rep in range(2):
    print("--- pass", rep)
    keep = tm("isin(r, relations).all()", lambda: bool(torch.isin(r, torch.arange(16, device=dev)).all()))
    g = tm("KGStructure.from_triples (device)", lambda: L.KGStructure.from_triples(n, h, t, r, device=dev))
    tm("long_rows x2", lambda: (g.long_rows(False), g.long_rows(True)))
    tm("clone lists", lambda: [x.clone() for x in (h, t, r)])
    val = tm("edge_softmax", lambda: ops.edge_softmax(g, ent, rel)[0])
    tm("coo_indices", lambda: g.coo_indices())
    tm("permute_values", lambda: ops.permute_values(val, g.t_perm))
