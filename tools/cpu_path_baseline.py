#!/usr/bin/env python3
"""The oracle (same ATen ops as the reference) on the host cores: update_att and one pre_training step at the C2 shape
(1M entities / 10M triples, D=128, 1 gcn layer, TransR, 2049 triples).  Context numbers for DESIGN.md; slow (~1 min)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import literalkg_oracle as O
from literalkg_amd.synth import make_batch, make_kg
torch.set_num_threads(os.cpu_count())
n, e, d = 1_000_000, 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 128
h, t, r = (torch.from_numpy(a) for a in make_kg(n, e))
cfg = O.default_cfg(embed_dim=d, relation_dim=d, conv_dim=d)
g = torch.Generator().manual_seed(0)
bound = (6.0 / (n + d)) ** 0.5
p = {"entity_embed.weight": (torch.rand(n, d, generator=g) * 2 - 1) * bound,
     "relation_embed.weight": torch.randn(16, d, generator=g) * 0.1,
     "gat_trans_M": torch.randn(16, 2 * d, d, generator=g) * 0.05,
     "aggregator_layers.0.linear.weight": torch.randn(d, d, generator=g) * 0.05,
     "aggregator_layers.0.linear.bias": torch.zeros(d),
     "aggregator_layers.0.layer_normalize.weight": torch.ones(d),
     "aggregator_layers.0.layer_normalize.bias": torch.zeros(d)}
t0 = time.perf_counter()
a = O.attention_refresh(n, p["entity_embed.weight"], p["relation_embed.weight"], h, t, r).coalesce()
t1 = time.perf_counter()
print(f"cpu update_att: {t1 - t0:.2f} s ({e / (t1 - t0) / 1e6:.2f} M edges/s), {os.cpu_count()} threads, torch {torch.__version__}")
pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
batch = [torch.from_numpy(x) for x in make_batch(n, 683, 3)]
for rep in range(2):
    t0 = time.perf_counter()
    loss = O.pre_training_loss(pp, cfg, a, *batch)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    print(f"cpu pre_training step D={d}: fwd {t1 - t0:.2f} s + bwd {t2 - t1:.2f} s = {t2 - t0:.2f} s ({e / (t2 - t0) / 1e6:.2f} M edges/s)")
