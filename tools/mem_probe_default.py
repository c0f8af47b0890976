#!/usr/bin/env python3
"""Peak device memory of one pre_training step at the reference's default architecture, phase by phase (GPU box only)."""
import os, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd.synth import make_kg, make_batch
dev = torch.device("cuda:0")
n, e = 1_000_000, 10_000_000
cfg = SimpleNamespace(use_pretrain=0, device=dev, embed_dim=300, relation_dim=300, scale_gat_dim=300, use_residual=False, alpha=0.1,
                      lamda=0.5, aggregation_type="gcn", n_conv_layers=8, conv_dim=32, mess_dropout=0.1, kg_l2loss_lambda=1e-5,
                      fine_tuning_l2loss_lambda=1e-5, pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2,
                      txt_lit_dim=300, use_num_lit=True, use_txt_lit=True, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64)
h, t, r = make_kg(n, e)
model = L.LiteralKG(cfg, n, 16, None, torch.rand(n, 2, device=dev), torch.randn(n, 300, device=dev)).to(dev)
hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
model(hd, td, rd, list(range(16)), device=dev, mode="update_att")
batch = [torch.from_numpy(a).to(dev) for a in make_batch(n, 683, 3)]
gib = lambda: (torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30)
print("after setup: alloc %.1f GiB peak %.1f" % gib())
for it in range(3):
    torch.cuda.reset_peak_memory_stats()
    model.zero_grad(set_to_none=True)
    loss = model(*batch, device=dev, mode="pre_training")
    torch.cuda.synchronize()
    print(f"step {it} after forward: alloc %.1f GiB peak %.1f" % gib())
    loss.backward()
    torch.cuda.synchronize()
    print(f"step {it} after backward: alloc %.1f GiB peak %.1f" % gib())
    del loss
from literalkg_amd import ops
for k, ent in ops._RowScratch._tables.items():
    print("scratch", k[1:], "%.2f GiB" % (ent.buf.numel() * 4 / 2**30))
print("workspaces", {k[1]: "%.2f GiB" % (v.numel() / 2**30) for k, v in ops._workspaces.items()})
