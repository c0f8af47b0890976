#!/usr/bin/env python3
"""Memory over three epochs of a training loop (update_att + 15 steps with the fused Adam) at the reference's default architecture
on a 200 k / 2 M graph: allocated and peak MiB per epoch must stay flat (GPU box only; debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd.optim import Adam
from literalkg_amd.synth import make_kg, make_batch
dev = torch.device("cuda:0")
n, e = 200_000, 2_000_000
h, t, r = make_kg(n, e)
cfg = SimpleNamespace(use_pretrain=0, device=dev, use_residual=bool(int(os.environ.get("RES", "0"))), alpha=0.1, lamda=0.5, aggregation_type=os.environ.get("AGG", "gcn"), mess_dropout=0.1,
            kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5, pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2,
            txt_lit_dim=300, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64, embed_dim=300, relation_dim=300, scale_gat_dim=300,
            n_conv_layers=8, conv_dim=32, use_num_lit=True, use_txt_lit=True)
model = L.LiteralKG(cfg, n, 16, None, torch.rand(n, 2, device=dev), torch.randn(n, 300, device=dev)).to(dev)
hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
opt = Adam(model.parameters(), lr=1e-3)
sizes = []
for epoch in range(3):
    model(hd, td, rd, list(range(16)), device=dev, mode="update_att")
    for it in range(15):
        batch = [torch.from_numpy(a).to(dev) for a in make_batch(n, 683, 3, seed=epoch * 100 + it)]
        opt.zero_grad(set_to_none=True)
        loss = model(*batch, device=dev, mode="pre_training")
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    sizes.append((torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20, float(loss)))
print(sizes)
