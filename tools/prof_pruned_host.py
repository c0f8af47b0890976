#!/usr/bin/env python3
"""cProfile of the host side of the batch-pruned C2 step.  GPU box only; not part of the tests."""
import os, sys, cProfile, pstats, io
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
import literalkg_amd as L
from literalkg_amd.synth import make_kg, make_batch
dev = torch.device("cuda:0")
cfg = SimpleNamespace(use_pretrain=0, device=dev, scale_gat_dim=None, use_residual=False, alpha=0.1, lamda=0.5,
            aggregation_type="gcn", mess_dropout=0.0, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
            pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300, milestone_score=0.5,
            n_mlp_layers=2, mlp_hidden_dim=64, embed_dim=128, relation_dim=128, conv_dim=128, n_conv_layers=1,
            use_num_lit=False, use_txt_lit=False)
n, e = 1_000_000, 10_000_000
h, t, r = make_kg(n, e)
model = L.LiteralKG(cfg, n, 16, None, None, None).to(dev)
model.prune_to_batch = True
hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
model(hd, td, rd, list(range(16)), device=dev, mode="update_att")
bh, br, bp, bn = (torch.from_numpy(a).to(dev) for a in make_batch(n, 683, 3))
def step():
    model.zero_grad(set_to_none=True)
    loss = model(bh, br, bp, bn, device=dev, mode="pre_training")
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
