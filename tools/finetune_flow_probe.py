#!/usr/bin/env python3
"""The fine-tuning flow at argument_finetuning.py's defaults on the 1 M / 10 M graph: one fine_tuning step and an evaluate()-style loop of
predict calls (GPU box only; timing aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd.optim import Adam
from literalkg_amd.synth import make_kg
dev = torch.device("cuda:0")
n, e = 1_000_000, 10_000_000
h, t, r = make_kg(n, e)
cfg = SimpleNamespace(use_pretrain=0, device=dev, use_residual=False, alpha=0.1, lamda=0.5, aggregation_type="gcn", mess_dropout=0.1,
            kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5, pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2,
            txt_lit_dim=300, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64, embed_dim=300, relation_dim=300, scale_gat_dim=300,
            n_conv_layers=1, conv_dim=32, use_num_lit=True, use_txt_lit=True)
model = L.LiteralKG(cfg, n, 16, None, torch.rand(n, 2, device=dev), torch.randn(n, 300, device=dev)).to(dev)
hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
model(hd, td, rd, list(range(16)), device=dev, mode="update_att")
opt = Adam(model.parameters(), lr=1e-3)
rng = np.random.default_rng(0)
B = 2048
heads, pos, neg = (torch.from_numpy(rng.integers(0, n, B)).to(dev) for _ in range(3))
model.train()
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(heads, pos, neg, device=dev, mode="fine_tuning")
    loss.backward(); opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): l = step()
torch.cuda.synchronize()
print(f"fine_tuning step (argument_finetuning.py defaults: gcn x1 300 -> 32, GateMul, linear_gat 332 -> 300, batch {B}) incl. fused Adam: {(time.perf_counter()-t0)/10*1e3:.2f} ms, loss {float(l):.4f}")
model.eval()
with torch.no_grad():
    hb = [torch.arange(i, i + 2048, device=dev) for i in range(0, 20480, 2048)]
    tails = torch.arange(0, 5000, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in hb: s = model(b, tails, device=dev, mode="predict")
    torch.cuda.synchronize()
    print(f"evaluate-style loop: 10 predict calls of 2048 heads x 5000 tails: {(time.perf_counter()-t0)*1e3:.1f} ms total")
