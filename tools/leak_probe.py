#!/usr/bin/env python3
"""Does a forward WITHOUT a backward leave its autograd graph alive?  Growth of allocated memory per forward-only call
for a few architectures (GPU box only; debugging aid)."""
import gc, os, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd.synth import make_kg, make_batch
dev = torch.device("cuda:0")
n, e = 200_000, 2_000_000
h, t, r = make_kg(n, e)
base = dict(use_pretrain=0, device=dev, use_residual=False, alpha=0.1, lamda=0.5, aggregation_type="gcn", mess_dropout=0.1,
            kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5, pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2,
            txt_lit_dim=300, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64)
variants = {
    "default": dict(embed_dim=300, relation_dim=300, scale_gat_dim=300, n_conv_layers=8, conv_dim=32, use_num_lit=True, use_txt_lit=True),
    "narrow, no gate, no scale": dict(embed_dim=128, relation_dim=128, scale_gat_dim=None, n_conv_layers=2, conv_dim=32, use_num_lit=False, use_txt_lit=False),
    "equal dims, gate": dict(embed_dim=128, relation_dim=128, scale_gat_dim=None, n_conv_layers=2, conv_dim=128, use_num_lit=True, use_txt_lit=True),
    "narrow, gate": dict(embed_dim=128, relation_dim=128, scale_gat_dim=None, n_conv_layers=2, conv_dim=32, use_num_lit=True, use_txt_lit=True),
    "equal dims, scale": dict(embed_dim=128, relation_dim=128, scale_gat_dim=64, n_conv_layers=2, conv_dim=128, use_num_lit=False, use_txt_lit=False),
}
hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
batch = [torch.from_numpy(a).to(dev) for a in make_batch(n, 683, 3)]
for name, over in variants.items():
    cfg = SimpleNamespace(**{**base, **over})
    num = torch.rand(n, 2, device=dev) if cfg.use_num_lit else None
    txt = torch.randn(n, 300, device=dev) if cfg.use_txt_lit else None
    model = L.LiteralKG(cfg, n, 16, None, num, txt).to(dev)
    model(hd, td, rd, list(range(16)), device=dev, mode="update_att")
    model.train()
    sizes = []
    for it in range(4):
        loss = model(*batch, device=dev, mode="pre_training")
        torch.cuda.synchronize()
        sizes.append(torch.cuda.memory_allocated() / 2**20)
    del loss
    gc.collect()
    print(f"{name:28s} allocated after forward-only calls (MiB): {[round(x) for x in sizes]}  after gc {torch.cuda.memory_allocated() / 2**20:.0f}")
    del model
    gc.collect(); torch.cuda.empty_cache()
