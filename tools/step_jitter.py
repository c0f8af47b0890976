"""Per-iteration wall time of the C3 pre_training step (hiccup hunting)."""
import os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd.synth import make_kg, make_batch
dev = torch.device("cuda:0")
cfg = SimpleNamespace(use_pretrain=0, device=dev, scale_gat_dim=None, use_residual=False, alpha=0.1, lamda=0.5,
                      aggregation_type="gcn", mess_dropout=0.0, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
                      pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300, milestone_score=0.5,
                      n_mlp_layers=2, mlp_hidden_dim=64, embed_dim=256, relation_dim=256, conv_dim=256, n_conv_layers=2,
                      use_num_lit=True, use_txt_lit=True)
n, e = 1_000_000, 10_000_000
h, t, r = make_kg(n, e)
model = L.LiteralKG(cfg, n, 16, None, torch.rand(n, 2, device=dev), torch.randn(n, 300, device=dev), scoring="transr").to(dev)
hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
model(hd, td, rd, list(range(16)), device=dev, mode="update_att")
batch = [torch.from_numpy(a).to(dev) for a in make_batch(n, 683, 3)]
ts = []
for i in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.zero_grad(set_to_none=True)
    model(*batch, device=dev, mode="pre_training").backward()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("step ms:", " ".join(f"{x:.1f}" for x in ts))
