#!/usr/bin/env python3
"""Whole-path timing on the BASELINE shapes (GPU box): update_att, pre_training forward, backward.
   python tools/step_profile.py --config c2|c3 [--agg gcn]"""
import argparse, os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd.transport import install_drain_excepthook

install_drain_excepthook()      # an uncaught exception drains the device before the interpreter releases the tensors
from literalkg_amd.synth import make_kg, make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--agg", default="gcn")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--scoring", default="transr")
ap.add_argument("--prune", action="store_true")
ap.add_argument("--fused-adam", action="store_true")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--e", type=int, default=10_000_000)
ap.add_argument("--groups", type=int, default=None, help="sampled heads per batch (default 683; the reference's pre_training_batch_size is 2048)")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfgs = {
    # main.py's defaults (argument.py:34-118): bi-interaction WITH the GCNII-style residual, eight layers of 32 over 300-wide
    # embeddings, GateMul, linear_gat 556 -> 256
    "main_default": dict(embed_dim=300, relation_dim=300, conv_dim=32, n_conv_layers=8, use_num_lit=True, use_txt_lit=True,
                         scale_gat_dim=256, mess_dropout=0.1, aggregation_type="bi-interaction", use_residual=True),
    # the reference's DEFAULT architecture (argument_pretraining.py:34-62): 300-wide embeddings, eight gcn layers of 32, GateMul,
    # linear_gat 556 -> 300, TransR 300 x 300 per relation
    "default": dict(embed_dim=300, relation_dim=300, conv_dim=32, n_conv_layers=8, use_num_lit=True, use_txt_lit=True,
                    scale_gat_dim=300, mess_dropout=0.1),
    # BASELINE config[0]'s shape (data/Small; run with --n 765957 --e 252000): the launch-bound end of the range
    "c1": dict(embed_dim=64, relation_dim=64, conv_dim=64, n_conv_layers=1, use_num_lit=False, use_txt_lit=False),
    "c2": dict(embed_dim=128, relation_dim=128, conv_dim=128, n_conv_layers=1, use_num_lit=False, use_txt_lit=False),
    "d256": dict(embed_dim=256, relation_dim=256, conv_dim=256, n_conv_layers=1, use_num_lit=False, use_txt_lit=False),
    "c3": dict(embed_dim=256, relation_dim=256, conv_dim=256, n_conv_layers=2, use_num_lit=True, use_txt_lit=True),
    # BASELINE config[4] shape on one GPU (scale the graph with --n/--e): D=512, 3 layers, TransR, K=256 negatives
    "c5": dict(embed_dim=512, relation_dim=512, conv_dim=512, n_conv_layers=3, use_num_lit=False, use_txt_lit=False,
               pre_training_neg_rate=256),
}
base = dict(use_pretrain=0, device=dev, scale_gat_dim=None, use_residual=False, alpha=0.1, lamda=0.5,
            aggregation_type=args.agg, mess_dropout=0.0, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
            pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300, milestone_score=0.5,
            n_mlp_layers=2, mlp_hidden_dim=64)
base.update(cfgs[args.config])
cfg = SimpleNamespace(**base)
n, e = args.n, args.e
h, t, r = make_kg(n, e)
num = torch.rand(n, 2, device=dev) if cfg.use_num_lit else None
txt = torch.randn(n, 300, device=dev) if cfg.use_txt_lit else None
model = L.LiteralKG(cfg, n, 16, None, num, txt, scoring=args.scoring).to(dev)
model.eval()
model.prune_to_batch = args.prune
hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
def sync_time(fn, iters=args.iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
t0 = time.perf_counter()
model(hd, td, rd, list(range(16)), device=dev, mode="update_att"); torch.cuda.synchronize()
print(f"first update_att incl. host CSR build: {(time.perf_counter()-t0)*1e3:.0f} ms")
ua = sync_time(lambda: model(hd, td, rd, list(range(16)), device=dev, mode="update_att"))
k_neg = cfg.pre_training_neg_rate
bh, br, bp, bn = (torch.from_numpy(a).to(dev) for a in make_batch(n, args.groups or (683 if k_neg == 3 else 128), k_neg))
if args.fused_adam:
    from literalkg_amd.optim import Adam
    opt = Adam(model.parameters(), lr=1e-4)
else:
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
def fwd():
    return model(bh, br, bp, bn, device=dev, mode="pre_training")
def step():
    opt.zero_grad(set_to_none=True)
    loss = fwd(); loss.backward()
def step_opt():
    step(); opt.step()
with torch.no_grad():
    f_ng = sync_time(fwd)
f = sync_time(fwd)
s = sync_time(step)
so = sync_time(step_opt)
L_ = cfg.n_conv_layers
print(f"{args.config} {args.agg} {args.scoring} prune={args.prune} fused_adam={args.fused_adam}: update_att {ua:.2f} ms ({e/ua/1e6:.2f} G edges/s) | fwd(no grad) {f_ng:.2f} | fwd {f:.2f} | fwd+bwd {s:.2f} ms "
      f"({e*L_/s/1e6:.2f} G edges/s) | +Adam {so:.2f} ms | mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
if os.environ.get("LKG_STEP_GPU_TIME"):      # device-side time of one step (event pair around it, queue kept full): host-bound if << wall
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / len(ev) * 1e3
    import numpy as np
    print(f"fwd+bwd: wall {wall:.2f} ms per step, device span {np.median([a.elapsed_time(b) for a, b in ev]):.2f} ms")
