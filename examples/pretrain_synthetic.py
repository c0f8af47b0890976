#!/usr/bin/env python3
"""A driver shaped like the reference's pre_training_train (main_pretraining.py:67-167) that calls the
MI355X module through the SAME surface: ctor, model.to(device), Adam over model.parameters(),
model(h, r, pos_t, neg_t, device=, mode='pre_training'), NaN check, loss.backward(), optimizer.step(),
model(h_list, t_list, r_list, relations, device=, mode='update_att') once per epoch, state_dict checkpoint.
Only the data source differs (synthetic KG instead of the reference's DataLoader, which is out of scope).

    python examples/pretrain_synthetic.py --entities 200000 --edges 2000000 --dim 128 --epochs 2 --iters 20
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from literalkg_amd import LiteralKG                      # noqa: E402  (the one-line change vs `from model import LiteralKG`)
from literalkg_amd.synth import make_batch, make_kg      # noqa: E402


def initial_a_in(n, h, t, r):
    """Random-walk Laplacian summed over relations, as the reference's loader builds it
    (dataloader.py:449-495): value 1/outdeg_r(h) per (h, r, t), summed over relations per (h, t)."""
    key = h * (int(r.max()) + 1) + r
    _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    w = 1.0 / cnt[inv]
    idx = torch.from_numpy(np.stack([h, t]))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(w.astype(np.float32)), (n, n)).coalesce()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entities", type=int, default=200_000)
    ap.add_argument("--edges", type=int, default=2_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch-groups", type=int, default=683)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args()
    device = torch.device("cuda:0")
    args = SimpleNamespace(use_pretrain=0, device=device, embed_dim=a.dim, relation_dim=a.dim, scale_gat_dim=None,
                           use_residual=False, alpha=0.1, lamda=0.5, aggregation_type="gcn", n_conv_layers=a.layers,
                           conv_dim=a.dim, mess_dropout=0.1, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
                           pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300,
                           use_num_lit=True, use_txt_lit=False, milestone_score=0.5, n_mlp_layers=2,
                           mlp_hidden_dim=64)
    torch.manual_seed(2022)
    h, t, r = make_kg(a.entities, a.edges)
    num = torch.rand(a.entities, 2)
    model = LiteralKG(args, a.entities, 16, initial_a_in(a.entities, h, t, r), num, None)
    optimizer = torch.optim.Adam(model.parameters(), lr=a.lr)
    model.to(device)
    h_list, t_list, r_list = (torch.from_numpy(x).to(device) for x in (h, t, r))
    relations = list(range(16))
    for epoch in range(1, a.epochs + 1):
        model.train()
        t0, total = time.time(), 0.0
        for it in range(a.iters):
            bh, br, bp, bn = (torch.from_numpy(x).to(device)
                              for x in make_batch(a.entities, a.batch_groups, 3, seed=epoch * 10_000 + it))
            optimizer.zero_grad()
            loss = model(bh, br, bp, bn, device=device, mode="pre_training")
            if np.isnan(loss.cpu().detach().numpy()):
                sys.exit("ERROR (Pre-training): loss is nan")
            loss.backward()
            optimizer.step()
            total += loss.item()
        t1 = time.time()
        model(h_list, t_list, r_list, relations, device=device, mode="update_att")
        torch.cuda.synchronize()
        print(f"epoch {epoch}: mean loss {total / a.iters:.4f} | {a.iters} iters {t1 - t0:.2f} s "
              f"({(t1 - t0) / a.iters * 1e3:.1f} ms/iter) | update_att {time.time() - t1:.3f} s")
    path = "/tmp/pre-training_model_epoch%d.pth" % a.epochs
    torch.save({"model_state_dict": model.state_dict(), "epoch": a.epochs}, path)
    back = LiteralKG(args, a.entities, 16)
    back.load_state_dict(torch.load(path, map_location="cpu")["model_state_dict"])
    print("checkpoint round trip ok:", path, "A_in nnz", back.A_in._nnz())


if __name__ == "__main__":
    main()
