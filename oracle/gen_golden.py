#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE on CPU in the build container.

    python oracle/gen_golden.py            # needs /root/reference (read-only mount)

The reference cannot travel to the GPU box, so what is committed is DATA only:
seeded inputs (graph, weights keyed by the reference's state_dict names,
literals, batches) and the outputs the reference's own model.py / model_bce.py /
gate.py produced for them with torch CPU in this container.  No reference source
text is written anywhere.  Re-running this script regenerates the same files
(torch CPU RNG, fixed seeds).

Fixture families:
  attention_*  a4 per-edge logits + a5 refreshed A_in (indices int64 bit-exact, values)
  encoder_*    a1/a2/a3/a6/a7 gat_embeddings for aggregator x residual x gate x scale combos,
               a8 TransR loss, pos/neg scores, gradients of every trainable parameter
  transe_*     a9 TransE loss (model_bce.py) + gradients
  heads_*      f1 fine-tuning loss, calc_score matrix, predict_links
  statedict    key -> shape manifest of the reference's state_dict
"""
import json
import os
import sys
import warnings
from types import SimpleNamespace

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def ref_modules():
    sys.path.insert(0, REF)
    import model as ref_model          # noqa: E402
    import model_bce as ref_model_bce  # noqa: E402
    return ref_model, ref_model_bce


def make_args(**over):
    a = dict(use_pretrain=0, device="cpu", embed_dim=16, relation_dim=16, scale_gat_dim=None,
             use_residual=False, alpha=0.1, lamda=0.5, aggregation_type="gcn", n_conv_layers=1,
             conv_dim=16, mess_dropout=0.0, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
             pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=20,
             use_num_lit=False, use_txt_lit=False, milestone_score=0.5, n_mlp_layers=2,
             mlp_hidden_dim=12)
    a.update(over)
    return a


def random_graph(rng, n, e, n_rel, n_dup_pairs):
    """Distinct (h,r,t) triples, skewed heads, plus forced duplicate (h,t) pairs under a 2nd relation."""
    h = (n * rng.random(e) ** 1.7).astype(np.int64)
    t = rng.integers(0, n, e)
    r = rng.integers(0, n_rel, e)
    trip = np.unique(np.stack([h, r, t], 1), axis=0)
    extra = trip[rng.choice(len(trip), n_dup_pairs, replace=False)].copy()
    extra[:, 1] = (extra[:, 1] + 1 + rng.integers(0, n_rel - 1, n_dup_pairs)) % n_rel
    trip = np.unique(np.concatenate([trip, extra]), axis=0)
    trip = trip[rng.permutation(len(trip))]
    return trip[:, 0].copy(), trip[:, 2].copy(), trip[:, 1].copy()


def laplacian(n, h, t, r):
    """Same construction as the loader's random-walk sum over relations (own scipy code)."""
    import scipy.sparse as sp
    acc = None
    for rid in sorted(set(r.tolist())):
        m = r == rid
        adj = sp.coo_matrix((np.ones(m.sum()), (h[m], t[m])), shape=(n, n)).tocsr()
        deg = np.asarray(adj.sum(axis=1)).ravel()
        inv = np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0)
        lap = sp.diags(inv) @ adj
        acc = lap if acc is None else acc + lap
    acc = acc.tocoo()
    idx = torch.from_numpy(np.vstack([acc.row, acc.col]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(acc.data.astype(np.float32)), (n, n)).coalesce()


def sd_to_np(sd):
    return {"p/" + k: v.detach().cpu().numpy().copy() for k, v in sd.items() if k != "A_in"}


def build(cls, args, n, n_rel, a_in, num, txt, seed):
    torch.manual_seed(seed)
    m = cls(SimpleNamespace(**args), n, n_rel, a_in, num, txt)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if "gate_bias" in name or name.endswith(".bias"):
                prm.add_(0.05 * torch.randn_like(prm))   # make biases non-trivial
        if args["aggregation_type"] == "gin":
            # Aggregator.weight of 'gin' is re-allocated uninitialised (model.py:61): pin it
            for k in range(args["n_conv_layers"]):
                m.aggregator_layers[k].weight.uniform_(-0.25, 0.25)
    m.eval()
    return m


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def attention_case(ref_model, name, n, h, t, r, dim, seed):
    n_rel = int(r.max()) + 1
    args = make_args(embed_dim=dim, relation_dim=dim, conv_dim=dim)
    m = build(ref_model.LiteralKG, args, n, n_rel, None, None, None, seed)
    ht, tt, rt = (torch.from_numpy(x) for x in (h, t, r))
    logits = np.zeros(len(h), np.float32)
    with torch.no_grad():
        for rid in range(n_rel):
            sel = np.nonzero(r == rid)[0]
            if len(sel):
                logits[sel] = m.update_attention_batch(ht[sel], tt[sel], rid).numpy()
        m(ht, tt, rt, list(range(n_rel)), device=torch.device("cpu"), mode="update_att")
    a = m.A_in.data.coalesce()
    save(name, n=np.int64(n), h=h, t=t, r=r,
         entity=m.entity_embed.weight.detach().numpy(), relation=m.relation_embed.weight.detach().numpy(),
         logits=logits, a_indices=a.indices().numpy(), a_values=a.values().numpy())


def encoder_case(cls, name, args, n, h, t, r, seed, form, rng):
    n_rel = int(r.max()) + 1
    a_in = laplacian(n, h, t, r)
    num = torch.from_numpy(rng.random((n, args["num_lit_dim"])).astype(np.float32)) if args["use_num_lit"] else None
    txt = torch.from_numpy(rng.standard_normal((n, args["txt_lit_dim"])).astype(np.float32)) if args["use_txt_lit"] else None
    m = build(cls, args, n, n_rel, a_in, num, txt, seed)
    k = args["pre_training_neg_rate"]
    groups = 40
    bh = np.repeat(rng.integers(0, n, groups), k)
    br = np.repeat(rng.integers(0, n_rel, groups), k)
    bp = np.repeat(rng.integers(0, n, groups), k)
    bn = rng.integers(0, n, groups * k)
    tb = [torch.from_numpy(x) for x in (bh, br, bp, bn)]
    dev = torch.device("cpu")
    loss = m(*tb, device=dev, mode="pre_training")
    loss.backward()
    grads = {"g/" + k_: v.grad.detach().numpy() for k_, v in m.named_parameters()
             if v.grad is not None and k_ != "A_in"}
    with torch.no_grad():
        gat = m.gat_embeddings()
        hd, rl = gat[tb[0]], m.relation_embed(tb[1])
        if form == "transr":
            w = m.gat_trans_M[tb[1]]
            pr = lambda x: torch.bmm(x.unsqueeze(1), w).squeeze(1)  # noqa: E731
            ph, pp, pn = pr(hd), pr(gat[tb[2]]), pr(gat[tb[3]])
        else:
            ph, pp, pn = hd, gat[tb[2]], gat[tb[3]]
        pos = ((ph + rl - pp) ** 2).sum(1).numpy()
        neg = ((ph + rl - pn) ** 2).sum(1).numpy()
    arrs = dict(cfg=np.array(json.dumps(args)), form=np.array(form), n=np.int64(n), n_rel=np.int64(n_rel),
                h=h, t=t, r=r, a_indices=a_in.indices().numpy(), a_values=a_in.values().numpy(),
                bh=bh, br=br, bp=bp, bn=bn, gat=gat.numpy(), loss=loss.detach().numpy(),
                pos=pos, neg=neg)
    if num is not None:
        arrs["num"] = num.numpy()
    if txt is not None:
        arrs["txt"] = txt.numpy()
    arrs.update(sd_to_np(m.state_dict()))
    arrs.update(grads)
    if form == "transr":
        # f1 heads ride on the same encoder
        hid = torch.from_numpy(rng.integers(0, n, 9))
        tid = torch.from_numpy(rng.integers(0, n, 11))
        with torch.no_grad():
            arrs["score_heads"], arrs["score_tails"] = hid.numpy(), tid.numpy()
            arrs["score"] = m.calc_score(hid, tid).numpy()
            arrs["predict"] = m(hid, tid, device=dev, mode="predict").numpy()
        m.zero_grad()
        ft = m(tb[0], tb[2], tb[3], device=dev, mode="fine_tuning")
        ft.backward()
        arrs["ft_loss"] = ft.detach().numpy()
        arrs["ft_g/entity_embed.weight"] = m.entity_embed.weight.grad.numpy()
    save(name, **arrs)
    return m


def trajectory_case(ref_model, name, args, n, h, t, r, seed, n_steps, refresh_after, rng):
    """A short training run of the REFERENCE shaped like pre_training_train (main_pretraining.py:86-139):
    Adam steps on fresh batches, one update_att in the middle, losses recorded per step."""
    n_rel = int(r.max()) + 1
    a_in = laplacian(n, h, t, r)
    num = torch.from_numpy(rng.random((n, args["num_lit_dim"])).astype(np.float32)) if args["use_num_lit"] else None
    txt = torch.from_numpy(rng.standard_normal((n, args["txt_lit_dim"])).astype(np.float32)) if args["use_txt_lit"] else None
    m = build(ref_model.LiteralKG, args, n, n_rel, a_in, num, txt, seed)
    m.train()                                  # mess_dropout = 0 in args: train mode is deterministic
    init = sd_to_np(m.state_dict())
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    dev = torch.device("cpu")
    k, groups = args["pre_training_neg_rate"], 30
    batches, losses = [], []
    ht, tt, rt = (torch.from_numpy(x) for x in (h, t, r))
    for step in range(n_steps):
        b = [np.repeat(rng.integers(0, n, groups), k), np.repeat(rng.integers(0, n_rel, groups), k),
             np.repeat(rng.integers(0, n, groups), k), rng.integers(0, n, groups * k)]
        batches.append(np.stack(b))
        opt.zero_grad()
        loss = m(*[torch.from_numpy(x) for x in b], device=dev, mode="pre_training")
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if step == refresh_after:
            m(ht, tt, rt, list(range(n_rel)), device=dev, mode="update_att")
    final = {"f/" + k_: v.detach().numpy() for k_, v in m.state_dict().items() if k_ != "A_in"}
    a = m.A_in.data.coalesce()
    arrs = dict(cfg=np.array(json.dumps(args)), n=np.int64(n), n_rel=np.int64(n_rel), h=h, t=t, r=r,
                a_indices=a_in.indices().numpy(), a_values=a_in.values().numpy(), batches=np.stack(batches),
                losses=np.array(losses, np.float64), refresh_after=np.int64(refresh_after), lr=np.float64(1e-2),
                final_a_indices=a.indices().numpy(), final_a_values=a.values().numpy())
    if num is not None:
        arrs["num"] = num.numpy()
    if txt is not None:
        arrs["txt"] = txt.numpy()
    arrs.update(init)
    arrs.update(final)
    save(name, **arrs)


def loader_laplacian_case(name, n, h, t, r):
    """The reference's OWN create_adjacency_dict / create_laplacian_dict (dataloader.py:449-495) run on a bare
    DataLoader object holding just the attributes those two methods read (its __init__ needs pickles and
    files that are not shipped)."""
    import collections
    sys.path.insert(0, REF)
    import dataloader as ref_dl
    out = dict(n=np.int64(n), h=h, t=t, r=r)
    for kind in ("random-walk", "symmetric"):
        dl = object.__new__(ref_dl.DataLoader)
        dl.n_head_tail = n
        dl.laplacian_type = kind
        dl.train_relation_dict = collections.defaultdict(list)
        for hh, tt, rr in zip(h.tolist(), t.tolist(), r.tolist()):
            dl.train_relation_dict[rr].append((hh, tt))
        with np.errstate(divide="ignore"):
            dl.create_adjacency_dict()
            dl.create_laplacian_dict()
        a = dl.A_in.coalesce()
        key = kind.replace("-", "_")
        out[key + "_indices"] = a.indices().numpy()
        out[key + "_values"] = a.values().numpy()
    save(name, **out)


def loader_sampler_case(name, n, h, t, r, neg_rate, seed):
    """DataLoader.generate_kg_batch (dataloader.py:283-316 with its helpers 249-281, 318-330) run by the reference's own
    code under fixed seeds of the two generators it draws from (``random`` and ``numpy.random``).  Only the branch with
    fewer groups than heads (random.sample) can run: the other one calls random.choice on a dict_keys view, which raises
    TypeError under Python 3 (dataloader.py:290-291)."""
    import collections
    import random
    sys.path.insert(0, REF)
    import dataloader as ref_dl
    dl = object.__new__(ref_dl.DataLoader)
    dl.pre_training_neg_rate = neg_rate
    kg = collections.defaultdict(list)
    for hh, tt, rr in zip(h.tolist(), t.tolist(), r.tolist()):
        kg[hh].append((tt, rr))                                  # construct_data, dataloader.py:402
    tails = t.tolist()                                           # list(data.training_tails), main_pretraining.py:101
    out = dict(n=np.int64(n), h=h, t=t, r=r, neg_rate=np.int64(neg_rate), seed=np.int64(seed))
    half = {k: kg[k] for k in list(kg.keys())[::2]}              # an epoch's sampled dict (main_pretraining.py:93-96)
    for tag, d, bs in (("a", kg, 60 * neg_rate), ("b", half, 25 * neg_rate)):
        random.seed(seed)
        np.random.seed(seed)
        bh, br, bp, bn = dl.generate_kg_batch(d, bs, tails)
        out.update({f"{tag}_batch_size": np.int64(bs), f"{tag}_heads": np.array(list(d.keys()), np.int64),
                    f"{tag}_h": bh.numpy(), f"{tag}_r": br.numpy(), f"{tag}_p": bp.numpy(), f"{tag}_n": bn.numpy()})
    save(name, **out)


def mlp_case(cls, name, args, n, h, t, r, seed, rng, init_mlp):
    """mode='mlp' (model.py:499-519 / model_bce.py:423-436): train-mode forward (BatchNorm batch statistics, running
    buffers updated), BCE backward as main_finetuning_BCE.py does, then an eval-mode forward."""
    n_rel = int(r.max()) + 1
    a_in = laplacian(n, h, t, r)
    m = build(cls, args, n, n_rel, a_in, None, None, seed)
    if init_mlp:
        torch.manual_seed(seed + 1)
        m.initialize_MLP()
    with torch.no_grad():
        for bn in (m.norm1, m.norm2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.1)
    init = sd_to_np(m.state_dict())
    heads = torch.from_numpy(rng.integers(0, n, 96))
    tails = torch.from_numpy(rng.integers(0, n, 96))
    labels = torch.from_numpy((rng.random(96) > 0.5).astype(np.float32))
    dev = torch.device("cpu")
    m.train()
    out = m(heads, tails, device=dev, mode="mlp").reshape(-1)
    loss = torch.nn.functional.binary_cross_entropy(out, labels)
    loss.backward()
    grads = {"g/" + k_: v.grad.detach().numpy().copy() for k_, v in m.named_parameters()
             if v.grad is not None and k_ != "A_in"}
    after = {"after/" + k_: v.detach().numpy().copy() for k_, v in m.state_dict().items() if "running" in k_ or "tracked" in k_}
    m.eval()
    with torch.no_grad():
        out_eval = m(heads, tails, device=dev, mode="mlp").reshape(-1)
    arrs = dict(cfg=np.array(json.dumps(args)), n=np.int64(n), n_rel=np.int64(n_rel), h=h, t=t, r=r,
                a_indices=a_in.indices().numpy(), a_values=a_in.values().numpy(), heads=heads.numpy(),
                tails=tails.numpy(), labels=labels.numpy(), out_train=out.detach().numpy(), loss=loss.detach().numpy(),
                out_eval=out_eval.numpy(), init_mlp=np.bool_(init_mlp))
    arrs.update(init)
    arrs.update(grads)
    arrs.update(after)
    save(name, **arrs)


def main():
    ref_model, ref_model_bce = ref_modules()
    rng = np.random.default_rng(2022)
    if "--only-mlp" in sys.argv:
        rng = np.random.default_rng(515)
        mh, mt, mr = random_graph(rng, 220, 1500, 4, 6)
        mlp_case(ref_model.LiteralKG, "mlp_model_gcn_l1_scale", make_args(scale_gat_dim=24), 220, mh, mt, mr, 61, rng, True)
        mlp_case(ref_model_bce.LiteralKG, "mlp_bce_gcn_l2_scale", make_args(scale_gat_dim=16, n_conv_layers=2), 220, mh,
                 mt, mr, 62, rng, False)
        return
    if "--only-sampler" in sys.argv:
        rng = np.random.default_rng(909)
        sh, st, sr = random_graph(rng, 300, 2400, 5, 10)
        loader_sampler_case("sampler_ref_batch", 300, sh, st, sr, 4, 31)
        return
    if "--only-laplacian" in sys.argv:
        rng = np.random.default_rng(4242)
        lh, lt, lr = random_graph(rng, 260, 2600, 5, 10)       # (h,r,t) distinct, some (h,t) under two relations
        loader_laplacian_case("laplacian_rand260", 260, lh, lt, lr)
        return
    if "--only-trajectory" in sys.argv:
        rng = np.random.default_rng(777)
        th, tt_, tr = random_graph(rng, 200, 1400, 4, 6)
        trajectory_case(ref_model, "trajectory_gcn_l2_gatemul_scale",
                        make_args(n_conv_layers=2, scale_gat_dim=16, use_num_lit=True, use_txt_lit=True),
                        200, th, tt_, tr, 99, 6, 2, rng)
        trajectory_case(ref_model, "trajectory_bi_l1",
                        make_args(aggregation_type="bi-interaction"), 200, th, tt_, tr, 98, 5, 1, rng)
        return

    # ---- attention ------------------------------------------------------
    # 5-node toy with a duplicate (h,t) pair under two relations
    h = np.array([0, 0, 0, 1, 2, 2, 4], np.int64)
    t = np.array([1, 1, 2, 3, 0, 4, 4], np.int64)
    r = np.array([0, 1, 0, 1, 0, 1, 0], np.int64)
    attention_case(ref_model, "attention_toy5", 5, h, t, r, 8, 1)
    hh, tt, rr = random_graph(rng, 300, 2400, 5, 12)
    attention_case(ref_model, "attention_rand300", 300, hh, tt, rr, 32, 2)
    # slice of the reference's shipped KG file, chosen to contain its duplicate (h,t) pairs
    trip = np.loadtxt(os.path.join(REF, "data/Test/pre_training_train.txt"), dtype=np.int64)
    key = trip[:, 0] * (trip.max() + 1) + trip[:, 2]
    uk, cnt = np.unique(key, return_counts=True)
    dup_heads = np.unique(uk[cnt > 1] // (trip.max() + 1))[:12]
    sel = np.isin(trip[:, 0], dup_heads)
    sel[:1500] = True
    sub = trip[sel][:2000]
    ents, inv = np.unique(np.concatenate([sub[:, 0], sub[:, 2]]), return_inverse=True)
    sh, st = inv[:len(sub)].astype(np.int64), inv[len(sub):].astype(np.int64)
    attention_case(ref_model, "attention_testslice", len(ents), sh, st, sub[:, 1].copy(), 16, 3)

    # ---- encoder + TransR ----------------------------------------------
    n = 240
    gh, gt, gr = random_graph(rng, n, 1500, 4, 8)
    combos = [
        ("gcn_l1", dict()),
        ("gcn_l2_scale", dict(n_conv_layers=2, scale_gat_dim=24)),
        ("gcn_l2_res_wide", dict(n_conv_layers=2, use_residual=True, conv_dim=16)),
        ("gcn_l1_conv8", dict(conv_dim=8)),
        ("sage_l2", dict(aggregation_type="graphsage", n_conv_layers=2, conv_dim=12)),
        ("sage_l1_res", dict(aggregation_type="graphsage", use_residual=True)),
        ("bi_l2", dict(aggregation_type="bi-interaction", n_conv_layers=2)),
        ("bi_l1_res", dict(aggregation_type="bi-interaction", use_residual=True)),
        ("gin_l2", dict(aggregation_type="gin", n_conv_layers=2)),
        ("gin_l1_res", dict(aggregation_type="gin", use_residual=True)),
        ("gcn_l1_gatemul", dict(use_num_lit=True, use_txt_lit=True)),
        ("gcn_l2_gatenum", dict(use_num_lit=True, n_conv_layers=2)),
        ("gcn_l1_gatetxt_scale", dict(use_txt_lit=True, scale_gat_dim=16)),
    ]
    manifest = {}
    for i, (nm, over) in enumerate(combos):
        m = encoder_case(ref_model.LiteralKG, "encoder_" + nm, make_args(**over), n, gh, gt, gr, 10 + i,
                         "transr", rng)
        manifest[nm] = {k: list(v.shape) for k, v in m.state_dict().items()}

    # ---- TransE (model_bce) --------------------------------------------
    for i, (nm, over) in enumerate([
        ("gcn_l1", dict(scale_gat_dim=16)),
        ("gcn_l2_gatemul", dict(scale_gat_dim=16, n_conv_layers=2, use_num_lit=True, use_txt_lit=True)),
    ]):
        encoder_case(ref_model_bce.LiteralKG, "transe_" + nm, make_args(**over), n, gh, gt, gr, 40 + i,
                     "transe", rng)

    with open(os.path.join(OUT, "statedict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
