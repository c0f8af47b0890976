"""CPU: the oracle (oracle/literalkg_oracle.py) against the fixtures produced by the reference."""
import numpy as np
import pytest
import torch

from conftest import golden_cfg, golden_names, golden_params, load_golden
from oracle import literalkg_oracle as O

TOL = 1e-5   # oracle and reference run the same ATen ops on the same CPU: near bit-equal


def _a_in(g):
    n = int(g["n"])
    return torch.sparse_coo_tensor(torch.from_numpy(g["a_indices"]), torch.from_numpy(g["a_values"]), (n, n)).coalesce()


@pytest.mark.parametrize("name", golden_names("attention_"))
def test_attention(name):
    g = load_golden(name)
    n = int(g["n"])
    ent, rel = torch.from_numpy(g["entity"]), torch.from_numpy(g["relation"])
    h, t, r = (torch.from_numpy(g[k]) for k in "htr")
    np.testing.assert_allclose(O.edge_logits(ent, rel, h, t, r).numpy(), g["logits"], rtol=TOL, atol=TOL)
    a = O.attention_refresh(n, ent, rel, h, t, r).coalesce()
    assert np.array_equal(a.indices().numpy(), g["a_indices"])          # bit-exact int64 indices
    np.testing.assert_allclose(a.values().numpy(), g["a_values"], rtol=TOL, atol=1e-7)
    rows, cols, vals = O.attention_refresh_explicit(n, ent, rel, h, t, r)
    assert np.array_equal(torch.stack([rows, cols]).numpy(), g["a_indices"])
    np.testing.assert_allclose(vals.numpy(), g["a_values"], rtol=1e-5, atol=1e-7)
    # rows of the refreshed matrix sum to one, duplicate pairs are merged
    rs = np.zeros(n)
    np.add.at(rs, g["a_indices"][0], g["a_values"])
    assert np.allclose(rs[np.unique(g["a_indices"][0])], 1.0, atol=1e-5)
    assert g["a_indices"].shape[1] <= len(g["h"])


@pytest.mark.parametrize("name", golden_names("encoder_") + golden_names("transe_"))
def test_encoder_loss_grads(name):
    g = load_golden(name)
    cfg, form = golden_cfg(g), str(g["form"])
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in golden_params(g).items()}
    a = _a_in(g)
    num = torch.from_numpy(g["num"]) if "num" in g else None
    txt = torch.from_numpy(g["txt"]) if "txt" in g else None
    bh, br, bp, bn = (torch.from_numpy(g[k]) for k in ("bh", "br", "bp", "bn"))
    gat = O.gat_embeddings(p, cfg, a, num, txt)
    np.testing.assert_allclose(gat.detach().numpy(), g["gat"], rtol=TOL, atol=TOL)
    if form == "transr":
        pos, neg, _ = O.triple_scores_transr(p, gat, bh, br, bp, bn)
        loss = O.triple_loss_transr(p, cfg, gat, bh, br, bp, bn)
    else:
        pos, neg, _ = O.triple_scores_transe(p, gat, bh, br, bp, bn)
        loss = O.triple_loss_transe(p, cfg, gat, bh, br, bp, bn)
    np.testing.assert_allclose(pos.detach().numpy(), g["pos"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(neg.detach().numpy(), g["neg"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=TOL)
    if form == "transr":   # the relation-by-relation form the big-batch GPU tests use is the same function
        pos2, neg2, _ = O.triple_scores_transr_by_relation(p, gat, bh, br, bp, bn)
        np.testing.assert_allclose(pos2.detach().numpy(), g["pos"], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(neg2.detach().numpy(), g["neg"], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(O.triple_loss_transr(p, cfg, gat, bh, br, bp, bn, by_relation=True).item(),
                                   g["loss"], rtol=TOL)
    loss.backward()
    checked = 0
    for k, v in g.items():
        if k.startswith("g/"):
            got = p[k[2:]].grad
            assert got is not None, k
            np.testing.assert_allclose(got.numpy(), v, rtol=1e-4, atol=1e-7, err_msg=k)
            checked += 1
    assert checked >= 4
    # parameters the reference left without a gradient stay without one here too
    for k, v in p.items():
        if "g/" + k not in g:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("name", golden_names("encoder_"))
def test_heads(name):
    g = load_golden(name)
    cfg = golden_cfg(g)
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in golden_params(g).items()}
    num = torch.from_numpy(g["num"]) if "num" in g else None
    txt = torch.from_numpy(g["txt"]) if "txt" in g else None
    gat = O.gat_embeddings(p, cfg, _a_in(g), num, txt)
    hid, tid = torch.from_numpy(g["score_heads"]), torch.from_numpy(g["score_tails"])
    np.testing.assert_allclose(O.link_scores(gat, hid, tid).detach().numpy(), g["score"], rtol=TOL, atol=TOL)
    assert np.array_equal(O.predict_links(cfg, gat.detach(), hid, tid).numpy(), g["predict"])
    bh, bp, bn = (torch.from_numpy(g[k]) for k in ("bh", "bp", "bn"))
    ft = O.prediction_loss(cfg, gat, bh, bp, bn)
    np.testing.assert_allclose(ft.item(), g["ft_loss"], rtol=TOL)
    ft.backward()
    np.testing.assert_allclose(p["entity_embed.weight"].grad.numpy(), g["ft_g/entity_embed.weight"],
                               rtol=1e-4, atol=1e-7)


def test_laplacian_matches_fixture_inputs():
    g = load_golden("encoder_gcn_l1")
    h, t, r = (torch.from_numpy(g[k]) for k in "htr")
    a = O.laplacian_a_in(int(g["n"]), h, t, r)
    assert np.array_equal(a.indices().numpy(), g["a_indices"])
    np.testing.assert_allclose(a.values().numpy(), g["a_values"], rtol=1e-6)


@pytest.mark.parametrize("name", golden_names("trajectory_"))
def test_training_trajectory(name):
    """Oracle + torch Adam replays the reference's short training run (Adam steps, update_att in the middle)."""
    g = load_golden(name)
    cfg = golden_cfg(g)
    n = int(g["n"])
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in golden_params(g).items()}
    a = torch.sparse_coo_tensor(torch.from_numpy(g["a_indices"]), torch.from_numpy(g["a_values"]), (n, n)).coalesce()
    num = torch.from_numpy(g["num"]) if "num" in g else None
    txt = torch.from_numpy(g["txt"]) if "txt" in g else None
    opt = torch.optim.Adam([v for v in p.values() if v.requires_grad], lr=float(g["lr"]))
    h, t, r = (torch.from_numpy(g[k]) for k in "htr")
    for step, b in enumerate(g["batches"]):
        opt.zero_grad()
        loss = O.pre_training_loss(p, cfg, a, *[torch.from_numpy(x) for x in b], num=num, txt=txt)
        loss.backward()
        opt.step()
        np.testing.assert_allclose(float(loss), g["losses"][step], rtol=1e-5, err_msg=f"step {step}")
        if step == int(g["refresh_after"]):
            with torch.no_grad():
                a = O.attention_refresh(n, p["entity_embed.weight"], p["relation_embed.weight"], h, t, r).coalesce()
    assert np.array_equal(a.indices().numpy(), g["final_a_indices"])
    np.testing.assert_allclose(a.values().numpy(), g["final_a_values"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(p["entity_embed.weight"].detach().numpy(), g["f/entity_embed.weight"], rtol=1e-4,
                               atol=1e-6)


# ----------------------------------------------------------------------------- plain-C restatement (oracle/lkg_oracle.c)
@pytest.mark.parametrize("name", golden_names("attention_"))
def test_c_oracle_attention(name):
    from oracle import c_oracle
    g = load_golden(name)
    rows, cols, vals = c_oracle.attention(g["h"], g["t"], g["r"], g["entity"], g["relation"])
    assert np.array_equal(np.stack([rows, cols]), g["a_indices"])            # bit-exact vs the reference's coalesce
    np.testing.assert_allclose(vals, g["a_values"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["encoder_gcn_l1", "encoder_sage_l2"])
def test_c_oracle_spmm(name):
    from oracle import c_oracle
    g = load_golden(name)
    n = int(g["n"])
    a = _a_in(g)
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, g["a_indices"][0] + 1, 1)
    x = g["p/entity_embed.weight"]
    got = c_oracle.spmm(np.cumsum(rowptr), g["a_indices"][1], g["a_values"], x)
    np.testing.assert_allclose(got, O.aggregate(a, torch.from_numpy(x)).numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", golden_names("mlp_"))
def test_mlp_head(name):
    g = load_golden(name)
    cfg = golden_cfg(g)
    p = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in golden_params(g).items()}
    gat = O.gat_embeddings(p, cfg, _a_in(g))
    heads, tails = torch.from_numpy(g["heads"]), torch.from_numpy(g["tails"])
    out = O.mlp_head(p, gat, heads, tails, training=True).reshape(-1)
    np.testing.assert_allclose(out.detach().numpy(), g["out_train"], rtol=TOL, atol=1e-6)
    loss = torch.nn.functional.binary_cross_entropy(out, torch.from_numpy(g["labels"]))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=TOL)
    loss.backward()
    for k, v in g.items():
        if k.startswith("g/"):
            np.testing.assert_allclose(p[k[2:]].grad.numpy(), v, rtol=1e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(p["norm1.running_var"].numpy(), g["after/norm1.running_var"], rtol=1e-5)
    with torch.no_grad():
        out_eval = O.mlp_head(p, O.gat_embeddings(p, cfg, _a_in(g)), heads, tails, training=False).reshape(-1)
    np.testing.assert_allclose(out_eval.numpy(), g["out_eval"], rtol=TOL, atol=1e-6)


# ----------------------------------------------------------------------------- f2 batch sampler restatement
def test_sampler_oracle_reproduces_the_reference_batches():
    """oracle/sampler_oracle.py draws from the same generators in the same order as dataloader.py:249-330: under the
    fixture's seeds it returns the reference's batch element for element."""
    import random
    from oracle import sampler_oracle as S
    g = load_golden("sampler_ref_batch")
    kg = S.build_kg_dict(g["h"], g["t"], g["r"])
    tails = g["t"].tolist()
    k, seed = int(g["neg_rate"]), int(g["seed"])
    for tag in ("a", "b"):
        d = {int(h): kg[int(h)] for h in g[f"{tag}_heads"]}
        random.seed(seed)
        np.random.seed(seed)
        bh, br, bp, bn = S.generate_kg_batch(d, int(g[f"{tag}_batch_size"]), k, tails)
        assert np.array_equal(bh, g[f"{tag}_h"]) and np.array_equal(br, g[f"{tag}_r"])
        assert np.array_equal(bp, g[f"{tag}_p"]) and np.array_equal(bn, g[f"{tag}_n"])
        # the contract the device sampler is held to as well
        pos = set(zip(g["h"].tolist(), g["r"].tolist(), g["t"].tolist()))
        for hh, rr, pp, negs in zip(bh[::k], br[::k], bp[::k], bn.reshape(-1, k)):
            assert (hh, rr, pp) in pos and len(set(negs.tolist())) == k
            assert all((hh, rr, x) not in pos for x in negs)


# ----------------------------------------------------------------------------- the fuzz sweep's second arbiter
@pytest.mark.parametrize("agg", ["gcn", "graphsage", "bi-interaction", "gin"])
def test_device_association_of_the_residual_products_is_the_same_function(agg):
    """tests/test_gpu_fuzz.py arbitrates gradient differences next to a LeakyReLU kink with the oracle evaluated in the DEVICE's
    association of the residual layers' products (two small matrices first, one N-row product).  In float64 that restatement
    must be the oracle itself: loss and every gradient to 1e-11 on a three-layer residual model of each aggregator."""
    from test_gpu_fuzz import device_association
    from literalkg_amd.synth import make_batch, make_kg
    from literalkg_amd import io
    import literalkg_amd as L
    n, dim = 400, 24
    h, t, r = make_kg(n, 1600, "uniform", seed=3)
    cfg = O.default_cfg(embed_dim=dim, relation_dim=dim, conv_dim=dim, n_conv_layers=3, aggregation_type=agg,
                        use_residual=True, scale_gat_dim=20, mlp_hidden_dim=16)
    torch.manual_seed(2)
    a_in = io.initial_a_in(n, h, t, r).double()
    m = L.LiteralKG(cfg, n, 16, io.initial_a_in(n, h, t, r), None, None)
    params = {k: v.detach().double() for k, v in m.state_dict().items() if k != "A_in"}
    batch = [torch.from_numpy(x) for x in make_batch(n, 30, 3, seed=4)]

    def grads(device_form):
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        if device_form:
            with device_association(O):
                loss = O.pre_training_loss(p, cfg, a_in, *batch)
        else:
            loss = O.pre_training_loss(p, cfg, a_in, *batch)
        loss.backward()
        return float(loss), {k: v.grad for k, v in p.items() if v.grad is not None}

    (l0, g0), (l1, g1) = grads(False), grads(True)
    assert abs(l0 - l1) <= 1e-12 * abs(l0)
    assert g0.keys() == g1.keys() and any("linear_h0" in k for k in g0)
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-11 * (float(g0[k].abs().max()) + 1e-300), k
