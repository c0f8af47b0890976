"""-m gpu: the HIP path on every configuration BASELINE.json names, at the named sizes, against the oracle.

    C1  data/Small shape: 765 957 entity rows (125 422 used, ~84 % empty head rows), 252 k triples, D=64, 1 layer
    C2  1 M entities / 10 M edges, D=128, 1 layer
    C3  1 M / 10 M, D=256, 2 layers + literal gate (GateMul, text literals 300 wide)
    C4  5 M / 100 M, D=256  (the 8-GPU config's whole graph on ONE GPU: it fits 288 GB)
    C5  5 M / 100 M, D=512, 3 layers, TransR W_r projection, K=256 negatives, B=32 768

The oracle cannot evaluate a 1 M-row model in seconds, and does not have to: layer-k rows of a set S depend only on
S's k-hop out-neighbourhood, so the torch oracle is evaluated EXACTLY on the relabelled frontier sub-problem (all
triples of the rows that need them, the rows' embeddings / literals, the same weights) -- loss, link-prediction
scores, every weight gradient and the entity-gradient rows are then comparable one to one (tolerance 1e-4 on scores,
as BASELINE.json states).  The sparse kernels are additionally held to the plain-C oracle on sampled rows, and to
size-independent properties over the whole graph.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(gpu_device):
    import __graft_entry__ as ge
    ge.build()
    import literalkg_amd
    return literalkg_amd


@pytest.fixture(scope="module")
def ops(L):
    from literalkg_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def O():
    from oracle import literalkg_oracle
    return literalkg_oracle


@pytest.fixture(scope="module")
def kg_1m(L, gpu_device):
    """C2 / C3 graph: 1 M entities, 10 M triples, zipf heads (bench.py's graph)."""
    from literalkg_amd.synth import make_kg
    n = 1_000_000
    h, t, r = make_kg(n, 10_000_000)
    return n, h, t, r, L.KGStructure.from_triples(n, h, t, r, device=gpu_device)


@pytest.fixture(scope="module")
def kg_5m(L, gpu_device):
    """C4 / C5 graph: 5 M entities, 100 M triples."""
    from literalkg_amd.synth import make_kg
    n = 5_000_000
    h, t, r = make_kg(n, 100_000_000)
    return n, h, t, r, L.KGStructure.from_triples(n, h, t, r, device=gpu_device)


# ----------------------------------------------------------------------------- helpers
def frontier(h, t, seeds, hops):
    """Nested row sets F[hops] = seeds, F[k-1] = F[k] + tails(F[k]).  Returns (nodes = F[0] sorted, mask of the triples
    whose head is in F[1], i.e. of every row whose aggregation is needed)."""
    need = np.unique(seeds)
    mask = None
    for _ in range(hops):
        mask = np.isin(h, need)
        need = np.union1d(need, t[mask])
    return need, mask


def sub_problem(h, t, r, seeds, hops):
    nodes, mask = frontier(h, t, seeds, hops)
    pos = lambda x: np.searchsorted(nodes, x)
    return nodes, pos, (torch.from_numpy(pos(h[mask])), torch.from_numpy(pos(t[mask])), torch.from_numpy(r[mask]))


def c_oracle_rows_check(ops, g, val, ent_dev, rel_dev, x_dev, out_dev, h, t, r, rows, d):
    """`rows` of the attention refresh and of out = A @ x against oracle/lkg_oracle.c (relabelled to the rows' tails)."""
    from oracle import c_oracle
    mask = np.isin(h, rows)
    nodes = np.union1d(rows, t[mask])
    pos = lambda x: np.searchsorted(nodes, x)
    nd = torch.from_numpy(nodes).to(ent_dev.device)
    ent_sub = ent_dev[nd].cpu().numpy()
    rr, cc, want = c_oracle.attention(pos(h[mask]), pos(t[mask]), r[mask], ent_sub, rel_dev.cpu().numpy())
    rp = g.host("rowptr")
    sel = np.concatenate([np.arange(rp[i], rp[i + 1]) for i in rows])           # rows is sorted: (h,t) order
    assert np.array_equal(nodes[rr], np.repeat(rows, [rp[i + 1] - rp[i] for i in rows]))    # indices bit-exact
    assert np.array_equal(nodes[cc], g.host("col")[sel].astype(np.int64))
    got_val = val[torch.from_numpy(sel).to(val.device)].cpu().numpy()
    np.testing.assert_allclose(got_val, want, rtol=1e-4, atol=1e-7)
    sub_rp = np.r_[0, np.cumsum([rp[i + 1] - rp[i] for i in rows])]
    ref = c_oracle.spmm(sub_rp, cc, got_val, x_dev[nd].cpu().numpy())
    got = out_dev[torch.from_numpy(rows).to(out_dev.device)].cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)


def sparse_properties(ops, g, val, d, dev, scale=1.0):
    """Size-independent properties over the WHOLE graph: softmax rows sum to one, linearity of the SpMM, and
    <A x, y> = <x, A^T y> (forward kernel against the transpose kernel)."""
    n = g.n
    rp = g.host("rowptr")
    ones = torch.ones(n, 4, device=dev)
    rs = ops.spmm_raw(g.rowptr, g.col, val, ones, n, long_rows=g.long_rows(False))[:, 0]
    nonempty = torch.from_numpy(np.diff(rp) > 0).to(dev)
    assert float((rs[nonempty] - 1).abs().max()) < 1e-5
    assert float(rs[~nonempty].abs().max()) == 0.0
    del ones, rs
    x1 = torch.randn(n, d, device=dev) * scale
    a = ops.spmm_raw(g.rowptr, g.col, val, x1, n, long_rows=g.long_rows(False))
    y = torch.randn(n, d, device=dev)
    aty = ops.spmm_raw(g.t_rowptr, g.t_col, ops.permute_values(val, g.t_perm), y, n, long_rows=g.long_rows(True))
    lhs, rhs = float((a.double() * y.double()).sum()), float((x1.double() * aty.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0) + 1e-2 * scale
    del y, aty
    x2 = torch.randn(n, d, device=dev) * scale
    b = ops.spmm_raw(g.rowptr, g.col, val, x2, n, long_rows=g.long_rows(False))
    x2.mul_(2).add_(x1)
    ab = ops.spmm_raw(g.rowptr, g.col, val, x2, n, long_rows=g.long_rows(False))
    b.mul_(2).add_(a)
    assert float((ab - b).abs().max()) < 1e-4 * max(scale, 1.0)
    return x1, a


def module_vs_frontier_oracle(L, O, dev, n, h, t, r, cfg, num, txt, seeds_batch, scoring="transr", ent_scale=100.0,
                              by_relation=False, seed=11):
    """Build the module at full size, refresh the attention, run one pre_training step (eval mode: no dropout) and
    compare loss / scores / gradients with the oracle on the batch's L-hop frontier sub-problem."""
    torch.manual_seed(seed)
    m = L.LiteralKG(cfg, n, 16, None, num, txt, scoring=scoring)
    with torch.no_grad():     # xavier at N = 1 M is ~2e-3: scale up so that tanh / sigmoid / softmax all matter
        m.entity_embed.weight.mul_(ent_scale)
        m.relation_embed.weight.mul_(3)
    m.to(dev).eval()
    hd, td, rd = (torch.from_numpy(x).to(dev) for x in (h, t, r))
    m(hd, td, rd, list(range(16)), device=dev, mode="update_att")
    bh, br, bp, bn = (torch.from_numpy(x) for x in seeds_batch)
    loss = m(bh.to(dev), br.to(dev), bp.to(dev), bn.to(dev), device=dev, mode="pre_training")
    loss.backward()
    seeds = np.unique(np.concatenate([x.numpy() for x in (bh, bp, bn)]))
    nodes, pos, (hs, ts, rs) = sub_problem(h, t, r, seeds, cfg.n_conv_layers)
    nd = torch.from_numpy(nodes).to(dev)
    p = {}
    for k, v in m.state_dict().items():
        if k == "A_in":
            continue
        v = v.detach()
        p[k] = (v[nd] if k == "entity_embed.weight" else v).cpu().clone().requires_grad_(v.is_floating_point())
    a_sub = O.attention_refresh(len(nodes), p["entity_embed.weight"].detach(), p["relation_embed.weight"].detach(),
                                hs, ts, rs)
    num_s = num[nd].cpu() if num is not None else None
    txt_s = txt[nd].cpu() if txt is not None else None
    sb = [torch.from_numpy(pos(x.numpy())) for x in (bh, bp, bn)]
    gat = O.gat_embeddings(p, cfg, a_sub, num_s, txt_s)
    if scoring == "transr":
        want = O.triple_loss_transr(p, cfg, gat, sb[0], br, sb[1], sb[2], by_relation)
    else:
        want = O.triple_loss_transe(p, cfg, gat, sb[0], br, sb[1], sb[2])
    want.backward()
    # loss and the embeddings of the batch's entities
    np.testing.assert_allclose(float(loss), float(want), rtol=1e-4)
    sd = torch.from_numpy(seeds).to(dev)
    got_rows = m.gat_embed.detach()[sd].cpu()
    torch.testing.assert_close(got_rows, gat.detach()[torch.from_numpy(pos(seeds))], rtol=1e-4, atol=1e-4)
    # link-prediction scores of the sampled entities (BASELINE.json: within 1e-4)
    heads, tails = sd[:64], sd[-64:]
    score = m.calc_score(heads, tails).cpu()
    ref_score = O.link_scores(gat.detach(), torch.from_numpy(pos(seeds[:64])), torch.from_numpy(pos(seeds[-64:])))
    assert float((score - ref_score).abs().max()) <= 1e-4 * max(1.0, float(ref_score.abs().max()))
    # refreshed attention: the frontier rows of A_in, indices bit-exact
    a_full = m.A_in.data
    rows_needed = np.unique(hs.numpy())
    idx = a_full.indices()
    keep = torch.isin(idx[0], nd[torch.from_numpy(rows_needed).to(dev)])
    sub_idx = idx[:, keep].cpu().numpy()
    ref_a = a_sub.coalesce()
    assert np.array_equal(sub_idx, nodes[ref_a.indices().numpy()])
    np.testing.assert_allclose(a_full.values()[keep].cpu().numpy(), ref_a.values().numpy(), rtol=1e-4, atol=1e-7)
    # gradients: every weight, and the rows of the entity table the loss can reach (all other rows are exactly 0)
    for k, v in m.named_parameters():
        if k == "A_in" or v.grad is None:
            continue
        ref = p[k].grad
        got = v.grad
        if k == "entity_embed.weight":
            rest = got.clone()
            rest[nd] = 0
            assert float(rest.abs().max()) == 0.0                     # rows outside the frontier: exactly zero
            del rest
            got = got[nd]
        scale_ = float(ref.abs().max()) + 1e-12
        err = float((got.cpu() - ref).abs().max()) / scale_
        assert err < 2e-3, (k, err)
    return m


def batch_for(n, groups, k, h, seed):
    """generate_kg_batch layout: each head K consecutive entries with the same (h, r, t+) and K negatives; heads with
    at least one triple, drawn from the graph so that their rows are non-trivial."""
    rng = np.random.default_rng(seed)
    heads = h[rng.integers(0, len(h), groups)]
    return (np.repeat(heads, k), np.repeat(rng.integers(0, 16, groups), k),
            np.repeat(rng.integers(0, n, groups), k), rng.integers(0, n, groups * k))


# ----------------------------------------------------------------------------- C1
def test_config_c1_small_kg_shape_d64(L, O, gpu_device):
    """data/Small's shape: a sparse id space -- 765 957 rows of which 125 422 are used -- with
    252 k triples, D=64, 1 layer; the WHOLE model is small enough for the full oracle."""
    from literalkg_amd.synth import make_kg
    n, used, e = 765_957, 125_422, 252_000
    hh, tt, r = make_kg(used, e, seed=64)
    ids = np.sort(np.random.default_rng(1).choice(n, used, replace=False))
    h, t = ids[hh], ids[tt]
    assert 1.0 - len(np.unique(h)) / n > 0.84                      # mostly empty head rows
    cfg = O.default_cfg(embed_dim=64, relation_dim=64, conv_dim=64, n_conv_layers=1, aggregation_type="gcn",
                        kg_l2loss_lambda=1e-4, device=gpu_device)
    torch.manual_seed(5)
    m = L.LiteralKG(cfg, n, 16)
    with torch.no_grad():
        m.entity_embed.weight.mul_(100)
        m.relation_embed.weight.mul_(3)
    params = {k: v.detach().clone() for k, v in m.state_dict().items() if k != "A_in"}
    m.to(gpu_device).eval()
    hd, td, rd = (torch.from_numpy(x).to(gpu_device) for x in (h, t, r))
    m(hd, td, rd, list(range(16)), device=gpu_device, mode="update_att")
    ref_a = O.attention_refresh(n, params["entity_embed.weight"], params["relation_embed.weight"],
                                torch.from_numpy(h), torch.from_numpy(t), torch.from_numpy(r)).coalesce()
    got_a = m.A_in.data.cpu()
    assert torch.equal(got_a.indices(), ref_a.indices())
    torch.testing.assert_close(got_a.values(), ref_a.values(), rtol=1e-4, atol=1e-6)
    batch = [torch.from_numpy(x) for x in batch_for(n, 683, 3, h, 3)]
    loss = m(*[b.to(gpu_device) for b in batch], device=gpu_device, mode="pre_training")
    loss.backward()
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in params.items()}
    want = O.pre_training_loss(p, cfg, ref_a, *batch)
    want.backward()
    np.testing.assert_allclose(float(loss), float(want), rtol=1e-4)
    torch.testing.assert_close(m.gat_embed.detach().cpu(), O.gat_embeddings(params, cfg, ref_a), rtol=1e-4, atol=1e-4)
    for k, v in m.named_parameters():
        if v.grad is not None and k != "A_in":
            ref = p[k].grad
            err = float((v.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
            assert err < 2e-3, (k, err)


# ----------------------------------------------------------------------------- C2
def test_config_c2_1m_10m_d128_one_layer(L, ops, O, gpu_device, kg_1m):
    from literalkg_amd.synth import xavier_table
    n, h, t, r, g = kg_1m
    d = 128
    ent = xavier_table(n, d, gpu_device) * 100
    rel = xavier_table(16, d, gpu_device, seed=7) * 3
    val, _ = ops.edge_softmax(g, ent, rel)
    x1, a = sparse_properties(ops, g, val, d, gpu_device)
    rows = np.sort(np.random.default_rng(0).choice(np.flatnonzero(np.diff(g.host("rowptr")) > 0), 200, replace=False))
    c_oracle_rows_check(ops, g, val, ent, rel, x1, a, h, t, r, rows, d)
    # transpose kernel on sampled tail rows: CSC row = the heads pointing at a tail
    y = torch.randn(n, d, device=gpu_device)
    val_t = ops.permute_values(val, g.t_perm)
    aty = ops.spmm_raw(g.t_rowptr, g.t_col, val_t, y, n, long_rows=g.long_rows(True))
    trp = g.host("t_rowptr")
    for i in np.random.default_rng(1).choice(n, 50, replace=False):
        sl = slice(int(trp[i]), int(trp[i + 1]))
        want = (val_t[sl][:, None].double() * y[g.t_col[sl].long()].double()).sum(0).float()
        assert float((aty[i] - want).abs().max()) < 1e-4
    del x1, a, y, aty, ent, rel
    # the drop-in module at this size: 1 layer, D=128, TransR, batch of 683 groups x 3 (main_pretraining defaults)
    cfg = O.default_cfg(embed_dim=d, relation_dim=d, conv_dim=d, n_conv_layers=1, aggregation_type="gcn",
                        kg_l2loss_lambda=1e-4, device=gpu_device)
    module_vs_frontier_oracle(L, O, gpu_device, n, h, t, r, cfg, None, None, batch_for(n, 683, 3, h, 2))


# ----------------------------------------------------------------------------- C3
def test_config_c3_1m_10m_d256_two_layers_gatemul(L, O, gpu_device, kg_1m):
    n, h, t, r, g = kg_1m
    d = 256
    cfg = O.default_cfg(embed_dim=d, relation_dim=d, conv_dim=d, n_conv_layers=2, aggregation_type="gcn",
                        use_num_lit=True, use_txt_lit=True, num_lit_dim=2, txt_lit_dim=300, kg_l2loss_lambda=1e-4,
                        device=gpu_device)
    gen = torch.Generator(device=gpu_device).manual_seed(4)
    num = torch.rand(n, 2, device=gpu_device, generator=gen)
    txt = torch.randn(n, 300, device=gpu_device, generator=gen)
    m = module_vs_frontier_oracle(L, O, gpu_device, n, h, t, r, cfg, num, txt, batch_for(n, 64, 3, h, 5))
    assert m.gat_embed.shape == (n, 3 * d)


# ----------------------------------------------------------------------------- C4
def test_config_c4_5m_100m_d256_on_one_gpu(L, ops, gpu_device, kg_5m):
    from literalkg_amd.synth import xavier_table
    n, h, t, r, g = kg_5m
    d = 256
    assert g.nnz < len(h) and g.has_dups
    rp = g.host("rowptr")
    assert rp[-1] == g.nnz and np.all(np.diff(rp) >= 0)
    ent = xavier_table(n, d, gpu_device) * 200
    rel = xavier_table(16, d, gpu_device, seed=7) * 3
    val, _ = ops.edge_softmax(g, ent, rel)
    x1, a = sparse_properties(ops, g, val, d, gpu_device)
    rows = np.sort(np.random.default_rng(0).choice(np.flatnonzero(np.diff(rp) > 0), 200, replace=False))
    c_oracle_rows_check(ops, g, val, ent, rel, x1, a, h, t, r, rows, d)


# ----------------------------------------------------------------------------- C5
def test_config_c5_5m_100m_d512_sparse_path(L, ops, gpu_device, kg_5m):
    from literalkg_amd.synth import xavier_table
    n, h, t, r, g = kg_5m
    d = 512
    ent = xavier_table(n, d, gpu_device) * 200
    rel = xavier_table(16, d, gpu_device, seed=7) * 3
    val, _ = ops.edge_softmax(g, ent, rel)
    x1, a = sparse_properties(ops, g, val, d, gpu_device)
    rows = np.sort(np.random.default_rng(3).choice(np.flatnonzero(np.diff(g.host("rowptr")) > 0), 200, replace=False))
    c_oracle_rows_check(ops, g, val, ent, rel, x1, a, h, t, r, rows, d)


def test_config_c5_transr_k256_b32768(ops, O, gpu_device):
    """The W_r projection + scoring at C5's batch: C = 2048 (512 x (3+1)), D = 512, K = 256 negatives, 128 groups.
    The oracle projects relation by relation (the B x C x D gather of the reference is 137 GB here)."""
    n, c, dout, k, groups = 100_000, 2048, 512, 256, 128
    from literalkg_amd.synth import make_batch
    gen = torch.Generator().manual_seed(2)
    emb = torch.randn(n, c, generator=gen) * 0.2
    relemb = torch.randn(16, dout, generator=gen) * 0.2
    wm = torch.randn(16, c, dout, generator=gen) * (1.0 / c ** 0.5)
    bh, br, bp, bn = (torch.from_numpy(x) for x in make_batch(n, groups, k, seed=8))
    lam = 1e-3
    pg = [x.to(gpu_device).requires_grad_(True) for x in (emb, relemb, wm)]
    keep = {}
    dev_b = [x.to(gpu_device) for x in (bh, br, bp, bn)]
    assert ops.is_grouped_batch(dev_b[0], dev_b[1], dev_b[2], k)      # h and t+ projected once per 256 rows
    loss = ops.transr_loss(pg[0], pg[1], pg[2], *dev_b, lam, keep, k)
    loss.backward()
    pc = {"relation_embed.weight": relemb.clone().requires_grad_(True), "gat_trans_M": wm.clone().requires_grad_(True)}
    gat = emb.clone().requires_grad_(True)
    cfg = O.default_cfg(kg_l2loss_lambda=lam)
    pos, neg, _ = O.triple_scores_transr_by_relation(pc, gat, bh, br, bp, bn)
    want = O.triple_loss_transr(pc, cfg, gat, bh, br, bp, bn, by_relation=True)
    want.backward()
    np.testing.assert_allclose(float(loss), float(want), rtol=1e-5)
    torch.testing.assert_close(keep["pos"].cpu(), pos.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(keep["neg"].cpu(), neg.detach(), rtol=1e-4, atol=1e-4)
    for got, ref, name in ((pg[0].grad, gat.grad, "emb"), (pg[1].grad, pc["relation_embed.weight"].grad, "rel"),
                           (pg[2].grad, pc["gat_trans_M"].grad, "W")):
        err = float((got.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
        assert err < 1e-3, (name, err)


def test_config_c5_module_three_layers_transr_k256(L, O, gpu_device):
    """C5's module (D=512, 3 layers, TransR, K=256) at the config's average degree (20) on a graph the FULL oracle can
    still evaluate on the host: 100 k entities / 2 M triples, 16 groups x 256 negatives."""
    from literalkg_amd.synth import make_batch, make_kg
    n, e, d = 100_000, 2_000_000, 512
    h, t, r = make_kg(n, e, seed=9)
    cfg = O.default_cfg(embed_dim=d, relation_dim=d, conv_dim=d, n_conv_layers=3, aggregation_type="gcn",
                        kg_l2loss_lambda=1e-4, pre_training_neg_rate=256, device=gpu_device)
    torch.manual_seed(6)
    m = L.LiteralKG(cfg, n, 16)
    with torch.no_grad():
        m.entity_embed.weight.mul_(40)
        m.relation_embed.weight.mul_(3)
    params = {k: v.detach().clone() for k, v in m.state_dict().items() if k != "A_in"}
    m.to(gpu_device).eval()
    hd, td, rd = (torch.from_numpy(x).to(gpu_device) for x in (h, t, r))
    m(hd, td, rd, list(range(16)), device=gpu_device, mode="update_att")
    a_ref = O.attention_refresh(n, params["entity_embed.weight"], params["relation_embed.weight"],
                                torch.from_numpy(h), torch.from_numpy(t), torch.from_numpy(r)).coalesce()
    assert torch.equal(m.A_in.data.indices().cpu(), a_ref.indices())
    torch.testing.assert_close(m.A_in.data.values().cpu(), a_ref.values(), rtol=1e-4, atol=1e-6)
    batch = [torch.from_numpy(x) for x in make_batch(n, 16, 256, seed=4)]
    loss = m(*[b.to(gpu_device) for b in batch], device=gpu_device, mode="pre_training")
    loss.backward()
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in params.items()}
    want = O.pre_training_loss(p, cfg, a_ref, *batch, by_relation=True)
    want.backward()
    np.testing.assert_allclose(float(loss), float(want), rtol=1e-4)
    assert m.gat_embed.shape == (n, 4 * d)
    for k, v in m.named_parameters():
        if v.grad is not None and k != "A_in":
            ref = p[k].grad
            err = float((v.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
            assert err < 2e-3, (k, err)
