"""CPU restatement of the LiteralKG hot path (aggregation, attention refresh,
literal gate, triple scoring) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Status of the pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so this restatement is pinned against outputs of the
reference itself: ``oracle/gen_golden.py`` imports ``/root/reference/model.py``
on CPU and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
replays those fixtures through the functions below.

Shape of the code: the reference keeps its math inside ``nn.Module`` classes.
This file is deliberately *functional*: every routine takes a flat ``params``
mapping whose keys are the reference's ``state_dict`` names (so one set of
weights drives the reference, this oracle and the HIP module alike) and a
``cfg`` object carrying the reference's ``args`` field names.  Only ATen ops the
reference itself dispatches are used on the timed paths (sparse ``matmul``,
``torch.sparse.softmax``, ``linear``, ``layer_norm`` ...), so that timing this
file on the host cores is a fair stand-in for timing the reference there.

Every function cites the reference lines (``/root/reference/<file>:<lines>``)
whose behaviour it restates.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Mapping[str, torch.Tensor]

LEAKY_SLOPE = 0.01      # nn.LeakyReLU() default, model.py:29
LN_EPS = 1e-5           # nn.LayerNorm default, model.py:30
NORMALIZE_EPS = 1e-12   # F.normalize default, model.py:305


def default_cfg(**over) -> SimpleNamespace:
    """Field names follow argument_pretraining.py:3-137 (the subset model.py:169-263 reads)."""
    cfg = SimpleNamespace(
        use_pretrain=0, device="cpu",
        embed_dim=16, relation_dim=16, scale_gat_dim=None,
        use_residual=False, alpha=0.1, lamda=0.5,
        aggregation_type="gcn", n_conv_layers=1, conv_dim=16,
        mess_dropout=0.0, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
        pre_training_neg_rate=3, fine_tuning_neg_rate=3,
        num_lit_dim=2, txt_lit_dim=300, use_num_lit=False, use_txt_lit=False,
        milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64,
    )
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


# --------------------------------------------------------------------------
# a10  _L2_loss_mean                                           model.py:8-9
# --------------------------------------------------------------------------
def l2_loss_mean(x: torch.Tensor) -> torch.Tensor:
    return (x.square().sum(dim=1) * 0.5).mean()


# --------------------------------------------------------------------------
# a4  per-edge attention logit                              model.py:430-442
# --------------------------------------------------------------------------
def edge_logits(ent: torch.Tensor, rel: torch.Tensor, h: torch.Tensor,
                t: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """v_e = sum_d ent[t_e,d] * tanh(ent[h_e,d] + rel[r_e,d]) for every edge.

    The reference evaluates this relation by relation (model.py:451-460); the
    per-edge value does not depend on that grouping, so it is computed for
    arbitrary per-edge ``r`` here.
    """
    return (ent[t] * torch.tanh(ent[h] + rel[r])).sum(dim=1)


# --------------------------------------------------------------------------
# a5  attention refresh                                     model.py:444-471
# --------------------------------------------------------------------------
def attention_refresh(n_entities: int, ent: torch.Tensor, rel: torch.Tensor,
                      h: torch.Tensor, t: torch.Tensor, r: torch.Tensor,
                      relations: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Timed form: same ATen ops as the reference (per-relation loop, COO,
    ``torch.sparse.softmax(dim=1)`` which coalesces -- i.e. SUMS duplicate (h,t)
    logits -- first).  Returns a coalesced sparse COO N x N tensor."""
    if relations is None:
        relations = sorted(set(r.tolist()))
    hs, ts, vs = [], [], []
    for rid in relations:
        sel = (r == rid).nonzero(as_tuple=True)[0]
        hh, tt = h[sel], t[sel]
        hs.append(hh)
        ts.append(tt)
        vs.append((ent[tt] * torch.tanh(ent[hh] + rel[rid])).sum(dim=1))
    idx = torch.stack([torch.cat(hs), torch.cat(ts)])
    a = torch.sparse_coo_tensor(idx, torch.cat(vs), (n_entities, n_entities))
    return torch.sparse.softmax(a.cpu(), dim=1)


def attention_refresh_explicit(n_entities: int, ent: torch.Tensor, rel: torch.Tensor,
                               h: torch.Tensor, t: torch.Tensor, r: torch.Tensor
                               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Independent restatement of the same semantics without torch.sparse:
    sort by (h,t), sum duplicate pairs in logit space, softmax per head row.
    Returns (rows int64, cols int64, values fp32) sorted by (row, col)."""
    v = edge_logits(ent, rel, h, t, r).double()
    key = h.to(torch.int64) * n_entities + t.to(torch.int64)
    ukey, inv = torch.unique(key, sorted=True, return_inverse=True)
    merged = torch.zeros(ukey.numel(), dtype=torch.float64).index_add_(0, inv, v)
    rows = torch.div(ukey, n_entities, rounding_mode="floor")
    cols = ukey - rows * n_entities
    rmax = torch.full((n_entities,), -float("inf"), dtype=torch.float64)
    rmax = rmax.scatter_reduce(0, rows, merged, reduce="amax")
    ex = torch.exp(merged - rmax[rows])
    den = torch.zeros(n_entities, dtype=torch.float64).index_add_(0, rows, ex)
    return rows, cols, (ex / den[rows]).float()


# --------------------------------------------------------------------------
# a1  neighbour aggregation (SpMM)                              model.py:106
# --------------------------------------------------------------------------
def aggregate(a_in: torch.Tensor, ego: torch.Tensor) -> torch.Tensor:
    """side[h] = sum_t A[h,t] * ego[t]; rows are heads, columns are tails."""
    return torch.matmul(a_in, ego)


# --------------------------------------------------------------------------
# a6  literal gates                                    gate.py:22-28, 45-51
# --------------------------------------------------------------------------
def gate_mul(p: Params, prefix: str, x: torch.Tensor, num: torch.Tensor,
             txt: torch.Tensor) -> torch.Tensor:
    stacked = torch.cat([x, num, txt], dim=1)
    g = torch.tanh(F.linear(stacked, p[prefix + "g.weight"], p[prefix + "g.bias"]))
    z = torch.sigmoid(F.linear(x, p[prefix + "gate_ent.weight"])
                      + F.linear(num, p[prefix + "gate_num_lit.weight"])
                      + F.linear(txt, p[prefix + "gate_txt_lit.weight"])
                      + p[prefix + "gate_bias"])
    return (1 - z) * x + z * g


def gate_single(p: Params, prefix: str, x: torch.Tensor, lit: torch.Tensor) -> torch.Tensor:
    stacked = torch.cat([x, lit], dim=1)
    g = torch.tanh(F.linear(stacked, p[prefix + "g.weight"], p[prefix + "g.bias"]))
    z = torch.sigmoid(F.linear(x, p[prefix + "gate_ent.weight"])
                      + F.linear(lit, p[prefix + "gate_lit.weight"])
                      + p[prefix + "gate_bias"])
    return (1 - z) * x + z * g


def gate_embeddings(p: Params, cfg, num: Optional[torch.Tensor],
                    txt: Optional[torch.Tensor]) -> torch.Tensor:
    """model.py:265-279 -- which gate runs depends on the two use_* flags."""
    x = p["entity_embed.weight"]
    if cfg.use_num_lit and cfg.use_txt_lit:
        return gate_mul(p, "emb_mul_lit.", x, num, txt)
    if cfg.use_num_lit:
        return gate_single(p, "emb_num_lit.", x, num)
    if cfg.use_txt_lit:
        return gate_single(p, "emb_txt_lit.", x, txt)
    return x


# --------------------------------------------------------------------------
# a3  GCNII-style residual                                   model.py:90-99
# --------------------------------------------------------------------------
def residual_mix(p: Params, lp: str, hi: torch.Tensor, h0: torch.Tensor, cfg, layer_no: int,
                 use_residual: bool) -> torch.Tensor:
    if not use_residual:
        return hi
    h0p = F.linear(h0, p[lp + "linear_h0.weight"], p[lp + "linear_h0.bias"])
    mixed = (1 - cfg.alpha) * hi + cfg.alpha * h0p
    beta = math.log(cfg.lamda / layer_no + 1)
    # (1 - beta) is added to EVERY entry of weight (model.py:96), not to its diagonal
    return mixed @ ((1 - beta) + beta * p[lp + "weight"])


def _lin(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def _act(x: torch.Tensor) -> torch.Tensor:
    return F.leaky_relu(x, LEAKY_SLOPE)


def _relu(x: torch.Tensor) -> torch.Tensor:
    """the MLP head's activation (model.py:508-515); a function of its own so that tests can watch its inputs"""
    return torch.relu(x)


def _ln(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[1],), p[name + ".weight"], p[name + ".bias"], LN_EPS)


# --------------------------------------------------------------------------
# a1 + a2  one aggregation layer                           model.py:101-164
# --------------------------------------------------------------------------
def aggregator_layer(p: Params, lp: str, cfg, ego: torch.Tensor, a_in: torch.Tensor,
                     earlier: List[torch.Tensor], layer_no: int,
                     dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    kind = cfg.aggregation_type
    res = bool(cfg.use_residual)
    side = aggregate(a_in, ego)
    h0 = earlier[0]
    if kind == "gcn":
        out = _act(_lin(p, lp + "linear", residual_mix(p, lp, ego + side, h0, cfg, layer_no, res)))
    elif kind == "graphsage":
        both = torch.cat([ego, side], dim=1)
        if res:
            both = residual_mix(p, lp, _lin(p, lp + "linear_h", both), h0, cfg, layer_no, True)
        out = _act(_lin(p, lp + "linear", both))
    elif kind == "bi-interaction":
        s = _act(_lin(p, lp + "linear1", residual_mix(p, lp, ego + side, h0, cfg, layer_no, res)))
        b = _act(_lin(p, lp + "linear2", residual_mix(p, lp, ego * side, h0, cfg, layer_no, res)))
        out = b + s
    elif kind == "gin":
        # model.py:131-158.  The n_mlp_layers == 1 branch of the reference cannot run
        # (inp_linear / out_linear are never created, model.py:66-68) so it is not restated.
        if cfg.n_mlp_layers < 2:
            raise AttributeError("gin with n_mlp_layers < 2 fails in the reference (model.py:133)")
        stack = [_lin(p, lp + "inp_linear", ego)]
        hcur = _lin(p, lp + "inp_linear", ego + side)
        for i in range(cfg.n_mlp_layers - 1):
            hcur = _ln(p, f"{lp}mlp_layer_norms.{i}", _act(_lin(p, f"{lp}linears.{i}", hcur)))
            stack.append(hcur)
        x = torch.stack(stack).sum(dim=0)
        x = residual_mix(p, lp, x, h0, cfg, layer_no, res)
        out = _act(_lin(p, lp + "out_linear", x))
        if len(earlier) > 1:
            out = torch.stack([_ln(p, lp + "layer_normalize", out)] + list(earlier[1:])).sum(dim=0)
    else:
        raise NotImplementedError(kind)
    out = _ln(p, lp + "layer_normalize", out)
    return F.dropout(out, dropout_p, training)


# --------------------------------------------------------------------------
# a7  encoder driver                                       model.py:298-314
# --------------------------------------------------------------------------
def gat_embeddings(p: Params, cfg, a_in: torch.Tensor, num: Optional[torch.Tensor] = None,
                   txt: Optional[torch.Tensor] = None, training: bool = False) -> torch.Tensor:
    cur = gate_embeddings(p, cfg, num, txt)
    kept = [cur]
    for k in range(cfg.n_conv_layers):
        cur = aggregator_layer(p, f"aggregator_layers.{k}.", cfg, cur, a_in, kept, k + 1,
                               cfg.mess_dropout, training)
        kept.append(F.normalize(cur, p=2.0, dim=1, eps=NORMALIZE_EPS))
    cat = torch.cat(kept, dim=1)
    if cfg.scale_gat_dim is not None:
        return _act(_lin(p, "linear_gat", cat))
    return cat


# --------------------------------------------------------------------------
# a8  TransR-form triple loss                              model.py:364-428
# --------------------------------------------------------------------------
def triple_scores_transr(p: Params, gat: torch.Tensor, h, r, pos_t, neg_t):
    r_e = p["relation_embed.weight"][r]
    w_r = p["gat_trans_M"][r]
    ph = torch.bmm(gat[h].unsqueeze(1), w_r).squeeze(1)
    pp = torch.bmm(gat[pos_t].unsqueeze(1), w_r).squeeze(1)
    pn = torch.bmm(gat[neg_t].unsqueeze(1), w_r).squeeze(1)
    pos = (ph + r_e - pp).square().sum(dim=1)
    neg = (ph + r_e - pn).square().sum(dim=1)
    return pos, neg, (ph, r_e, pp, pn)


def triple_scores_transr_by_relation(p: Params, gat: torch.Tensor, h, r, pos_t, neg_t):
    """The same quantities for batches whose B x C x D gather of ``gat_trans_M[r]`` (model.py:372) does not fit in
    memory (C5: 32 768 x 2048 x 512 fp32 = 137 GB): rows are projected relation by relation,
    ``x @ gat_trans_M[k]`` for the rows with r == k -- row by row the product ``bmm`` forms.  Held to
    ``triple_scores_transr`` on the fixtures by tests/test_oracle_golden.py."""
    r_e = p["relation_embed.weight"][r]
    m = p["gat_trans_M"]
    ph, pp, pn = (torch.zeros(h.numel(), m.shape[2], dtype=gat.dtype) for _ in range(3))
    for k in torch.unique(r).tolist():
        sel = torch.nonzero(r == k, as_tuple=True)[0]
        ph = ph.index_add(0, sel, gat[h[sel]] @ m[k])
        pp = pp.index_add(0, sel, gat[pos_t[sel]] @ m[k])
        pn = pn.index_add(0, sel, gat[neg_t[sel]] @ m[k])
    pos = (ph + r_e - pp).square().sum(dim=1)
    neg = (ph + r_e - pn).square().sum(dim=1)
    return pos, neg, (ph, r_e, pp, pn)


def triple_loss_transr(p: Params, cfg, gat: torch.Tensor, h, r, pos_t, neg_t, by_relation: bool = False) -> torch.Tensor:
    scores = triple_scores_transr_by_relation if by_relation else triple_scores_transr
    pos, neg, (ph, r_e, pp, pn) = scores(p, gat, h, r, pos_t, neg_t)
    rank = (-F.logsigmoid(neg - pos)).mean()
    reg = l2_loss_mean(ph) + l2_loss_mean(r_e) + l2_loss_mean(pp) + l2_loss_mean(pn)
    return rank + cfg.kg_l2loss_lambda * reg


# --------------------------------------------------------------------------
# a9  TransE-form triple loss                          model_bce.py:329-368
# --------------------------------------------------------------------------
def triple_scores_transe(p: Params, gat: torch.Tensor, h, r, pos_t, neg_t):
    r_e = p["relation_embed.weight"][r]
    eh, ep, en = gat[h], gat[pos_t], gat[neg_t]
    pos = (eh + r_e - ep).square().sum(dim=1)
    neg = (eh + r_e - en).square().sum(dim=1)
    return pos, neg, (eh, r_e, ep, en)


def triple_loss_transe(p: Params, cfg, gat: torch.Tensor, h, r, pos_t, neg_t) -> torch.Tensor:
    pos, neg, (eh, r_e, ep, en) = triple_scores_transe(p, gat, h, r, pos_t, neg_t)
    rank = (-F.logsigmoid(neg - pos)).mean()
    reg = l2_loss_mean(eh) + l2_loss_mean(r_e) + l2_loss_mean(ep) + l2_loss_mean(en)
    return rank + cfg.kg_l2loss_lambda * reg


# --------------------------------------------------------------------------
# f1  link-prediction heads                       model.py:316-348, 473-491
# --------------------------------------------------------------------------
def prediction_loss(cfg, gat: torch.Tensor, head, pos_t, neg_t) -> torch.Tensor:
    eh, ep, en = gat[head], gat[pos_t], gat[neg_t]
    pos = (eh * ep).sum(dim=1)
    neg = (eh * en).sum(dim=1)
    rank = (-F.logsigmoid(pos - neg)).mean()
    reg = l2_loss_mean(eh) + l2_loss_mean(ep) + l2_loss_mean(en)
    return rank + cfg.fine_tuning_l2loss_lambda * reg


def link_scores(gat: torch.Tensor, head_ids, tail_ids) -> torch.Tensor:
    return gat[head_ids] @ gat[tail_ids].t()


def predict_links(cfg, gat: torch.Tensor, head_ids, tail_ids) -> torch.Tensor:
    s = link_scores(gat, head_ids, tail_ids)
    lo, hi = s.min(), s.max()
    return ((s - lo) / (hi - lo) > cfg.milestone_score).int()


# --------------------------------------------------------------------------
# whole modes, for end-to-end parity and for the timed CPU baseline
# --------------------------------------------------------------------------
def pre_training_loss(p: Params, cfg, a_in, h, r, pos_t, neg_t, num=None, txt=None,
                      form: str = "transr", training: bool = False, by_relation: bool = False) -> torch.Tensor:
    gat = gat_embeddings(p, cfg, a_in, num, txt, training)
    if form == "transr":
        return triple_loss_transr(p, cfg, gat, h, r, pos_t, neg_t, by_relation)
    return triple_loss_transe(p, cfg, gat, h, r, pos_t, neg_t)


def laplacian_a_in(n_entities: int, h: torch.Tensor, t: torch.Tensor, r: torch.Tensor,
                   kind: str = "random-walk") -> torch.Tensor:
    """Initial A_in = sum_r norm(A_r)  (dataloader.py:449-495), A_r binary per relation with
    duplicate (h,t) inside one relation already removed by the loader (dataloader.py:189).
    random-walk: D_r^-1 A_r;  symmetric: D_r^-1/2 A_r D_r^-1/2 (both with the ROW sums of A_r,
    exactly as the reference computes them)."""
    acc = None
    for rid in sorted(set(r.tolist())):
        sel = (r == rid).nonzero(as_tuple=True)[0]
        hh, tt = h[sel].long(), t[sel].long()
        deg = torch.zeros(n_entities, dtype=torch.float64).index_add_(
            0, hh, torch.ones(hh.numel(), dtype=torch.float64))
        if kind == "random-walk":
            w = 1.0 / deg[hh]
        elif kind == "symmetric":
            dis = deg.pow(-0.5)
            dis[torch.isinf(dis)] = 0
            w = dis[hh] * dis[tt]
        else:
            raise NotImplementedError(kind)
        m = torch.sparse_coo_tensor(torch.stack([hh, tt]), w, (n_entities, n_entities))
        acc = m if acc is None else acc + m
    acc = acc.coalesce()
    keep = acc.values() != 0          # scipy's diagonal products do not store the inf -> 0 entries
    return torch.sparse_coo_tensor(acc.indices()[:, keep], acc.values()[keep].float(), acc.shape).coalesce()


# --------------------------------------------------------------------------
# f1  MLP pair head                   model.py:499-519, model_bce.py:423-436
# --------------------------------------------------------------------------
def mlp_head(p: Params, gat: torch.Tensor, head_ids, tail_ids, training: bool = True) -> torch.Tensor:
    """sigmoid(fc3(bn2(relu(fc2(bn1(relu(fc1([e_h | e_t]))))))));  training=True uses batch statistics (the running
    buffers in ``p`` are updated in place like nn.BatchNorm1d does)."""
    x = torch.cat([gat[head_ids], gat[tail_ids]], dim=1)
    for fc, bn in (("fc1", "norm1"), ("fc2", "norm2")):
        x = _relu(F.linear(x, p[fc + ".weight"], p[fc + ".bias"]))
        x = F.batch_norm(x, p[bn + ".running_mean"], p[bn + ".running_var"], p[bn + ".weight"], p[bn + ".bias"],
                         training, 0.1, 1e-5)
    return torch.sigmoid(F.linear(x, p["fc3.weight"], p["fc3.bias"]))
