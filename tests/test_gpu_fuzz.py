"""Randomised configuration sweep: the HIP module against the oracle over drawn model / graph / batch shapes.
The hand-picked parity cases of test_gpu_parity.py pin the families; this sweep draws COMBINATIONS nobody picked (odd widths
behind gates, one relation, batches of one, row-sparse machinery on or off, pruning on or off) and checks, per case,
   the pre-training loss, the propagated table and every parameter gradient,
   the fine-tuning loss and its gradients, the MLP head in training mode (output and gradients),
   a training-mode step with message dropout: the row-sparse backward machinery against the dense backward (large graphs),
   the link scores (calc_score) and the attention refresh,
each against oracle/literalkg_oracle.py on the same seeded inputs (1e-4 of the largest entry on values, 2e-3 on gradients --
the tolerances of the hand-picked cases; where a drawn configuration is ill-conditioned in fp32 the oracle is also evaluated
in float64 and the HIP result has to be as close to THAT as the fp32 oracle is, see within_reference_noise).
LKG_FUZZ_CASES (default 16) sets how many cases are drawn and LKG_FUZZ_SEED the first seed, or LKG_FUZZ_SEEDS lists them
("1020,1067"); a failing case prints its seed and configuration, `LKG_FUZZ_SEEDS=<seed>` replays it alone and
LKG_FUZZ_OVERRIDE='{"prune": false}' replays it with some of the drawn choices replaced."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("LKG_FUZZ_CASES", "16"))
SEED0 = int(os.environ.get("LKG_FUZZ_SEED", "1000"))
SEEDS = [int(x) for x in os.environ.get("LKG_FUZZ_SEEDS", "").split(",") if x] or [SEED0 + i for i in range(N_CASES)]


@pytest.fixture(scope="module")
def L(gpu_device):
    import __graft_entry__ as ge
    ge.build()
    import literalkg_amd
    return literalkg_amd


@pytest.fixture(scope="module")
def O():
    from oracle import literalkg_oracle
    return literalkg_oracle


def draw(seed):
    """one configuration; every choice from the seed alone"""
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    agg = pick(["gcn", "gcn", "graphsage", "bi-interaction", "gin"])
    layers = pick([1, 1, 2, 2, 3])
    dim = pick([8, 12, 30, 32, 50, 64, 64, 100, 128, 256, 300])
    conv = pick([dim, dim, 16, 32, 20, 2 * dim if dim <= 64 else dim])
    residual = bool(rng.random() < 0.2)
    if residual and agg in ("gcn", "bi-interaction"):
        conv = dim                              # (the reference mixes h0 into ego + side: the widths have to agree, model.py:92-97)
    gate = pick([None, None, "num", "txt", "mul"])
    txt_dim = pick([300, 7, 64, 33])
    scale = pick([None, None, 48, 64, 100, dim])
    scoring = pick(["transr", "transr", "transe"])
    final = scale or dim + conv * layers
    rel_dim = dim if rng.random() < 0.7 else pick([16, 40, 64])
    if scoring == "transe":
        rel_dim = final
    n = int(pick([300, 2_000, 9_000, 20_000, 20_000, 35_000]))       # (>= 16 384 rows: the row-sparse backward is active)
    e = int(n * pick([1, 3, 8]))
    n_rel = int(pick([1, 3, 16, 16, 100]))
    batch = int(pick([1, 17, 200, 683]))
    neg = int(pick([1, 3]))
    c = dict(agg=agg, layers=layers, dim=dim, conv=conv, residual=residual, gate=gate, txt_dim=txt_dim, scale=scale,
             scoring=scoring, rel_dim=rel_dim, n=n, e=e, n_rel=n_rel, batch=batch, neg=neg,
             prune=bool(rng.random() < 0.3), mlp_hidden=int(pick([24, 48, 64])), skew=pick(["zipf", "uniform"]),
             weight_scale=float(pick([1.0, 10.0, 30.0])))
    # (drawn AFTER everything above, so that a seed keeps the configuration earlier rounds of the sweep recorded for it)
    if rng.random() < 0.25:
        c["e"] = max(1, int(n * 0.3))           # most entities are nobody's head
    c["cfg"] = dict(num_lit_dim=int(pick([2, 2, 1, 5])), n_mlp_layers=int(pick([2, 2, 3])), alpha=float(pick([0.1, 0.1, 0.5])),
                    lamda=float(pick([0.5, 0.5, 2.0])))
    c["batch_pool"] = int(pick([0, 0, 3, 40]))  # > 0: batch ids from the first few entities only (repeated ids, shared rows)
    return c


NEAR_KINK = 1e-5          # relative to the largest LeakyReLU input of the pass


class NearKink(AssertionError):
    """a gradient comparison failed in a pass that holds a LeakyReLU input within fp32 rounding reach of zero"""
REPORT = bool(os.environ.get("LKG_FUZZ_REPORT"))      # print every comparison's three distances instead of stopping at the first
# how every comparison of a run was settled (LKG_FUZZ_EXITS=<file>: the tally is written there as JSON at interpreter exit):
#   direct       within tolerance of the fp32 oracle
#   float64      not, but no further from the float64 oracle than 10 x the fp32 oracle is
#   association  not, but reproduced by the fp32 oracle in the device's association of the residual products
#   near_kink    not: a LeakyReLU input within rounding reach of zero -- the case is drawn again with other values
#   failed       none of the above
EXITS = {"direct": 0, "float64": 0, "association": 0, "near_kink": 0, "failed": 0}
if os.environ.get("LKG_FUZZ_EXITS"):
    import atexit

    def _dump_exits():
        path = os.environ["LKG_FUZZ_EXITS"]
        old = json.load(open(path)) if os.path.exists(path) else {}
        for k, v in EXITS.items():
            old[k] = old.get(k, 0) + v
        json.dump(old, open(path, "w"), indent=1)
    atexit.register(_dump_exits)


def within_reference_noise(got, want32, want64, tol, what, kink=None, alt32=None, floor=0.0):
    """`got` (HIP, fp32) against the fp32 oracle within `tol` of the largest entry.  Two properties of the reference's OWN fp32
    path keep a drawn configuration from meeting that, and both are decided against the oracle evaluated in float64:
      * ill-conditioning (the residual mix multiplies by a matrix of near-equal entries and LayerNorm then removes the common
        part: the fp32 oracle itself is percent-level off the float64 one) -- accepted when the HIP result is no further from
        float64 than 10 x the fp32 oracle is;
      * a LeakyReLU input within fp32 rounding reach of zero (`kink()`: the smallest |input| over the largest, from the
        float64 run): which slope such an element takes is decided by the summation order, and every gradient upstream of it
        differs by a finite step either way.  Two arbiters: `alt32()` -- the fp32 oracle evaluated in the DEVICE's association
        of the residual products (device_association below): the one reformulation of the device path that perturbs every
        pre-activation of a residual layer by a rounding; a deviation it reproduces to `tol` is that rounding falling on the
        other side of a kink, not a kernel's doing (seed 3130: four parameters off by 0.3 - 1.5 %, each reproduced to three
        digits) -- and, failing that, NearKink: the test draws the SAME configuration with other values, twice at most; a
        wrong kernel fails every time, a coin does not."""
    scale = max(float(want32.abs().max()), floor) + 1e-12    # (floor: a gradient that is zero next to the model's others --
    err = float((got - want32).abs().max()) / scale          # a saturated sigmoid in front of fc3 -- is compared on their scale)
    if REPORT:
        t = want64()
        mine, noise = float((got.double() - t).abs().max()) / scale, float((want32.double() - t).abs().max()) / scale
        print(f"  {'BAD' if mine > max(tol, 10 * noise) else 'ok '} {what}: hip-o32 {err:.2e}  hip-f64 {mine:.2e}  "
              f"o32-f64 {noise:.2e}  largest {scale:.3g}")
        return
    if err < tol:
        EXITS["direct"] += 1
        return
    truth = want64()
    noise = float((want32.double() - truth).abs().max()) / scale
    mine = float((got.double() - truth).abs().max()) / scale
    if mine <= max(tol, 10 * noise):
        EXITS["float64"] += 1
        return
    if alt32 is not None:
        other = alt32()
        if other is not None and float((got - other).abs().max()) / scale < tol:
            print(f"{what}: {err:.3g} from the fp32 oracle in the reference's association, within {tol:g} of it in the device's")
            EXITS["association"] += 1
            return
    at = lambda i: f"{np.unravel_index(int(i), tuple(got.shape))}: hip {got.flatten()[i]:.9g} oracle32 " \
                   f"{want32.flatten()[i]:.9g} oracle64 {truth.flatten()[i]:.12g}"
    near = kink is not None and kink() < NEAR_KINK
    EXITS["near_kink" if near else "failed"] += 1
    raise (NearKink if near else AssertionError)(
        f"{what}: hip vs fp32 oracle {err:.3g}, hip vs f64 {mine:.3g}, fp32 oracle vs f64 {noise:.3g} "
        f"(of the largest entry {scale:.3g}); worst hip element {at((got.double() - truth).abs().argmax())}; "
        f"worst oracle32 element {at((want32.double() - truth).abs().argmax())}; "
        f"smallest relative LeakyReLU input {kink() if kink else None}")


def close_grads(named, ref, ref64, tag, ref32_device=None, tol=2e-3):
    named = list(named)
    largest = max([float(ref[k].grad.abs().max()) for k, v in named if v.grad is not None and k != "A_in"
                   and ref[k].grad is not None] + [0.0])
    for k, v in named:
        if v.grad is None or k == "A_in":
            continue
        assert ref[k].grad is not None, (tag, k)
        within_reference_noise(v.grad.cpu(), ref[k].grad, lambda: ref64()[k].grad, tol, (tag, k),
                               kink=lambda: ref64()["smallest relative LeakyReLU input"],
                               alt32=None if ref32_device is None else (lambda: ref32_device()[k].grad), floor=1e-6 * largest)


class device_association:
    """The oracle with the residual layers' products associated as the device path associates them: the reference computes
    Linear((mixed @ W'), W) with W' = (1 - b) + b W_res (model.py:95-98), the device mixed @ (W W'^T)^T + bias -- the two small
    matrices first, one N-row product (model.Aggregator._lin_mapped).  Equal in exact arithmetic (1e-13 in float64)."""

    def __init__(self, O):
        self.O = O

    def __enter__(self):
        import math
        import torch.nn.functional as F
        O = self.O
        self.real = (O.residual_mix, O._lin)
        real_lin = O._lin

        def mix(p, lp, hi, h0, cfg, layer_no, use_residual):
            if not use_residual:
                return hi
            mixed = (1 - cfg.alpha) * hi + cfg.alpha * F.linear(h0, p[lp + "linear_h0.weight"], p[lp + "linear_h0.bias"])
            beta = math.log(cfg.lamda / layer_no + 1)
            return mixed, (1 - beta) + beta * p[lp + "weight"]

        def lin(p, name, x):
            if isinstance(x, tuple):      # (the fold of the two small matrices in float64, rounded once: ops.fold_nt)
                from literalkg_amd import ops as _ops
                w = p[name + ".weight"]
                fold = (w.double() @ x[1].double().t()).to(w.dtype) if _ops.FOLD_F64 else w @ x[1].t()
                return F.linear(x[0], fold, p[name + ".bias"])
            return real_lin(p, name, x)
        O.residual_mix, O._lin = mix, lin

    def __exit__(self, *exc):
        self.O.residual_mix, self.O._lin = self.real


@pytest.mark.parametrize("seed", SEEDS)
def test_drawn_configuration_matches_the_oracle(L, O, gpu_device, seed):
    c = draw(seed)
    over = json.loads(os.environ.get("LKG_FUZZ_OVERRIDE", "{}"))         # (replaying a case with some choices changed)
    c["cfg"].update(over.pop("cfg", {}))
    c.update(over)
    print("fuzz case", seed, c)
    for attempt in range(3):
        try:
            return run_case(L, O, gpu_device, c, seed, seed + 7919 * attempt)
        except NearKink as e:
            if attempt == 2:
                raise
            print("near a LeakyReLU kink:", e, "\n-> the same configuration with other values")


def run_case(L, O, gpu_device, c, seed, value_seed):
    """configuration c on the graph drawn from `seed`, parameters / literals / batch drawn from `value_seed`"""
    from literalkg_amd import io
    from literalkg_amd.synth import make_batch, make_kg
    n, n_rel = c["n"], c["n_rel"]
    h, t, r = make_kg(n, c["e"], c["skew"], seed=seed)
    r = np.random.default_rng(seed + 1).integers(0, n_rel, len(r))
    _, first = np.unique(np.stack([h, r, t], 1), axis=0, return_index=True)
    h, t, r = h[first], t[first], r[first]
    cfg = O.default_cfg(embed_dim=c["dim"], relation_dim=c["rel_dim"], conv_dim=c["conv"], n_conv_layers=c["layers"],
                        aggregation_type=c["agg"], scale_gat_dim=c["scale"], use_residual=c["residual"],
                        use_num_lit=c["gate"] in ("mul", "num"), use_txt_lit=c["gate"] in ("mul", "txt"),
                        txt_lit_dim=c["txt_dim"], mlp_hidden_dim=c["mlp_hidden"], kg_l2loss_lambda=1e-4,
                        fine_tuning_l2loss_lambda=1e-4, pre_training_neg_rate=c["neg"], fine_tuning_neg_rate=c["neg"],
                        device=gpu_device)
    for k, v in c.get("cfg", {}).items():       # (LKG_FUZZ_OVERRIDE='{"cfg": {"alpha": 0.0}}': configuration fields nobody draws)
        setattr(cfg, k, v)
    torch.manual_seed(value_seed)
    num = torch.rand(n, cfg.num_lit_dim) if cfg.use_num_lit else None
    txt = torch.randn(n, cfg.txt_lit_dim) if cfg.use_txt_lit else None
    a_in = io.initial_a_in(n, h, t, r)
    m = L.LiteralKG(cfg, n, n_rel, a_in, num, txt, scoring=c["scoring"])
    with torch.no_grad():
        m.entity_embed.weight.mul_(c["weight_scale"])
        m.relation_embed.weight.mul_(min(c["weight_scale"], 3.0))
    params = {k: v.detach().clone() for k, v in m.state_dict().items() if k != "A_in"}
    m.to(gpu_device).eval()
    m.prune_to_batch = c["prune"]
    if c.get("dense_backward"):                 # (override only: the row-sparse backward machinery switched off)
        m._table_grad_stays_inside = lambda: False
    pool = min(n, c.get("batch_pool", 0)) or n
    bh, br, bp, bn = (torch.from_numpy(x) for x in make_batch(pool, c["batch"], c["neg"], seed=value_seed + 2))
    br = torch.from_numpy(np.repeat(np.random.default_rng(value_seed + 3).integers(0, n_rel, c["batch"]), c["neg"]))
    dev = lambda *xs: [x.to(gpu_device) for x in xs]

    def in_f64(loss_of, params=params):
        """the oracle in float64 on the same inputs (evaluated once, only when a comparison asks for it)"""
        memo = {}

        def run():
            if not memo:
                dd = lambda x: None if x is None else x.double()
                p64 = {k: (v.double() if v.is_floating_point() else v).clone().requires_grad_(
                    v.is_floating_point() and "running_" not in k) for k, v in params.items()}
                seen, act, relu = [], O._act, O._relu

                def watch(fn):
                    def spy(x):
                        seen.append(float(x.detach().abs().min() / (x.detach().abs().max() + 1e-300)))
                        return fn(x)
                    return spy
                O._act, O._relu = watch(act), watch(relu)      # (LeakyReLU of the layers, ReLU of the MLP head: both kinked at 0)
                try:
                    loss_of(p64, a_in.double(), dd(num), dd(txt)).backward()
                finally:
                    O._act, O._relu = act, relu
                memo.update(p64)
                memo["smallest relative LeakyReLU input"] = min(seen) if seen else 1.0      # (ReLU inputs of the MLP head included)
            return memo
        return run

    def in_f32_device(loss_of, params=params):
        """the fp32 oracle in the device's association of the residual products (None without residual layers)"""
        memo = {}

        def run():
            if not memo:
                q = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in params.items()}
                with device_association(O):
                    loss_of(q, a_in, num, txt).backward()
                memo.update(q)
            return memo
        return run if c["residual"] else None

    # Three or more residual layers: every layer multiplies by the near-constant identity-mapping matrix and LayerNorm removes the
    # common part again -- rounding is amplified layer by layer and a few of the ~10^7 LeakyReLU inputs flip on EVERY draw of the
    # values (1 % of the sweep's cases, all of this family, missed 2e-3 three draws in a row at 2e-3 .. 7e-3).  Their gradients are
    # held to 2e-2 (a wrong kernel is off by O(1)); everything else of the case keeps its tolerance.
    grad_tol = 2e-2 if (c["residual"] and c["layers"] >= 3) else 2e-3

    # ---- pre-training: loss, table, gradients
    loss = m(*dev(bh, br, bp, bn), device=gpu_device, mode="pre_training")
    loss.backward()
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in params.items()}
    want = O.pre_training_loss(p, cfg, a_in, bh, br, bp, bn, num=num, txt=txt, form=c["scoring"])
    want.backward()
    np.testing.assert_allclose(float(loss), float(want), rtol=1e-4, err_msg=f"pre-training loss, case {seed}")
    pre = lambda q, a, nu, tx: O.pre_training_loss(q, cfg, a, bh, br, bp, bn, num=nu, txt=tx, form=c["scoring"])
    close_grads(m.named_parameters(), p, in_f64(pre), f"pre-training, case {seed}", in_f32_device(pre), grad_tol)
    gat = O.gat_embeddings(params, cfg, a_in, num, txt)
    dd = lambda x: None if x is None else x.double()
    gat64 = lambda: O.gat_embeddings({k: dd(v) if v.is_floating_point() else v for k, v in params.items()}, cfg,
                                     a_in.double(), dd(num), dd(txt))
    if not c["prune"]:                          # (a pruned forward holds the batch's rows only)
        within_reference_noise(m.gat_embed.detach().cpu(), gat, gat64, 1e-4, f"propagated table, case {seed}")

    # ---- fine-tuning head: loss and gradients
    m.zero_grad(set_to_none=True)
    loss = m(*dev(bh, bp, bn), device=gpu_device, mode="fine_tuning")
    loss.backward()
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in params.items()}
    want = O.prediction_loss(cfg, O.gat_embeddings(p, cfg, a_in, num, txt), bh, bp, bn)
    want.backward()
    np.testing.assert_allclose(float(loss), float(want), rtol=1e-4, err_msg=f"fine-tuning loss, case {seed}")
    fine = lambda q, a, nu, tx: O.prediction_loss(cfg, O.gat_embeddings(q, cfg, a, nu, tx), bh, bp, bn)
    close_grads(m.named_parameters(), p, in_f64(fine), f"fine-tuning, case {seed}", in_f32_device(fine), grad_tol)

    # ---- link scores (calc_score is what `predict` thresholds: the 0/1 cut itself flips on a 1e-7 difference)
    heads, tails = bh[:: max(1, len(bh) // 50)][:50], bp[:: max(1, len(bp) // 70)][:70]
    with torch.no_grad():
        got = m.calc_score(*dev(heads, tails))
    within_reference_noise(got.cpu(), O.link_scores(gat, heads, tails), lambda: O.link_scores(gat64(), heads, tails), 1e-4,
                           f"link scores, case {seed}")

    # ---- the MLP head (mode 'mlp', model.py:493-519 / model_bce.py:423-436) in training mode: batch statistics, every gradient
    if len(bh) >= 2 and c["scale"] is not None:  # (the head is 2 * scale_gat_dim wide, model.py:499; BatchNorm needs two rows)
        if not hasattr(m, "fc1"):                # (model.py builds the head on demand, model_bce.py in its constructor)
            torch.manual_seed(value_seed + 5)
            m.initialize_MLP()
            m.to(gpu_device)
        params_mlp = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if k != "A_in"}
        m.train()
        m.zero_grad(set_to_none=True)
        out = m(*dev(bh, bp), device=gpu_device, mode="mlp")
        weights = torch.linspace(-1.0, 1.0, len(bh)).reshape(-1, 1)
        (out * weights.to(gpu_device)).sum().backward()
        m.eval()

        def mlp_loss(q, a, nu, tx):
            q = dict(q)                          # (the running statistics are updated in place: every evaluation from a copy)
            for k in list(q):
                if "running_" in k or "num_batches" in k:
                    q[k] = q[k].clone()
            return (O.mlp_head(q, O.gat_embeddings(q, cfg, a, nu, tx), bh, bp, training=True) * weights.to(a.dtype)).sum()
        p = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in params_mlp.items()}
        q = dict(p)
        for k in list(q):
            if "running_" in k or "num_batches" in k:
                q[k] = q[k].clone()
        want = O.mlp_head(q, O.gat_embeddings(q, cfg, a_in, num, txt), bh, bp, training=True)
        (want * weights).sum().backward()
        def out64():
            q64 = {k: (v.double() if v.is_floating_point() else v).clone() for k, v in params_mlp.items()}
            dd_ = lambda x: None if x is None else x.double()
            return O.mlp_head(q64, O.gat_embeddings(q64, cfg, a_in.double(), dd_(num), dd_(txt)), bh, bp, training=True)
        within_reference_noise(out.detach().cpu(), want.detach(), out64, 1e-4, f"mlp head output, case {seed}")
        close_grads(m.named_parameters(), p, in_f64(mlp_loss, params_mlp), f"mlp head, case {seed}",
                    in_f32_device(mlp_loss, params_mlp), grad_tol)

    # ---- training mode with message dropout: the row-sparse backward machinery against the dense backward of the same model
    # (no oracle: the masks come from this package's generator; from 16 384 rows on the machinery is active)
    if n >= 16384 and not c["prune"]:
        from literalkg_amd import ops as _ops
        for layer in m.aggregator_layers:
            layer.dropout = 0.2
        m.train()

        def step(sparse):
            m.zero_grad(set_to_none=True)
            m._table_grad_stays_inside = (lambda: m.gat_rows is None) if sparse else (lambda: False)
            torch.manual_seed(value_seed + 11)           # (the dropout seeds follow torch's CPU generator)
            loss_ = m(*dev(bh, br, bp, bn), device=gpu_device, mode="pre_training")
            loss_.backward()
            return float(loss_.detach()), {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None and k != "A_in"}
        try:
            (l_s, g_s), (l_d, g_d) = step(True), step(False)
        finally:
            del m._table_grad_stays_inside
            _ops._RowScratch._tables.clear()
            for layer in m.aggregator_layers:
                layer.dropout = 0.0
            m.eval()
        assert l_s == l_d and np.isfinite(l_s), (f"training-mode loss, case {seed}", l_s, l_d)
        assert g_s.keys() == g_d.keys()
        top = max(float(v.abs().max()) for v in g_d.values())
        for k in g_d:                            # (float atomics in another order: 2e-4 of the parameter's, 1e-6 of the model's largest entry)
            scale_k = float(g_d[k].abs().max()) + 1e-30
            assert float((g_s[k] - g_d[k]).abs().max()) <= 2e-4 * scale_k + 1e-6 * top, (f"row-sparse against dense backward, case {seed}", k)

    # ---- attention refresh (the reference cannot add embeddings of different widths either, model.py:441)
    hd, td, rd = dev(*(torch.from_numpy(x) for x in (h, t, r)))
    if cfg.relation_dim != cfg.embed_dim:
        with pytest.raises(ValueError, match="embed_dim must equal relation_dim"):
            m(hd, td, rd, list(range(n_rel)), device=gpu_device, mode="update_att")
        return
    m(hd, td, rd, list(range(n_rel)), device=gpu_device, mode="update_att")
    ref_a = O.attention_refresh(n, params["entity_embed.weight"], params["relation_embed.weight"],
                                torch.from_numpy(h), torch.from_numpy(t), torch.from_numpy(r)).coalesce()
    got_a = m.A_in.data.cpu()
    assert torch.equal(got_a.indices(), ref_a.indices())
    torch.testing.assert_close(got_a.values(), ref_a.values(), rtol=1e-4, atol=1e-6)
    # ... and over a SUBSET of the relations: the reference visits the listed relations only (model.py:451), the triples of the
    # others are not part of the refreshed matrix; then the forward runs on it
    if n_rel > 1:
        sub_rng = np.random.default_rng(value_seed + 23)
        listed = sorted(sub_rng.choice(n_rel, max(1, n_rel // 2), replace=False).tolist())
        keep = np.isin(r, listed)
        m(hd, td, rd, listed, device=gpu_device, mode="update_att")
        got_b = m.A_in.data.cpu().coalesce()
        if keep.any():
            ref_b = O.attention_refresh(n, params["entity_embed.weight"], params["relation_embed.weight"],
                                        torch.from_numpy(h[keep]), torch.from_numpy(t[keep]), torch.from_numpy(r[keep])).coalesce()
            assert torch.equal(got_b.indices(), ref_b.indices()), f"refresh over relations {listed}, case {seed}"
            torch.testing.assert_close(got_b.values(), ref_b.values(), rtol=1e-4, atol=1e-6)
            with torch.no_grad():
                got_s = m.calc_score(*dev(bh[:20], bp[:30]))
            gat_b = O.gat_embeddings(params, cfg, ref_b, num, txt)
            within_reference_noise(got_s.cpu(), O.link_scores(gat_b, bh[:20], bp[:30]),
                                   lambda: O.link_scores(O.gat_embeddings({k: v.double() if v.is_floating_point() else v
                                                                           for k, v in params.items()}, cfg, ref_b.double(),
                                                                          None if num is None else num.double(),
                                                                          None if txt is None else txt.double()), bh[:20], bp[:30]),
                                   1e-4, f"link scores on the subset's matrix, case {seed}")
        else:
            assert got_b._nnz() == 0


# ============================================================================= the fused SpMM entry point, option by option
SPMM_CASES = int(os.environ.get("LKG_FUZZ_SPMM_CASES", "150"))


def draw_csr(rng, n_rows, n_cols, mean_deg, share_empty, n_long, gpu_device):
    """a CSR with empty rows, a few rows beyond the long-row threshold and unsorted column ids"""
    deg = rng.poisson(mean_deg, n_rows)
    deg[rng.random(n_rows) < share_empty] = 0
    for i in rng.choice(n_rows, min(n_long, n_rows), replace=False):
        deg[i] = int(rng.integers(257, 700))
    rowptr = np.zeros(n_rows + 1, np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n_cols, int(rowptr[-1]))
    to = lambda a, dt: torch.from_numpy(a).to(dt).to(gpu_device)
    return to(rowptr, torch.int32), to(col, torch.int32), torch.from_numpy(rng.random(int(rowptr[-1])).astype(np.float32)).to(gpu_device)


@pytest.mark.parametrize("seed", [7000 + i for i in range(SPMM_CASES)])
def test_fused_spmm_options_against_a_float64_product(gpu_device, seed):
    """lkg_spmm_csr_fused_f32 under drawn combinations of its options (self addend, second addend or bias row, row copy, row
    maxima, flagged operand rows with garbage outside the flags, the rows-written flags, row lists, long-row list, a row offset
    into x, strided operands, widths on the 16-byte and on the scalar path) against the same sums in float64.  Rows the call
    must not write keep their sentinel; operands declared zero outside their flags hold NaN there."""
    import __graft_entry__ as ge
    ge.build()
    from literalkg_amd import ops
    from literalkg_amd.graph import LONG_ROW_THRESHOLD
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    n_rows = int(pick([1, 5, 64, 300, 4097, 9000]))
    d = int(pick([1, 3, 4, 8, 20, 30, 32, 64, 100, 128, 132, 256, 300, 512, 520]))
    offset = int(pick([0, 0, 7, 1000]))
    n_x = int(pick([n_rows, 50, 5000]))                    # rows of x handed over (col - offset indexes them)
    rowptr, col, val = draw_csr(rng, n_rows, n_x, pick([0.5, 3, 12]), pick([0.0, 0.3, 0.8]), pick([0, 0, 2]), gpu_device)
    col = col + offset
    wide = lambda r, c: torch.randn(r, c + 8, device=gpu_device)[:, 4:4 + c] if rng.random() < 0.3 else torch.randn(r, c, device=gpu_device)
    x = wide(n_x, d)
    vec = d % 4 == 0 and x.data_ptr() % 16 == 0 and x.stride(0) % 4 == 0
    opt = dict(add_self=rng.random() < 0.4, second=pick([None, None, "add2", "bias"]), copy=rng.random() < 0.2,
               rowmax=rng.random() < 0.3, x_rows=rng.random() < 0.35, lists=rng.random() < 0.3, long=rng.random() < 0.7)
    opt["out_rows"] = bool(opt["x_rows"] and vec and d <= 1024 and rng.random() < 0.6)
    if opt["out_rows"]:
        opt["copy"] = opt["rowmax"] = False                # (the rows-written form excludes both, lkg_spmm.hip)
    if opt["add_self"] and n_x != n_rows:
        opt["add_self"] = False
    # (operands declared zero outside their flags hold NaN there where the kernel promises not to read them: on the 16-byte
    # path; the scalar path ignores the flags -- include/literalkg_hip.h -- and reads the zeros)
    kw, nan = {}, float("nan") if vec else 0.0
    x64 = x.double()
    if opt["x_rows"]:
        flags = (torch.rand(n_x, device=gpu_device) < pick([0.05, 0.5])).to(torch.uint8)
        x = x.clone() if x.is_contiguous() else x
        x[flags == 0] = nan
        x64 = torch.where(flags.bool()[:, None], x.double(), torch.zeros((), dtype=torch.float64, device=gpu_device))
        kw["x_rows"] = flags
    want = torch.zeros(n_rows, d, dtype=torch.float64, device=gpu_device)
    rows = torch.repeat_interleave(torch.arange(n_rows, device=gpu_device), (rowptr[1:] - rowptr[:-1]).long())
    want.index_add_(0, rows, val.double()[:, None] * x64[(col - offset).long()])
    touched = torch.zeros(n_rows, dtype=torch.bool, device=gpu_device)
    if opt["x_rows"]:
        touched[rows[kw["x_rows"][(col - offset).long()].bool()]] = True
    if opt["add_self"]:
        s = wide(n_rows, d)
        if opt["x_rows"] and rng.random() < 0.7:
            sf = (torch.rand(n_rows, device=gpu_device) < 0.3).to(torch.uint8)
            s = s.clone() if s.is_contiguous() else s
            s[sf == 0] = nan
            kw["self_rows"] = sf
            want += torch.where(sf.bool()[:, None], s.double(), torch.zeros((), dtype=torch.float64, device=gpu_device))
            touched |= sf.bool()
        else:
            want += s.double()
            touched[:] = True
        kw["add_self"] = s
    if opt["second"] == "add2" and not opt["out_rows"]:
        a2 = wide(n_rows, d)
        if rng.random() < 0.4:
            af = (torch.rand(n_rows, device=gpu_device) < 0.5).to(torch.uint8)
            a2 = a2.clone() if a2.is_contiguous() else a2
            a2[af == 0] = nan
            kw["add2_rows"] = af
            want += torch.where(af.bool()[:, None], a2.double(), torch.zeros((), dtype=torch.float64, device=gpu_device))
        else:
            want += a2.double()
        kw["add2"] = a2
    elif opt["second"] == "bias" and not opt["out_rows"]:
        kw["bias"] = torch.randn(d, device=gpu_device)
        want += kw["bias"].double()
    cdst = None
    if opt["copy"]:
        csrc, cdst = wide(n_rows, d), torch.full((n_rows, d), 5.0, device=gpu_device)
        kw["copy"] = (csrc, cdst)
    rm = torch.full((n_rows,), -3.0, device=gpu_device) if opt["rowmax"] else None
    deg = (rowptr[1:] - rowptr[:-1])
    if opt["long"]:
        lr = torch.nonzero(deg > LONG_ROW_THRESHOLD).flatten().int()
        kw["long_rows"] = lr if lr.numel() else None
    if opt["lists"] and not opt["x_rows"]:
        kw["row_lists"] = (torch.nonzero(deg > 0).flatten().int(), torch.nonzero(deg == 0).flatten().int())
    flagged = torch.zeros(n_rows, dtype=torch.uint8, device=gpu_device) if opt["out_rows"] else None
    out = torch.full((n_rows, d + 4), 9.0, device=gpu_device)[:, :d] if rng.random() < 0.3 else torch.full((n_rows, d), 9.0, device=gpu_device)
    got = ops.spmm_raw(rowptr, col, val, x, n_rows, out=out, x_row_offset=offset, rowmax=rm, out_rows=flagged, **kw)
    torch.cuda.synchronize()
    what = (seed, n_rows, d, offset, {k: v for k, v in opt.items() if v})
    scale = float(want.abs().max()) + 1e-30
    if opt["out_rows"]:
        f = flagged.bool()
        assert bool((f | ~touched).all()), ("a row with a contribution is not flagged", what)
        assert bool((got[~f] == 9.0).all()), ("an unflagged row was written", what)
        assert float((got[f].double() - want[f]).abs().max() if f.any() else 0.0) <= 2e-5 * scale, what
    else:
        assert float((got.double() - want).abs().max()) <= 2e-5 * scale, what
    if cdst is not None:
        assert torch.equal(cdst, kw["copy"][0]), what
    if rm is not None:
        torch.testing.assert_close(rm.double(), want.abs().amax(1), rtol=1e-5, atol=2e-5 * scale)


# ============================================================================= the attention refresh, shape by shape
ATT_CASES = int(os.environ.get("LKG_FUZZ_ATT_CASES", "80"))


@pytest.mark.parametrize("seed", [9000 + i for i in range(ATT_CASES)])
def test_attention_refresh_against_the_oracle_over_drawn_shapes(L, O, gpu_device, seed):
    """lkg_edge_softmax_f32 over drawn graphs (empty rows, rows of 1 / <= 64 / > 64 / > 256 entries, (h, t) pairs under two and
    three relations or none at all), widths on the 16-byte and the scalar path, relation tables that fit the LDS stage and
    tables that do not, embeddings inside and outside the tanh series' range, whole-graph and row-range refreshes: coalesced
    indices bit for bit, values to 1e-4 against the oracle's explicit form (logit, merge, row softmax)."""
    from literalkg_amd import ops
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    n = int(pick([40, 500, 3000]))
    d = int(pick([4, 8, 30, 64, 100, 128, 256, 300, 512, 1024]))
    n_rel = int(pick([1, 2, 6, 16, 40, 130]))
    e = int(n * pick([1, 4, 15]))
    h = (n * rng.random(e) ** pick([1.0, 1.7, 3.0])).astype(np.int64)
    t, r = rng.integers(0, n, e), rng.integers(0, n_rel, e)
    for row, deg in [(int(rng.integers(n)), int(pick([65, 70, 200, 257, 333, 900]))) for _ in range(int(pick([0, 1, 3])))]:
        h = np.concatenate([h, np.full(deg, row)])
        t = np.concatenate([t, rng.choice(n, deg, replace=deg > n)])
        r = np.concatenate([r, rng.integers(0, n_rel, deg)])
    trip = np.stack([h, r, t], 1)
    if n_rel > 1 and rng.random() < 0.7:        # the same (h, t) under a second and a third relation
        k2, k3 = int(len(trip) * pick([0.01, 0.2])), int(len(trip) * 0.02)
        trip = np.concatenate([trip, np.stack([h[:k2], (r[:k2] + 1) % n_rel, t[:k2]], 1),
                               np.stack([h[:k3], (r[:k3] + 2) % n_rel, t[:k3]], 1)])
    trip = np.unique(trip, axis=0)
    trip = trip[(trip[:, 0] % 11) != 3]          # some head rows stay empty
    trip = trip[rng.permutation(len(trip))]
    if len(trip) == 0:
        trip = np.array([[0, 0, 0]])
    h, r, t = trip[:, 0].copy(), trip[:, 1].copy(), trip[:, 2].copy()
    # |h + r| below 0.25 everywhere: the tanh series alone; larger: the exp / rcp form -- in a few columns only at the larger
    # widths, so that the logits stay O(1) (the oracle forms them in fp32 like the reference: at |logit| ~ 50 its own rounding
    # would exceed the 1e-4 the values are held to)
    mag = pick([0.02, 0.3, 1.5])
    ent, rel = torch.randn(n, d) * mag, torch.randn(n_rel, d) * mag
    if mag * mag * d > 8:
        keep = torch.zeros(d)
        keep[rng.choice(d, max(1, int(8 / (mag * mag))), replace=False)] = 1.0
        ent, rel = ent * (keep + (1 - keep) * 0.02 / mag), rel * (keep + (1 - keep) * 0.02 / mag)
    g = L.KGStructure.from_triples(n, h, t, r, device=gpu_device)
    what = (seed, n, d, n_rel, len(trip), mag, g.has_dups)
    val, logits = ops.edge_softmax(g, ent.to(gpu_device), rel.to(gpu_device), want_logits=bool(rng.random() < 0.3))
    rows, cols, want = O.attention_refresh_explicit(n, ent, rel, *(torch.from_numpy(a) for a in (h, t, r)))
    assert torch.equal(g.coo_indices().cpu(), torch.stack([rows, cols])), what
    torch.testing.assert_close(val.cpu(), want, rtol=1e-4, atol=1e-6, msg=lambda m: f"{what}: {m}")
    if logits is not None:                       # the merged pre-softmax logits: the row softmax of them is the value array
        lg = logits.cpu().double()
        rp = g.host("rowptr")
        for row in rng.choice(n, min(n, 20), replace=False):
            a, b = int(rp[row]), int(rp[row + 1])
            if b > a:
                torch.testing.assert_close(torch.softmax(lg[a:b], 0).float(), want[a:b], rtol=1e-4, atol=1e-6)
    # a row range into an existing array: the other rows' entries keep what they held
    lo = int(rng.integers(0, n))
    hi = int(rng.integers(lo, n + 1))
    out = torch.full((g.nnz,), -7.0, device=gpu_device)
    ops.edge_softmax(g, ent.to(gpu_device), rel.to(gpu_device), row_lo=lo, row_hi=hi, out=out)
    rp = g.host("rowptr")
    a, b = int(rp[lo]), int(rp[hi])
    torch.testing.assert_close(out[a:b].cpu(), want[a:b], rtol=1e-4, atol=1e-6, msg=lambda m: f"{what} rows [{lo}, {hi}): {m}")
    assert bool((out[:a] == -7.0).all()) and bool((out[b:] == -7.0).all()), what


# ============================================================================= the structure build: device against host
BUILD_CASES = int(os.environ.get("LKG_FUZZ_BUILD_CASES", "60"))


@pytest.mark.parametrize("seed", [11000 + i for i in range(BUILD_CASES)])
def test_device_structure_build_equals_the_host_build_on_drawn_edge_lists(L, gpu_device, seed):
    """lkg_csr_build_device / lkg_csr_transpose_device (radix sort + scans on the GPU) against lkg_csr_build / lkg_csr_transpose
    (host) on drawn edge lists -- sizes around the sort's digit and block boundaries, id spaces from 1 to 2^21, heavy heads,
    repeated (h, t) pairs under other relations and exact duplicate triples, no edges at all: every array bit for bit."""
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    n = int(pick([1, 2, 255, 256, 257, 4096, 65_535, 65_537, 300_000, 2_097_152]))
    e = int(pick([0, 1, 2, 255, 256, 257, 1023, 1025, 4097, 70_000, 250_001]))
    n_rel = int(pick([1, 2, 16, 255, 1000]))
    h = (n * rng.random(e) ** pick([1.0, 2.0, 6.0])).astype(np.int64)
    t, r = rng.integers(0, n, e), rng.integers(0, n_rel, e)
    if e > 4 and rng.random() < 0.6:
        k = int(rng.integers(1, e // 2 + 1))
        src = rng.integers(0, e, k)
        h, t = np.concatenate([h, h[src]]), np.concatenate([t, t[src]])
        r = np.concatenate([r, np.where(rng.random(k) < 0.5, r[src], (r[src] + 1) % n_rel)])     # exact repeats and other relations
    gh = L.KGStructure.from_triples(n, h, t, r, device="cpu")
    inputs = (h, t, r) if rng.random() < 0.5 else tuple(torch.from_numpy(x).to(gpu_device) for x in (h, t, r))
    gd = L.KGStructure.from_triples(n, *inputs, device=gpu_device)
    what = (seed, n, len(h), n_rel)
    assert (gd.n, gd.nnz, gd.n_raw) == (gh.n, gh.nnz, gh.n_raw), what
    for name in ("rowptr", "col", "eptr", "rel", "rel_first", "dup_entries", "dup_rows", "t_rowptr", "t_col", "t_perm"):
        a, b = gd.host(name), gh.host(name)
        assert (a is None) == (b is None), (name, what)
        if a is not None:
            assert a.dtype == b.dtype and np.array_equal(a, b), (name, what)
    assert np.array_equal(gd.order, gh.order), what


# ============================================================================= the tall GEMM, shape by shape
TALL_CASES = int(os.environ.get("LKG_FUZZ_TALL_CASES", "30"))


@pytest.mark.parametrize("seed", [13000 + i for i in range(TALL_CASES)])
def test_tall_gemm_against_float64_over_drawn_shapes(gpu_device, seed):
    """lkg_gemm_tall_f32 (one to three K-panels from different arrays, aligned and unaligned, strided; output widths across the
    column-tile boundary; both weight layouts; alpha / beta / bias; row counts off the 128-row tile) against the float64
    product: within 3 x an fp32 GEMM's own error of the same product (2e-6 of the largest entry at least)."""
    import __graft_entry__ as ge
    ge.build()
    from literalkg_amd import ops
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    m = int(pick([16384, 16385, 16384 + 127, 20_000, 33_333]))
    n = int(pick([1, 5, 32, 50, 64, 100, 128, 200, 255, 256, 257, 300, 384, 512]))
    ks = [int(pick([1, 2, 3, 7, 16, 30, 64, 100, 128, 255, 256, 300])) for _ in range(int(pick([1, 2, 3])))]
    tb = bool(rng.random() < 0.7)
    if not ops.tall_ok(m, n, ks, single_panel_too=True):
        pytest.skip(f"not a product the tall engine takes: {m} x {n} x {ks}")

    def panel(k):
        kind = pick(["plain", "offset", "wide"])
        if kind == "plain":
            return torch.randn(m, k, generator=gen).to(gpu_device)
        off = int(pick([1, 3, 4]))
        return torch.randn(m, k + off + 2, generator=gen).to(gpu_device)[:, off:off + k]
    panels = [panel(k) for k in ks]
    blocks = [(torch.randn((n, k) if tb else (k, n), generator=gen) * 0.1).to(gpu_device) for k in ks]
    a64 = torch.cat([p.double() for p in panels], 1)
    b64 = torch.cat([b.double() if tb else b.double().t() for b in blocks], 1)
    want = a64 @ b64.t()
    scale = float(want.abs().max()) + 1e-30
    f32 = float(((a64.float() @ b64.float().t()).double() - want).abs().max()) / scale
    what = (seed, m, n, ks, tb)
    got = ops.gemm_tall(panels, (blocks,), tb)
    err = float((got.double() - want).abs().max()) / scale
    # (floor: the operands carry 22 significant bits -- a K = 1 "product" is 2^-22 off where an fp32 multiply is 2^-24 off)
    assert err <= max(3.0 * f32, 2e-6), (what, err, f32)
    bias = torch.randn(n, generator=gen).to(gpu_device)
    alpha, beta = float(pick([1.0, 0.5, -2.0])), float(pick([0.0, 1.0, 2.0]))
    c0 = torch.randn(m, n + 5, generator=gen).to(gpu_device)[:, 5:] if rng.random() < 0.5 else torch.randn(m, n, generator=gen).to(gpu_device)
    got2 = ops.gemm_tall(panels, (blocks,), tb, bias, alpha=alpha, beta=beta, out=c0.clone())
    want2 = alpha * want + beta * c0.double() + bias.double()
    scale2 = float(want2.abs().max()) + 1e-30
    assert float((got2.double() - want2).abs().max()) / scale2 <= max(6.0 * f32, 4e-6), (what, alpha, beta)
