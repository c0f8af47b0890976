"""Where does gat_trans_M's gradient of fuzz case 44053 (residual-only sweep) leave the oracle?  (debug aid)"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import literalkg_amd as L
from oracle import literalkg_oracle as O
import test_gpu_fuzz as F
from literalkg_amd import io
from literalkg_amd.synth import make_batch, make_kg
dev = torch.device("cuda:0")
seed = 44053
c = F.draw(seed); c["residual"] = True
for k_, v_ in json.loads(os.environ.get("OVERRIDE", "{}")).items(): c[k_] = v_
print(c)
n, n_rel = c["n"], c["n_rel"]
h, t, r = make_kg(n, c["e"], c["skew"], seed=seed)
r = np.random.default_rng(seed + 1).integers(0, n_rel, len(r))
_, first = np.unique(np.stack([h, r, t], 1), axis=0, return_index=True)
h, t, r = h[first], t[first], r[first]
cfg = O.default_cfg(embed_dim=c["dim"], relation_dim=c["rel_dim"], conv_dim=c["conv"], n_conv_layers=c["layers"], aggregation_type=c["agg"],
                    scale_gat_dim=c["scale"], use_residual=c["residual"], use_num_lit=c["gate"] in ("mul", "num"),
                    use_txt_lit=c["gate"] in ("mul", "txt"), txt_lit_dim=c["txt_dim"], mlp_hidden_dim=c["mlp_hidden"], kg_l2loss_lambda=1e-4,
                    fine_tuning_l2loss_lambda=1e-4, pre_training_neg_rate=c["neg"], fine_tuning_neg_rate=c["neg"], device=dev)
torch.manual_seed(seed)
a_in = io.initial_a_in(n, h, t, r)
m = L.LiteralKG(cfg, n, n_rel, a_in, None, None, scoring=c["scoring"])
with torch.no_grad():
    m.entity_embed.weight.mul_(c["weight_scale"]); m.relation_embed.weight.mul_(min(c["weight_scale"], 3.0))
m.to(dev).eval(); m.prune_to_batch = c["prune"]
bh, br, bp, bn = (torch.from_numpy(x) for x in make_batch(n, c["batch"], c["neg"], seed=seed + 2))
br = torch.from_numpy(np.repeat(np.random.default_rng(seed + 3).integers(0, n_rel, c["batch"]), c["neg"]))
loss = m(bh.to(dev), br.to(dev), bp.to(dev), bn.to(dev), device=dev, mode="pre_training")
loss.backward()
emb = m.gat_embed.detach().double().cpu()          # the projected table the HIP path produced
M = m.gat_trans_M.detach().double().cpu().requires_grad_(True)
rel = m.relation_embed.weight.detach().double().cpu().requires_grad_(True)
W = M[br]
rh = torch.bmm(emb[bh].unsqueeze(1), W).squeeze(1); rp = torch.bmm(emb[bp].unsqueeze(1), W).squeeze(1); rn = torch.bmm(emb[bn].unsqueeze(1), W).squeeze(1)
re = rel[br]
pos = ((rh + re - rp) ** 2).sum(1); neg = ((rh + re - rn) ** 2).sum(1)
l2 = lambda x: (x ** 2).sum(1).mean() / 2
want = (-torch.nn.functional.logsigmoid(neg - pos)).mean() + 1e-4 * (l2(rh) + l2(re) + l2(rp) + l2(rn))
want.backward()
g, w = m.gat_trans_M.grad.double().cpu(), M.grad
print("loss hip", float(loss), "f64 from hip's table", float(want))
print("g_M: hip vs f64-from-hip's-table", float((g - w).abs().max() / w.abs().max()), "largest", float(w.abs().max()))
print("g_rel:", float((m.relation_embed.weight.grad.double().cpu() - rel.grad).abs().max() / rel.grad.abs().max()))
print("embedding magnitudes: max |emb|", float(emb.abs().max()), "rms", float(emb.pow(2).mean().sqrt()), " M max", float(M.abs().max()), " weight_scale", c["weight_scale"])
# ---- the tables: hip vs the fp32 oracle vs the float64 oracle, row by row
params = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if k != "A_in"}
t32 = O.gat_embeddings(params, cfg, a_in, None, None)
p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
t64 = O.gat_embeddings(p64, cfg, a_in.double(), None, None)
hip = m.gat_embed.detach().cpu().double()
d_hip = (hip - t64).abs().max(1).values
d_o32 = (t32.double() - t64).abs().max(1).values
print("table rows: hip vs f64   max", float(d_hip.max()), "mean", float(d_hip.mean()), "| o32 vs f64   max", float(d_o32.max()), "mean", float(d_o32.mean()))
worst = torch.topk(d_hip, 5)
print("worst hip rows", worst.indices.tolist(), worst.values.tolist(), " their o32 error", d_o32[worst.indices].tolist())
used = torch.unique(torch.cat([bh, bp, bn]))
print("rows the batch reads:", used.numel(), " hip err on them max", float(d_hip[used].max()), " o32", float(d_o32[used].max()))
def g_m_from(table):
    M_ = m.gat_trans_M.detach().double().cpu().requires_grad_(True)
    W_ = M_[br]; e = table.double()
    rh = torch.bmm(e[bh].unsqueeze(1), W_).squeeze(1); rp = torch.bmm(e[bp].unsqueeze(1), W_).squeeze(1); rn = torch.bmm(e[bn].unsqueeze(1), W_).squeeze(1)
    re_ = m.relation_embed.weight.detach().double().cpu()[br]
    pos = ((rh + re_ - rp) ** 2).sum(1); neg = ((rh + re_ - rn) ** 2).sum(1)
    ((-torch.nn.functional.logsigmoid(neg - pos)).mean() + 1e-4 * (l2(rh) + l2(re_) + l2(rp) + l2(rn))).backward()
    return M_.grad
g64, g_h, g_o = g_m_from(t64), g_m_from(hip), g_m_from(t32)
sc = float(g64.abs().max())
print("g_M in float64 from: hip's table vs f64 table", float((g_h - g64).abs().max()) / sc, " o32 table vs f64 table", float((g_o - g64).abs().max()) / sc, " hip op result vs f64 table", float((g - g64).abs().max()) / sc, "largest", sc)
