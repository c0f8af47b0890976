#!/usr/bin/env python3
"""bench.py -- LiteralKG aggregation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torch.distributed environment the script starts its own N workers (a child
`python -m torch.distributed.run`, before anything touches the GPU) and passes their JSON line and exit code through;
under torch.distributed.run it is one rank per GPU.

Metric (BASELINE.json): KG edges aggregated / s, 1 GAT layer, dim 256, plus % of the 8 TB/s HBM roofline.
A STEP is one pass of the aggregation layer's sparse hot path over the whole graph:
    forward   side      = A_in   @ ego        (lkg_spmm_csr_f32 on the CSR,  model.py:106)
    backward  grad_ego  = A_in^T @ grad_side  (the same kernel on the CSC,   autograd of model.py:106)
    N > 1     + the exchange step of the sharding scheme (both schemes are measured, see --sharding)
A_in holds a real refreshed attention (edge logits + row softmax from the fused K1+K2 kernel).
`value` counts one aggregation per edge per pass: 2 * E edge-aggregations per step / wall time.
Workload at N = 1: synthetic KG 1M entities / 10M edges, D = 256 (the config the metric is quoted on);
at N > 1 every rank contributes a head-row range of 625k entities / 12.5M edges (N = 8 is BASELINE config[3]:
5M entities / 100M edges): weak scaling.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SPMM_SOURCE = os.path.join(ROOT, "literalkg_amd", "csrc", "lkg_spmm.hip")


def algorithmic_bytes(nnz, n_out_rows, d):
    """one D-row gather + int32 col + fp32 value per stored entry, one output row per
    destination, the row pointer array."""
    return nnz * (4 * d + 8) + n_out_rows * 4 * d + 4 * (n_out_rows + 1)


def source_sha(path=SPMM_SOURCE):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def pmc_traffic(by, skew="zipf"):
    """HBM bytes per forward launch from the newest committed rocprofv3 --pmc summary (tools/pmc_traffic.py; the
    counters cannot be read from inside the process).  Reported only when that summary was taken on THIS kernel
    source (sha of lkg_spmm.hip stored in the summary) and on this workload; otherwise null."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))
                   if "_gate_" not in os.path.basename(f))              # (the gate kernel's summaries are read further down)
    if not files:
        return None, "no PMC summary committed"
    recs = [(f, json.load(open(f))) for f in files]
    match = [fr for fr in recs if fr[1].get("spmm_source_sha16") == source_sha()]
    if not match:
        f, rec = recs[-1]
        return None, (f"{os.path.relpath(f, ROOT)} was taken on another lkg_spmm.hip "
                      f"({rec.get('spmm_source_sha16')} != {source_sha()})")
    path, rec = match[-1]
    name = os.path.relpath(path, ROOT)
    tr = rec.get("traffic_bytes_fwd")
    # (the summaries are taken on bench.py's default workload: zipf heads unless the record says otherwise)
    if tr is None or abs(tr - by) > 0.25 * by or rec.get("skew", "zipf") != skew:      # different shape: not comparable
        return None, f"{name} was taken on another workload"
    return tr, name


def cpu_baseline(g, val, n, d, seed):
    """The oracle (the ATen ops the reference dispatches: sparse COO matmul forward, its autograd
    transpose product backward) timed on this box's host cores on the SAME graph and attention values:
    1 forward + 1 backward pass."""
    from oracle import literalkg_oracle as O
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(seed)
    bound = (6.0 / (n + d)) ** 0.5
    ent = (torch.rand((n, d), generator=gen) * 2 - 1) * bound
    rp = g.host("rowptr")
    idx = torch.from_numpy(np.stack([np.repeat(np.arange(n, dtype=np.int64), np.diff(rp)),
                                     g.host("col").astype(np.int64)]))
    a = torch.sparse_coo_tensor(idx, val.cpu(), (n, n), is_coalesced=True)
    x = ent.clone().requires_grad_(True)
    t0 = time.perf_counter()
    side = O.aggregate(a, x)
    t1 = time.perf_counter()
    side.backward(torch.ones_like(side))
    t2 = time.perf_counter()
    edges = 2 * a._nnz()
    return {"value": edges / (t2 - t0), "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"1 forward ({t1 - t0:.2f} s) + 1 backward ({t2 - t1:.2f} s) torch-CPU sparse matmul pass over the "
                      f"same graph ({a._nnz()} stored entries, D={d}), torch {torch.__version__}, "
                      f"{cores} threads"}


def _timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    ms = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ms))


def c3_step_timing(h, t, r, n, d, dev):
    """BASELINE config[2] as a module on the same graph: 2 gcn layers + GateMul (2 numeric, 300 text literals), TransR,
    batch 2049 triples -- pre_training forward and forward + backward (median of 10 after 3 warm-ups)."""
    from types import SimpleNamespace
    import literalkg_amd as L
    from literalkg_amd.synth import make_batch
    cfg = SimpleNamespace(use_pretrain=0, device=dev, embed_dim=d, relation_dim=d, scale_gat_dim=None,
                          use_residual=False, alpha=0.1, lamda=0.5, aggregation_type="gcn", n_conv_layers=2,
                          conv_dim=d, mess_dropout=0.1, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
                          pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300,
                          use_num_lit=True, use_txt_lit=True, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64)
    gen = torch.Generator(device="cpu").manual_seed(7)
    num = torch.rand(n, 2, generator=gen).to(dev)
    txt = torch.randn(n, 300, generator=gen).to(dev)
    model = L.LiteralKG(cfg, n, 16, None, num, txt).to(dev)
    hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
    model(hd, td, rd, list(range(16)), device=dev, mode="update_att")
    batch = [torch.from_numpy(a).to(dev) for a in make_batch(n, 683, 3)]
    model.train()

    def fwd_bwd():
        model.zero_grad(set_to_none=True)
        model(*batch, device=dev, mode="pre_training").backward()
    with torch.no_grad():
        fwd = _timed(lambda: model(*batch, device=dev, mode="pre_training"))
    step = _timed(fwd_bwd)
    e = len(h)
    out = {"config": f"LiteralKG gcn x2 + GateMul (2 + 300 literals), D={d}, TransR, dropout 0.1, batch 2049 triples, same graph",
           "pre_training_forward_ms": fwd, "pre_training_forward_backward_ms": step,
           "pre_training_step_edges_per_s": 2 * e / step * 1e3,
           "note": "the backward of the last layer runs on the <= 3B rows the loss reaches and that of the layer below on the "
                   "frontier they reach (exact: only zero addends are dropped); the forward is the full-graph forward"}
    del model
    return out


def default_arch_step_timing(h, t, r, n, dev, main_py=False):
    """The reference's DEFAULT architectures on the same graph -- pre_training forward and forward + backward (median of 10
    after 3 warm-ups), batch 2049 triples, dropout 0.1, GateMul (2 numeric + 300 text literals), TransR 300 x 300:
    main_pretraining.py (argument_pretraining.py:34-62): embed_dim = relation_dim = scale_gat_dim = 300, eight gcn layers of 32;
    main.py (argument.py:34-118, main_py=True): bi-interaction with the GCNII-style residual, scale_gat_dim = 256."""
    from types import SimpleNamespace
    import literalkg_amd as L
    from literalkg_amd.synth import make_batch
    cfg = SimpleNamespace(use_pretrain=0, device=dev, embed_dim=300, relation_dim=300, scale_gat_dim=256 if main_py else 300,
                          use_residual=bool(main_py), alpha=0.1, lamda=0.5,
                          aggregation_type="bi-interaction" if main_py else "gcn", n_conv_layers=8,
                          conv_dim=32, mess_dropout=0.1, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
                          pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300,
                          use_num_lit=True, use_txt_lit=True, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64)
    gen = torch.Generator(device="cpu").manual_seed(7)
    num = torch.rand(n, 2, generator=gen).to(dev)
    txt = torch.randn(n, 300, generator=gen).to(dev)
    model = L.LiteralKG(cfg, n, 16, None, num, txt).to(dev)
    hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
    model(hd, td, rd, list(range(16)), device=dev, mode="update_att")
    batch = [torch.from_numpy(a).to(dev) for a in make_batch(n, 683, 3)]
    model.train()

    def fwd_bwd():
        model.zero_grad(set_to_none=True)
        model(*batch, device=dev, mode="pre_training").backward()
    with torch.no_grad():
        fwd = _timed(lambda: model(*batch, device=dev, mode="pre_training"))
    step = _timed(fwd_bwd)
    what = ("main.py's defaults: bi-interaction + GCNII-style residual x8 (conv_dim 32), linear_gat 556 -> 256" if main_py else
            "main_pretraining.py's defaults: gcn x8 (conv_dim 32), linear_gat 556 -> 300")
    out = {"config": f"the reference's default architecture ({what}) over 300-wide embeddings + GateMul (2 + 300 literals), "
                     "TransR 300 x 300, dropout 0.1, batch 2049 triples, same graph",
           "pre_training_forward_ms": fwd, "pre_training_forward_backward_ms": step,
           "pre_training_step_edges_per_s": 8 * len(h) / step * 1e3}
    del model
    return out


def whole_path_timings(h, t, r, n, d, dev):
    """Context numbers for the same graph (SURVEY.md 8d ii-iv), outside the headline metric: the drop-in
    module's update_att, one pre_training step (1 gcn layer, D=d, TransR, 2049 triples) forward /
    forward+backward, one aggregation layer, the layer's Linear and the literal gate against the MFMA peak.
    Median of 10 after 3 warm-ups, host-timed around a device sync."""
    from types import SimpleNamespace
    import literalkg_amd as L
    from literalkg_amd import ops
    from literalkg_amd.synth import make_batch
    cfg = SimpleNamespace(use_pretrain=0, device=dev, embed_dim=d, relation_dim=d, scale_gat_dim=None,
                          use_residual=False, alpha=0.1, lamda=0.5, aggregation_type="gcn", n_conv_layers=1,
                          conv_dim=d, mess_dropout=0.1, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
                          pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300,
                          use_num_lit=False, use_txt_lit=False, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64)
    model = L.LiteralKG(cfg, n, 16).to(dev)
    hd, td, rd = (torch.from_numpy(a).to(dev) for a in (h, t, r))
    batch = [torch.from_numpy(a).to(dev) for a in make_batch(n, 683, 3)]

    rel = list(range(16))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model(hd, td, rd, rel, device=dev, mode="update_att")
    torch.cuda.synchronize()
    upd_first = (time.perf_counter() - t0) * 1e3           # includes the structure build of the edge lists
    upd = _timed(lambda: model(hd, td, rd, rel, device=dev, mode="update_att"))
    model.train()

    def fwd_bwd():
        model.zero_grad(set_to_none=True)
        model(*batch, device=dev, mode="pre_training").backward()
    with torch.no_grad():
        fwd = _timed(lambda: model(*batch, device=dev, mode="pre_training"))
    step = _timed(fwd_bwd)
    model.prune_to_batch = True          # exact: layers evaluated on the batch's L-hop frontier only
    pruned = _timed(fwd_bwd)
    model.prune_to_batch = False
    # (ii) one aggregation layer forward (a1 + a2): SpMM(+self), Linear on the MFMA GEMM, LeakyReLU/LayerNorm/
    # dropout/normalise epilogue; and the dense GEMM of that layer alone against the MFMA peak
    att = model._attention()
    ego = model.entity_embed.weight.detach()
    layer = model.aggregator_layers[0]
    with torch.no_grad():
        layer_ms = _timed(lambda: layer(ego, att, [ego], model.lamda, model.alpha, 1))
        w, b = layer.linear.weight.detach(), layer.linear.bias.detach()
        # the layer's Linear as the layer runs it: its input arrives from the SpMM with its row maxima (no scale pass)
        rm = ops.row_absmax(ego)
        y = torch.empty_like(ego)
        tall_ms = _timed(lambda: ops.gemm_tall((ego,), ((w,),), True, b, out=y, rowmax=rm))
        ops_engine = ops._ENGINE
        ops._ENGINE = "bf16x3"                                   # round-1 engine (still serves untagged one-panel products)
        gemm_ms = _timed(lambda: ops.gemm(ego, w, trans_b=True, bias=b, out=y))
        ops._ENGINE = ops_engine
        del y
        # (iv) the attention refresh kernel alone (K1 + K2: per-edge logit, merge, row softmax) against the HBM roofline
        g_att = att.graph
        val_buf = torch.empty(g_att.nnz, dtype=torch.float32, device=dev)
        rel_w = model.relation_embed.weight.detach()
        att_ms = _timed(lambda: ops.edge_softmax(g_att, ego, rel_w, out=val_buf))
        att_bytes = g_att.nnz * (4 * d + 12) + n * 4 * d + rel_w.shape[0] * 4 * d + 4 * (n + 1)
        del val_buf
    flops = 2.0 * n * d * d
    e = len(h)
    out = {"config": f"LiteralKG gcn x1, D={d}, TransR, dropout 0.1, batch 2049 triples, same graph",
           "roofline_attention": {"bound": "hbm", "kernel": "edge_softmax_kernel (lkg_edge_softmax_f32: logit + merge + row softmax, "
                                                            "one launch)",
                                  "algorithmic_bytes": att_bytes, "avg_launch_ms": att_ms,
                                  "achieved": att_bytes / att_ms / 1e6, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                  "frac": att_bytes / att_ms / 1e6 / HBM_PEAK_GBS,
                                  "note": "one random WHOLE row (4 d bytes) gathered per stored entry from the entity table: the "
                                          "access pattern MI355X_MICROARCH.md measures at 5.5-5.8 TB/s chip-wide; the SpMM "
                                          "passes that ceiling with 128-column slabs, a dot product over the whole row cannot "
                                          "(DESIGN.md section 9: two rewrites measured)"},
           "update_att_first_call_ms": upd_first,
           "update_att_ms": upd, "update_att_edges_per_s": e / upd * 1e3,
           "pre_training_forward_ms": fwd, "pre_training_forward_backward_ms": step,
           "pre_training_step_edges_per_s": e / step * 1e3,
           "pre_training_forward_backward_ms_prune_to_batch": pruned,
           "layer_forward_ms": layer_ms, "layer_forward_edges_per_s": e / layer_ms * 1e3,
           "roofline_gemm": {"bound": "mfma",
                             "kernel": f"gemm_tall_kernel<256, plain, one accumulator> Linear forward {n}x{d}x{d}: f32 product "
                                       f"as 3 v_mfma_f32_32x32x16_f16 per 16 k (row-scaled exact fp16 hi/mid split, f32-accurate)",
                             "achieved": flops / tall_ms / 1e9, "unit": "TFLOP/s (f32-equivalent)",
                             "peak": 2500.0 / 3, "frac": flops / tall_ms / 1e9 / (2500.0 / 3),
                             "avg_launch_ms": tall_ms,
                             # round 1 priced this product against the bf16 pipe / 6 products (416.7 TF: 0.376 then); the
                             # same yardstick for this engine, so that the two rounds compare
                             "frac_of_bf16_pipe_over_6": flops / tall_ms / 1e9 / (2500.0 / 6),
                             "hbm_frac": 2.0 * n * d * 4 / tall_ms / 1e6 / HBM_PEAK_GBS,
                             "note": "peak = dense fp16 MFMA peak / 3 products (the f32-input MFMA it replaces peaks at 157.3 "
                                     "TFLOP/s); this shape moves 2 GB (read x, write y): its HBM floor is ~0.31 ms, so it is as "
                                     "much HBM- as MFMA-bound (hbm_frac = those bytes / time / 8 TB/s)",
                             "bf16x3_engine": {"avg_launch_ms": gemm_ms, "achieved": flops / gemm_ms / 1e9,
                                               "frac_of_bf16_pipe_over_6": flops / gemm_ms / 1e9 / (2500.0 / 6)}}}
    del model, att, layer
    out["gate"] = gate_timing(n, d, dev)
    out["c3_step"] = c3_step_timing(h, t, r, n, d, dev)
    torch.cuda.empty_cache()
    out["default_architecture_step"] = default_arch_step_timing(h, t, r, n, dev)
    torch.cuda.empty_cache()
    out["main_py_default_architecture_step"] = default_arch_step_timing(h, t, r, n, dev, main_py=True)
    return out


def c4_regime_timing(d, dev, n=5_000_000, e=100_000_000):
    """The SpMM pair where NOTHING fits a cache (BASELINE config[3]'s graph on one GPU: a 5 GB source table, 20 entries
    per row): what one rank of the row-range scheme runs per 1/N of the rows.  Graph drawn on the device (zipf heads)."""
    import literalkg_amd as L
    from literalkg_amd import ops
    from literalkg_amd.synth import make_kg_device
    # the two 5 GB tables FIRST, as a model allocates its embeddings before any edge list arrives: allocated behind the graph
    # generator's and the structure build's temporaries the same launches measured 10 % slower (19.2 / 18.3 ms against 17.6 /
    # 16.7 ms in one process on one graph, tools/graph_family_check.py: placement in physical memory, not the kernel)
    x = torch.randn((n, d), device=dev) * 0.05
    out = torch.empty((n, d), device=dev)
    h, t, r = make_kg_device(n, e, "zipf", 2022, dev)
    g = L.KGStructure.from_triples(n, h, t, r, device=dev)
    del h, t, r
    val = torch.rand(g.nnz, device=dev)
    val_t = ops.permute_values(val, g.t_perm)

    def ev_time(fn, reps=8):
        for _ in range(2):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))
    fwd = ev_time(lambda: ops.spmm_raw(g.rowptr, g.col, val, x, n, out=out, long_rows=g.long_rows(False)))
    bwd = ev_time(lambda: ops.spmm_raw(g.t_rowptr, g.t_col, val_t, x, n, out=out, long_rows=g.long_rows(True)))
    by = algorithmic_bytes(g.nnz, n, d)
    copy_ms = ev_time(lambda: out.copy_(x))            # the box's plain-copy rate, same process, same two 5 GB tables
    res = {"plain_copy_GBs": 2 * x.numel() * 4 / copy_ms / 1e6,
           "config": f"{n} entities / {g.nnz} stored entries, D={d}, one GPU (zipf heads drawn on the device, out-degree clipped at 4096 like the N = 1 graph)",
           "algorithmic_bytes": by, "fwd_ms": fwd, "bwd_ms": bwd,
           "fwd_frac_of_hbm_roofline": by / fwd / 1e6 / HBM_PEAK_GBS, "bwd_frac_of_hbm_roofline": by / bwd / 1e6 / HBM_PEAK_GBS,
           "edges_per_s": 2 * g.nnz / (fwd + bwd) * 1e3}
    del g, val, val_t, x, out
    torch.cuda.empty_cache()
    return res


def gate_timing(n, d, dev):
    """K6 at the C3 shape: GateMul forward over n x (d + 2 + 300) (gate.py:22-28)."""
    import literalkg_amd as L
    gate = L.GateMul(d, 2, 300).to(dev)
    x = torch.randn(n, d, device=dev) * 0.05
    num = torch.rand(n, 2, device=dev)
    txt = torch.randn(n, 300, device=dev)
    out = torch.empty(n, d, device=dev)
    with torch.no_grad():
        ms = _timed(lambda: gate(x, num, txt, out))
    flops = 2.0 * n * (d + 302) * d * 2
    alg = 4.0 * n * (d + 302 + d)
    traffic, src = None, "no PMC summary committed"
    import glob
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gate_pmc_traffic.json")))
    rec_path = recs[-1] if recs else os.path.join(ROOT, "profiles", "none")
    if os.path.exists(rec_path):          # counter-measured HBM bytes of this launch, valid for THIS kernel source only
        rec = json.load(open(rec_path))
        sha = source_sha(os.path.join(ROOT, "literalkg_amd", "csrc", "lkg_gemm_tall.hip"))
        if rec.get("tall_source_sha16") == sha and abs(rec.get("algorithmic_bytes", 0) - alg) < 1:
            traffic, src = rec["traffic_bytes"], os.path.relpath(rec_path, ROOT)
        else:
            src = f"{os.path.relpath(rec_path, ROOT)} was taken on another lkg_gemm_tall.hip or shape"
    return {"kernel": f"GateMul forward {n} x ({d}+2+300) -> {d}: ONE launch of gemm_tall_kernel (f32 product as 3 "
                      f"v_mfma_f32_32x32x16_f16 per 16 k on a row-scaled exact fp16 hi/mid split, blend epilogue)",
            "ms": ms, "tflops_f32_equivalent": flops / ms / 1e9,
            "mfma_pipe_frac": 3 * flops / ms / 1e9 / 2500.0,
            "algorithmic_bytes": alg, "algorithmic_GBs": alg / ms / 1e6, "frac_of_hbm_roofline": alg / ms / 1e6 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src,
            "traffic_over_algorithmic": (traffic / alg) if traffic else None}


def sharded_step_timings(world, rank, dev, n_glob, d, h_own, t_own, r_own, steps):
    """N > 1 context numbers, outside the headline metric: the INTEGRATED multi-GPU pre_training step of
    literalkg_amd/distributed.py (row-sharded module: 1 gcn layer, D=d, TransR, 2049 triples, fused Adam on each rank's
    shard) under both aggregation exchange schemes, forward + backward + gradient sync + optimizer step."""
    from types import SimpleNamespace
    import literalkg_amd as L
    from literalkg_amd.distributed import ShardedLiteralKG
    from literalkg_amd.optim import Adam
    from literalkg_amd.synth import make_batch
    cfg = SimpleNamespace(use_pretrain=0, device=dev, embed_dim=d, relation_dim=d, scale_gat_dim=None,
                          use_residual=False, alpha=0.1, lamda=0.5, aggregation_type="gcn", n_conv_layers=1,
                          conv_dim=d, mess_dropout=0.1, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
                          pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300,
                          use_num_lit=False, use_txt_lit=False, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64)
    batch = [torch.from_numpy(a).to(dev) for a in make_batch(n_glob, 683, 3)]
    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    out = {}
    for scheme in ("rows", "features"):
        torch.manual_seed(1234)                                  # the same replicated weights on every rank
        full = L.LiteralKG(cfg, 1, 16)                           # weights only; the entity shard is drawn per rank
        state = {k: v for k, v in full.state_dict().items() if k not in ("A_in", "entity_embed.weight")}
        part_rows = -(-n_glob // world)
        lo = min(n_glob, rank * part_rows)
        rows = min(n_glob, lo + part_rows) - lo
        from literalkg_amd.distributed import RowPartition, HipKernels
        local = L.LiteralKG(cfg, rows, 16)
        local.load_state_dict(state, strict=False)
        local.to(dev)
        model = ShardedLiteralKG(local, RowPartition(n_glob, rank, world), scheme, HipKernels())
        if scheme == "features":                                 # replicated structure: every rank needs all triples
            mine = torch.from_numpy(np.stack([h_own, t_own, r_own])).to(cdev).reshape(-1)
            every = torch.empty(world * mine.numel(), dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(every, mine)
            every = every.view(world, 3, -1)
            hh, tt, rr = (every[:, i].reshape(-1).to(dev) for i in range(3))
        else:
            hh, tt, rr = (torch.from_numpy(a).to(dev) for a in (h_own, t_own, r_own))
        model(hh, tt, rr, list(range(16)), device=dev, mode="update_att")
        model.train()
        opt = Adam(model.parameters(), lr=1e-4)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = model(*batch, device=dev, mode="pre_training")
            loss.backward()
            model.sync_gradients()
            opt.step()
        from literalkg_amd import distributed as D
        for _ in range(2):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        D.TRAFFIC.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64).to(cdev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        out[scheme] = {"ms_per_step": float(el) / steps * 1e3, "steps": steps,
                       "bytes_sent_per_step_rank0": {k: v // steps for k, v in sorted(D.TRAFFIC.items())}}
        del model, opt, local
        torch.cuda.empty_cache()
    out["config"] = f"ShardedLiteralKG gcn x1, D={d}, TransR, dropout 0.1, batch 2049 triples, {n_glob} entities over {world} ranks"
    return out


def self_launch(args):
    """--gpus N > 1 without a torch.distributed environment: start the N workers as a CHILD process (nothing in this
    process has touched the GPU yet, and the child is a new program, not an exec of this one)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def spot_rows(g, val, table, rows):
    """side[rows] evaluated with plain torch ops from the full table (the bench's own sanity check of a sharded
    result; it launches no kernel of the library, so the profiles of this script hold the timed launches only)."""
    rp = g.host("rowptr")
    out = []
    for i in rows:
        sl = slice(int(rp[i]), int(rp[i + 1]))
        out.append((val[sl][:, None] * table[g.col[sl].long()]).sum(0, keepdim=True))
    return torch.cat(out)


class Timer:
    """K timed steps between barrier + device sync on both sides; MAX over ranks."""

    def __init__(self, world, cdev):
        self.world, self.cdev = world, cdev

    def run(self, step, steps, n_events):
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(n_events)] for _ in range(steps)]
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(events[i])
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if self.world > 1:
            el = el.to(self.cdev)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el), events


class Watchdog:
    """N > 1 only.  The robust measurements come first; every optional phase after them (the pipelined exchange, which
    had never run over RCCL on the development boxes; the rows scheme; the integrated step) runs under a deadline.  If a
    phase hangs -- a collective that never completes cannot be caught as an exception -- rank 0 prints the JSON line of
    what HAS been measured, naming the phase, and every rank leaves with exit code 3: a hang is a defect of the run, and the
    run's records must say so (the partial line is there to find its cause from, not to pass as a result)."""

    def __init__(self, rank, emit):
        self.rank, self.emit, self.timer, self.phase = rank, emit, None, None

    def arm(self, seconds, phase):
        import threading
        self.disarm()
        self.phase = phase
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _fire(self):
        try:
            if self.rank == 0:
                msg = f"HUNG: phase '{self.phase}' did not finish within its deadline; run abandoned with exit code 3"
                sys.stderr.write("bench.py: " + msg + "\n")
                sys.stderr.flush()
                out = self.emit(msg)
                if out is not None:
                    out["hung_phase"] = self.phase
                    sys.stdout.write(json.dumps(out) + "\n")
                    sys.stdout.flush()
        finally:
            os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--entities", type=int, default=None, help="per-GPU entities (default 1M at N=1, 625k at N>1)")
    ap.add_argument("--edges", type=int, default=None, help="per-GPU edges (default 10M at N=1, 12.5M at N>1)")
    ap.add_argument("--skew", default="zipf", choices=["zipf", "uniform"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the whole-path context timings (N=1)")
    ap.add_argument("--chunks", type=int, default=4, help="rows scheme: tail-row chunks of the backward (comm overlap)")
    ap.add_argument("--sharding", default="both", choices=["both", "features", "rows"],
                    help="N>1: 'features' = column-sharded tables, no collective in the SpMM, all-to-all exchange (the "
                         "headline `value`); 'rows' = head-row ranges + all-reduce of the entity-gradient table (the "
                         "scheme BASELINE.json's north star names); 'both' (default) times features for `value` and "
                         "rows next to it in `rows_scheme`")
    ap.add_argument("--no-sharded-step", action="store_true",
                    help="N>1: skip the context timing of the integrated multi-GPU pre_training step")
    ap.add_argument("--no-overlap", action="store_true",
                    help="features scheme: plain all-to-all after the SpMM instead of per-range sends behind it")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debug only: all ranks share cuda:0 and talk over gloo (N>1 code path on a 1-GPU box); "
                         "the numbers of such a run are meaningless")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
            cdev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    import literalkg_amd as L
    from literalkg_amd import ops
    from literalkg_amd._native import LkgError
    from literalkg_amd.sharding import FeatureShardedAggregation, ShardedAggregation
    from literalkg_amd.synth import make_kg, xavier_table

    d = args.dim
    n_loc = args.entities or (1_000_000 if world == 1 else 625_000)
    e_loc = args.edges or (10_000_000 if world == 1 else 12_500_000)
    n_glob = n_loc * world
    lo, hi = rank * n_loc, (rank + 1) * n_loc
    schemes = ["none"] if world == 1 else (["features", "rows"] if args.sharding == "both" else [args.sharding])
    timer = Timer(world, cdev)

    # every rank draws the triples of ITS head rows (heads inside [lo, hi), tails over all entities)
    t0 = time.perf_counter()
    if world == 1:
        h, t, r = make_kg(n_glob, e_loc, args.skew, seed=2022)
    else:
        hl, _, r = make_kg(n_loc, e_loc, args.skew, seed=2022 + rank)
        h = hl + lo
        t = np.random.default_rng(4044 + rank).integers(0, n_glob, len(h), dtype=np.int64)
    g_own = L.KGStructure.from_triples(n_glob, h, t, r, device=dev)       # this rank's head rows (N = 1: everything)
    t_build = time.perf_counter() - t0
    relemb = xavier_table(16, d, dev, seed=7)
    probe = [lo + int(x) for x in np.random.default_rng(rank).integers(0, n_loc, 8)]

    def reduce_flags(valid, nnz, sum_nnz):
        valid_t = torch.tensor([float(valid)])
        nnz_t = torch.tensor([nnz], dtype=torch.int64)
        if world > 1:
            valid_t, nnz_t = valid_t.to(cdev), nnz_t.to(cdev)
            dist.all_reduce(valid_t, op=dist.ReduceOp.MIN)
            if sum_nnz:
                dist.all_reduce(nnz_t, op=dist.ReduceOp.SUM)
        return float(valid_t) == 1.0, int(nnz_t)

    # ------------------------------------------------------------------ N = 1 and the row-range scheme
    def run_rows():
        ent = xavier_table(n_glob, d, dev, seed=2022)              # full replica of the source table on every rank
        val, _ = ops.edge_softmax(g_own, ent, relemb, row_lo=lo, row_hi=hi)     # real attention values, own rows
        shard = ShardedAggregation(g_own, val, lo, hi, n_chunks=args.chunks if world > 1 else 1)
        want = spot_rows(g_own, val, ent, probe)
        grad_side = torch.randn((hi - lo, d), device=dev)
        side = torch.empty((hi - lo, d), device=dev)
        grad_table = torch.empty((n_glob, d), device=dev)

        def step(ev=None):
            if ev is not None:
                ev[0].record()
            shard.forward(ent, out=side)
            if ev is not None:
                ev[1].record()
            shard.backward(grad_side, out=grad_table)          # N > 1: chunked SpMM overlapped with the all-reduce
            if ev is not None:
                ev[2].record()
        for _ in range(max(args.warmup, 1)):
            step()
        ok = torch.allclose(side[[i - lo for i in probe]], want, rtol=1e-4, atol=1e-6)
        elapsed, events = timer.run(step, args.steps, 3)
        ok, total = reduce_flags(ok, g_own.nnz, True)
        fwd = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
        bwd = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
        res = {"elapsed": elapsed, "total_entries": total, "valid": ok, "fwd_ms": fwd, "bwd_ms": bwd,
               "fwd_bytes": algorithmic_bytes(g_own.nnz, hi - lo, d), "bwd_bytes": algorithmic_bytes(g_own.nnz, n_glob, d),
               "val": val, "exchange": "none" if world == 1 else
               f"all_reduce of the N x D entity-gradient table in {len(shard.chunks)} tail-row chunks behind the SpMM"}
        return res

    # ------------------------------------------------------------------ feature sharding
    def run_features():
        # feature sharding replicates the (small) structure: exchange the triple lists
        mine = torch.from_numpy(np.stack([h, t, r])).to(cdev).reshape(-1)
        every = torch.empty(world * mine.numel(), dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(every, mine)
        every = every.view(world, 3, -1)
        ha, ta, ra = (every[:, i].reshape(-1).cpu().numpy() for i in range(3))
        del every, mine
        g = L.KGStructure.from_triples(n_glob, ha, ta, ra, device=dev)
        ent = xavier_table(n_glob, d, dev, seed=2022)
        val, _ = ops.edge_softmax(g, ent, relemb)                  # replicated, like the structure
        fs = FeatureShardedAggregation(g, val, rank, world, d, [i * n_loc for i in range(world + 1)])
        slab = fs.column_slab(ent)
        want = spot_rows(g, val, ent, probe)
        del ent
        dg = fs.dg
        side_slab = torch.empty((n_glob, dg), device=dev)
        row_block = torch.empty((world, n_loc, dg), device=dev)
        grad_block = torch.randn((world, n_loc, dg), device=dev)     # stand-in for the dense part's gradient
        grad_slab = torch.randn((n_glob, dg), device=dev)
        grad_table = torch.empty((n_glob, dg), device=dev)
        plain = [True]

        def step(ev=None):
            if ev is not None:
                ev[0].record()
            if plain[0]:
                fs.forward(slab, out=side_slab)                # all edges, my columns: no collective
                if ev is not None:
                    ev[1].record()
                fs.to_row_block(side_slab, out=row_block)      # exchange: rows for the dense part ...
            else:                                              # ... or the same, sends hidden behind the SpMM
                fs.forward_to_row_block(slab, side_slab=side_slab, out=row_block)
                if ev is not None:
                    ev[1].record()
            if plain[0]:
                fs.to_column_slab(grad_block, out=grad_slab)   # the dense part's gradient back to column slabs
                if ev is not None:
                    ev[2].record()
                fs.backward(grad_slab, out=grad_table)         # A^T, my columns: no collective, no all-reduce
            else:                                              # the same in column pieces: A^T of piece p runs while
                if ev is not None:                             # piece p + 1 is still on the links
                    ev[2].record()
                fs.backward_from_row_block(grad_block, out=grad_table)
            if ev is not None:
                ev[3].record()

        def measure(exchange):
            for _ in range(max(args.warmup, 1)):
                step()
            got = torch.cat([row_block[:, i - lo, :].reshape(1, -1) for i in probe])
            ok = torch.allclose(got, want, rtol=1e-4, atol=1e-6)
            elapsed, events = timer.run(step, args.steps, 4)
            ok, total = reduce_flags(ok, g.nnz, False)
            xch = float(np.mean([e[1].elapsed_time(e[2]) for e in events])) if plain[0] else None
            return {"elapsed": elapsed, "total_entries": total, "valid": ok,
                    "step_bwd_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in events])),
                    "exchange_ms": xch, "exchange": exchange}

        # (1) the plain form first: one all-to-all after the SpMM, one before the transpose SpMM -- standard collectives
        res = measure("all_to_all")

        # the same loop without the layout exchange (the SpMM hot path alone, which needs no collective): the
        # roofline of the dominant KERNEL comes from here (in the pipelined step the forward is cut into row-range
        # launches interleaved with the sends); not part of the timed region above
        def bare(ev):
            ev[0].record()
            fs.forward(slab, out=side_slab)
            ev[1].record()
            fs.backward(grad_slab, out=grad_table)
            ev[2].record()
        bare_elapsed, kev = timer.run(bare, args.steps, 3)
        res.update({"fwd_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in kev])),
                    "bwd_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in kev])),
                    "fwd_bytes": algorithmic_bytes(g.nnz, n_glob, dg), "bwd_bytes": algorithmic_bytes(g.nnz, n_glob, dg),
                    "spmm_only_ms": bare_elapsed / args.steps * 1e3, "dg": dg})
        results["features"] = res

        # (2) the pipelined exchange (batched point-to-point sends behind the SpMM, async column pieces), under the
        # watchdog.  Only an error of the COLLECTIVE LIBRARY is tolerated (and reported); a kernel launch error or a
        # shape bug is re-raised.  The ranks agree on the outcome; if that agreement itself fails the run exits non-zero.
        if not args.no_overlap:
            watchdog.arm(300, "features scheme, pipelined exchange")
            plain[0] = False
            ok, why = 1.0, ""
            try:
                step()
                torch.cuda.synchronize()
            except LkgError:
                raise
            except dist.DistError as exc:
                ok, why = 0.0, f"{type(exc).__name__}: {exc}"
            flag = torch.tensor([ok], device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag) == 1.0:
                pipe = measure("pipelined")
                pipe.update({k: res[k] for k in ("fwd_ms", "bwd_ms", "fwd_bytes", "bwd_bytes", "spmm_only_ms", "dg")})
                results["features_pipelined"] = pipe
            else:
                res["pipelined_exchange_error"] = (why or "failed on another rank")[:300]
            watchdog.disarm()

    # ------------------------------------------------------------------ run + report
    def assemble(note=None):
        """The JSON line from whatever has been measured so far (rank 0)."""
        if not results:
            return None
        head = next(k for k in ("none", "features_pipelined", "features", "rows") if k in results)
        if head == "features_pipelined" and not (results[head]["valid"] and
                                                 results[head]["elapsed"] <= results["features"]["elapsed"]):
            head = "features"                          # the pipelined exchange must be correct AND faster to be the headline
        res = results[head]
        kind = "features" if head.startswith("features") else head
        total = res["total_entries"]
        achieved = res["fwd_bytes"] / (res["fwd_ms"] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(res["fwd_bytes"], args.skew) if world == 1 else (None, "N > 1: not collected")
        workload = ((f"[value = scheme '{kind}'] " if world > 1 else "") +
                    f"1 aggregation (GAT) layer, D={d}: SpMM forward + transpose-SpMM backward"
                    + {"none": "", "rows": " + RCCL all-reduce of the entity-gradient table",
                       "features": " + RCCL exchange (column slab <-> row block) both ways"}[kind]
                    + f"; synthetic KG {n_glob} entities / {total} stored (h,t) entries "
                      f"({e_loc * world} triples, R=16, {args.skew} heads)")
        out = {
            "metric": "kg_edges_aggregated_per_sec",
            "value": 2 * total * args.steps / res["elapsed"],
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.rehearse_on_one_gpu else "REHEARSAL (gloo, one GPU): not a measurement",
            "config": {
                "workload": workload,
                "entities": n_glob, "stored_entries": total, "triples": e_loc * world, "dim": d,
                "edge_aggregations_per_step": 2 * total, "passes": ["spmm_csr_fwd", "spmm_csc_bwd"],
                "sharding": {"none": "none",
                             "rows": f"head-row ranges x{world}, replicated table, gradient all-reduce",
                             "features": f"feature columns x{world} (D/G={d // world}), replicated structure, "
                                         f"column slab <-> row block exchange"}[kind],
                "exchange": res["exchange"],
                "skew": args.skew,
                "host_graph_build_s": round(t_build, 2),
                "spot_check": "ok" if res["valid"] else "MISMATCH: 8 output rows per rank differ from a direct evaluation",
                "spmm_only_ms_per_step": res.get("spmm_only_ms"),
                "spmm_only_edges_per_s": (2 * total / res["spmm_only_ms"] * 1e3) if res.get("spmm_only_ms") else None,
            },
            "roofline": {"bound": "hbm",
                         "kernel": "spmm_csr_kernel (forward launch)" if kind != "features" else
                                   "spmm_csr_kernel (forward launch over this rank's column slab, timed in the "
                                   "exchange-free loop)",
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "spmm_source_sha16": source_sha(),
                         "algorithmic_bytes_per_launch": res["fwd_bytes"], "avg_launch_ms": res["fwd_ms"],
                         "bwd_launch_ms": res["bwd_ms"],
                         "bwd_achieved_GBs": res["bwd_bytes"] / (res["bwd_ms"] * 1e-3) / 1e9
                         if (kind != "rows") else None,
                         "step_bwd_ms_including_exchange": res.get("step_bwd_ms"),
                         "exchange_ms": res.get("exchange_ms"),
                         "note": "achieved = algorithmic bytes / HIP-event time of one lkg_spmm_csr_f32 call (one "
                                 "launch covering its 128-column slabs); it can exceed the ~6.3 TB/s of a plain HBM copy "
                                 "because slabs of the source table are partly served from the 256 MiB Infinity Cache, "
                                 "whose hits the fabric-side FETCH_SIZE counter (traffic) still counts"},
        }
        if kind == "features":      # both exchange forms of the headline scheme, whichever is the headline
            out["features_exchanges"] = {
                k: {"value": 2 * v["total_entries"] * args.steps / v["elapsed"], "ms_per_step": v["elapsed"] / args.steps * 1e3,
                    "exchange": v["exchange"], "spot_check": "ok" if v["valid"] else "MISMATCH",
                    "step_bwd_ms_including_exchange": v.get("step_bwd_ms"), "exchange_ms": v.get("exchange_ms")}
                for k, v in results.items() if k.startswith("features")}
            if "pipelined_exchange_error" in results["features"]:
                out["features_exchanges"]["pipelined_exchange_error"] = results["features"]["pipelined_exchange_error"]
        if world > 1:
            # both schemes at the top level: `value` is the scheme named in config.value_scheme; the north star's edge-range
            # sharding + gradient all-reduce (BASELINE.json configs[3]) is `value_rows`, the column-sharded scheme `value_features`
            for k_, name_ in (("rows", "value_rows"), (head if head.startswith("features") else "features", "value_features")):
                if k_ in results:
                    out[name_] = 2 * results[k_]["total_entries"] * args.steps / results[k_]["elapsed"]
            out["config"]["value_scheme"] = kind
            # what a scaling curve over N compares: `value` at N > 1 is scheme `kind`; the N = 1 line runs the same two SpMM
            # passes per rank without any exchange, so value(N) / (N x value(1)) is that scheme's weak-scaling efficiency --
            # `value_rows` beside it is the north star's edge-range scheme against the same N = 1 number
            out["scaling_basis"] = kind
            out["config"]["sharding_rows"] = f"head-row ranges x{world}, replicated table, RCCL all-reduce of the entity-gradient table"
            out["config"]["sharding_features"] = (f"feature columns x{world} (D/G={d // world}), replicated structure, column slab "
                                                  f"<-> row block exchange pipelined with the SpMM")
            out["n_ranks_seen"] = dist.get_world_size()
            try:
                out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as exc:   # noqa: BLE001 -- informational only
                out["rccl_version"] = f"unavailable ({type(exc).__name__})"
            out["collective_backend"] = dist.get_backend()
        if "rows" in results and kind != "rows":      # the north star's scheme, same graph shape, next to the headline
            rr = results["rows"]
            out["rows_scheme"] = {
                "sharding": f"head-row ranges x{world}, replicated table, gradient all-reduce",
                "value": 2 * rr["total_entries"] * args.steps / rr["elapsed"], "unit": "edges/s",
                "ms_per_step": rr["elapsed"] / args.steps * 1e3, "steps": args.steps,
                "fwd_launch_ms": rr["fwd_ms"], "bwd_ms_including_all_reduce": rr["bwd_ms"],
                "fwd_frac_of_hbm_roofline": rr["fwd_bytes"] / (rr["fwd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "exchange": rr["exchange"],
                "spot_check": "ok" if rr["valid"] else "MISMATCH"}
        if state["sharded_step"] is not None:
            out["sharded_pre_training_step"] = state["sharded_step"]
        if note:
            out["note"] = note
        return out

    results, state = {}, {"sharded_step": None}
    watchdog = Watchdog(rank, assemble)
    for scheme in schemes:
        if scheme == "features":
            run_features()                             # fills results["features"] (+ "features_pipelined")
        else:
            if world > 1:
                watchdog.arm(300, "rows scheme (chunked all-reduce behind the transpose SpMM)")
            results[scheme] = run_rows()
            watchdog.disarm()
        torch.cuda.empty_cache()

    if world > 1 and not args.no_sharded_step:
        # after the headline measurements; an ordinary error here is reported in the JSON and does not cost the run
        # (every rank takes the same branch: the agreement below is a collective), a hang is the watchdog's
        watchdog.arm(420, "integrated sharded pre_training step")
        ok, why, got = 1.0, "", None
        try:
            got = sharded_step_timings(world, rank, dev, n_glob, d, h, t, r, max(3, min(args.steps, 10)))
        except LkgError as exc:
            ok, why = 0.0, f"LkgError: {exc}"
        except RuntimeError as exc:
            ok, why = 0.0, f"{type(exc).__name__}: {exc}"
        flag = torch.tensor([ok], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        state["sharded_step"] = got if float(flag) == 1.0 else {"error": (why or "failed on another rank")[:300]}
        watchdog.disarm()

    if rank == 0:
        out = assemble()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g_own, results["none"]["val"], n_glob, d, 2022)
        if world == 1 and not args.no_extra:
            results.clear()
            torch.cuda.empty_cache()
            c4 = c4_regime_timing(d, dev)          # (first: its 5 GB tables want fresh, unfragmented device memory)
            # the headline's 1 GB table is half-resident in the 256 MiB Infinity Cache; the regime an 8-GPU row-range run
            # executes is the cache-free one: both fractions and this box's plain-copy rate ride in `roofline`
            out["roofline"].update({"no_cache_frac_fwd": c4["fwd_frac_of_hbm_roofline"],
                                    "no_cache_frac_bwd": c4["bwd_frac_of_hbm_roofline"],
                                    "no_cache_config": c4["config"], "plain_copy_GBs": c4["plain_copy_GBs"],
                                    "plain_copy_frac": c4["plain_copy_GBs"] / HBM_PEAK_GBS})
            out["extra"] = whole_path_timings(h, t, r, n_glob, d, dev)
            out["extra"]["c4_regime_spmm"] = c4
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        # nothing this process queued may still be running when its tensors are torn down (the round-3 fault)
        try:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
        except Exception:   # noqa: BLE001
            pass
        raise
