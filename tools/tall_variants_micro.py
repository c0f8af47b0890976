#!/usr/bin/env python3
"""lkg_gemm_tall_f32's tilings side by side in ONE process, interleaved rounds (GPU box): the plain Linear product
1 M x 256 x 256 (+ 300-wide and K = 64 / 558 shapes) and the gate's stacked product, per variant: median / min of HIP-event
times, f32-equivalent TFLOP/s, algorithmic GB/s, and the error against float64 on a row sample.
    python tools/tall_variants_micro.py [--rounds 7] [--json out.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ge.build()
import literalkg_amd as L
from literalkg_amd import ops
from literalkg_amd.transport import install_drain_excepthook

install_drain_excepthook()
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--variants", default="256x1,256x1w,ws")
ap.add_argument("--json", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
n = args.n
variants = [v for v in args.variants.split(",") if v]


def ev_time(fn, reps):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


res = {}
gen = torch.Generator(device=dev).manual_seed(1)
for name, k, d_out in (("linear 256 -> 256", 256, 256), ("linear 64 -> 256", 64, 256), ("linear 300 -> 300", 300, 300),
                       ("linear 558 -> 256", 558, 256)):
    x = torch.randn((n, k), generator=gen, device=dev)
    w = torch.randn((d_out, k), generator=gen, device=dev) * 0.06
    b = torch.randn(d_out, generator=gen, device=dev)
    out = torch.empty((n, d_out), device=dev)
    rm = ops.row_absmax(x)
    sample = torch.randint(0, n, (2048,), generator=gen, device=dev)
    want = x[sample].double() @ w.double().t() + b.double()
    times = {v: [] for v in variants}
    errs = {}
    for v in variants:
        ops.gemm_tall((x,), ((w,),), True, b, out=out, rowmax=rm, variant=v)
        errs[v] = float((out[sample].double() - want).abs().max() / want.abs().max())
    for _ in range(args.rounds):
        for v in variants:
            times[v] += ev_time(lambda: ops.gemm_tall((x,), ((w,),), True, b, out=out, rowmax=rm, variant=v), args.reps)
    fl, by = 2.0 * n * k * d_out, 4.0 * n * (k + d_out)
    for v in variants:
        med, mn = float(np.median(times[v])), float(np.min(times[v]))
        res[f"{name} [{v}]"] = {"median_ms": med, "min_ms": mn, "tflops_f32_equivalent": fl / med / 1e9,
                                "algorithmic_GBs": by / med / 1e6, "hbm_frac": by / med / 1e6 / 8000, "max_rel_err_vs_f64": errs[v]}
        print(f"{name:20s} {v:7s} median {med:.3f} ms  min {mn:.3f} ms  {fl / med / 1e9:6.0f} TF  {by / med / 1e6:6.0f} GB/s "
              f"({by / med / 1e6 / 8000:.3f} of 8 TB/s)  err {errs[v]:.2e}", flush=True)
    del x, w, out

# the gate: 1 M x (256 + 2 + 300) -> 256, stacked g / z projections + blend epilogue
d = 256
gate = L.GateMul(d, 2, 300).to(dev)
x = torch.randn((n, d), generator=gen, device=dev) * 0.05
num, txt = torch.rand((n, 2), generator=gen, device=dev), torch.randn((n, 300), generator=gen, device=dev)
out = torch.empty((n, d), device=dev)
times = {v: [] for v in variants}
with torch.no_grad():
    ref = None
    for v in variants:
        ops.DEFAULT_TALL_VARIANT = v
        gate(x, num, txt, out)
        ref = out.clone() if ref is None else ref
        print(f"gate [{v}] max |diff| vs [{variants[0]}]: {float((out - ref).abs().max()):.2e}")
    for _ in range(args.rounds):
        for v in variants:
            ops.DEFAULT_TALL_VARIANT = v
            times[v] += ev_time(lambda: gate(x, num, txt, out), args.reps)
ops.DEFAULT_TALL_VARIANT = None
fl, by = 2.0 * n * (d + 302) * d * 2, 4.0 * n * (2 * d + 302)
for v in variants:
    med, mn = float(np.median(times[v])), float(np.min(times[v]))
    res[f"gate forward [{v}]"] = {"median_ms": med, "min_ms": mn, "tflops_f32_equivalent": fl / med / 1e9, "algorithmic_GBs": by / med / 1e6}
    print(f"gate forward         {v:7s} median {med:.3f} ms  min {mn:.3f} ms  {fl / med / 1e9:6.0f} TF  {by / med / 1e6:6.0f} GB/s", flush=True)
# K5 in one launch against the unfused pair (Linear on the default tiling, then act + LayerNorm + normalised copy)
for name, k, d_out in (("layer 256 -> 256", 256, 256), ("layer 300 -> 32", 300, 32), ("layer 64 -> 64", 64, 64)):
    x = torch.randn((n, k), generator=gen, device=dev)
    w = torch.randn((d_out, k), generator=gen, device=dev) * 0.06
    b = torch.randn(d_out, generator=gen, device=dev)
    gamma, beta = torch.ones(d_out, device=dev), torch.zeros(d_out, device=dev)
    rm = ops.row_absmax(x)
    slot = torch.empty((n, d_out), device=dev)
    z = torch.empty((n, d_out), device=dev)

    def unfused():
        if ops.tall_ok(n, d_out, (k,), True):
            ops.gemm_tall((x,), ((w,),), True, b, out=z, rowmax=rm)
        else:
            ops.gemm(x, w, trans_b=True, bias=b, out=z)
        return ops.act_layernorm(z, gamma, beta, want_norm=True, drop_p=0.1, seed=7, yn_out=slot)

    def fused():
        return ops.linear_act_layernorm_fwd((x,), (w,), b, gamma, beta, 0.01, 1e-5, 1e-12, 0.1, 7, yn_out=slot, rowmax=rm)
    y0, _ = unfused()
    y1 = fused()[0]
    print(f"{name}: max |fused - unfused| = {float((y1 - y0).abs().max()):.2e}")
    tu, tf = [], []
    with torch.no_grad():
        for _ in range(args.rounds):
            tu += ev_time(unfused, args.reps)
            tf += ev_time(fused, args.reps)
    for tag, tt in (("unfused pair", tu), ("fused launch", tf)):
        med = float(np.median(tt))
        res[f"{name} [{tag}]"] = {"median_ms": med, "min_ms": float(np.min(tt))}
        print(f"{name:20s} {tag:13s} median {med:.3f} ms  min {float(np.min(tt)):.3f} ms", flush=True)
    del x, w, slot, z
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
