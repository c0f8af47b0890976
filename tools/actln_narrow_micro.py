#!/usr/bin/env python3
"""act_ln forward / backward at narrow widths (several rows per wave) against the bytes they move (GPU box only; tuning aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from literalkg_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
n = 1_000_000
for d in (32, 64, 128, 256):
    gb = n * d * 4 / 1e9
    z = torch.randn(n, d, device=dev); gamma = torch.randn(d, device=dev); beta = torch.randn(d, device=dev)
    for drop in (0.0, 0.1):
        t = timeit(lambda: ops._ActLayerNorm.apply(z, gamma, beta, True, 0.01, 1e-5, 1e-12, drop, 7, None, True))
        t2 = timeit(lambda: ops._ActLayerNorm.apply(z, gamma, beta, True, 0.01, 1e-5, 1e-12, drop, 7, None, False))
        print(f"d={d:4d} drop={drop}: fwd y+yn {t:.3f} ms {3*gb/t*1e3:.0f} GB/s | yn only {t2:.3f} ms {2*gb/t2*1e3:.0f} GB/s")
    zg = z.clone().requires_grad_(True)
    y, yn = ops.act_layernorm(zg, gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), want_norm=True, drop_p=0.1, seed=5)
    gy, gyn = torch.randn(n, d, device=dev), torch.randn(n, d, device=dev)
    t = timeit(lambda: torch.autograd.grad([y, yn], zg, [gy, gyn], retain_graph=True))
    print(f"d={d:4d} bwd (g_y + g_yn, dropout): {t:.3f} ms {6*gb/t*1e3:.0f} GB/s")
