#!/usr/bin/env python3
"""HBM efficiency of the row-wise kernels at the BASELINE shapes (GPU box only)."""
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
n, d = 1_000_000, 256
gb = n * d * 4 / 1e9
z = torch.randn(n, d, device=dev); gamma = torch.randn(d, device=dev); beta = torch.randn(d, device=dev)
cat = torch.randn(n, 2 * d, device=dev)
for drop in (0.0, 0.1):
    t = timeit(lambda: ops._ActLayerNorm.apply(z, gamma, beta, True, 0.01, 1e-5, 1e-12, drop, 7, None, True))
    print(f"act_ln fwd (y+yn, drop={drop})  {t:.3f} ms  {3*gb/t*1e3:.0f} GB/s")
zg = z.clone().requires_grad_(True)
y, yn = ops.act_layernorm(zg, gamma.requires_grad_(True), beta.requires_grad_(True), want_norm=True)
gyn = cat[:, d:]
t = timeit(lambda: torch.autograd.grad(yn, zg, gyn, retain_graph=True))
print(f"act_ln bwd (gyn strided)       {t:.3f} ms  {4*gb/t*1e3:.0f} GB/s")
t = timeit(lambda: torch.autograd.grad([y, yn], zg, [z, gyn], retain_graph=True))
print(f"act_ln bwd (gy + gyn)          {t:.3f} ms  {5*gb/t*1e3:.0f} GB/s")
t = timeit(lambda: ops.colsum(z))
print(f"colsum                         {t:.3f} ms  {gb/t*1e3:.0f} GB/s")
x = torch.randn(n, d, device=dev); gp = torch.randn(n, d, device=dev); zp = torch.randn(n, d, device=dev)
t = timeit(lambda: ops._GateBlend.apply(x, gp, zp, None))
print(f"gate blend fwd                 {t:.3f} ms  {4*gb/t*1e3:.0f} GB/s")
xg = x.clone().requires_grad_(True); gpg = gp.clone().requires_grad_(True); zpg = zp.clone().requires_grad_(True)
o = ops.gate_blend(xg, gpg, zpg)
t = timeit(lambda: torch.autograd.grad(o, [xg, gpg, zpg], z, retain_graph=True))
print(f"gate blend bwd                 {t:.3f} ms  {7*gb/t*1e3:.0f} GB/s")
t = timeit(lambda: torch.cat([x, gp], 1))
print(f"torch.cat of two N x D         {t:.3f} ms  {4*gb/t*1e3:.0f} GB/s")
p = torch.randn(n, d, device=dev); g = torch.randn(n, d, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
from literalkg_amd import _native as N
t = timeit(lambda: N.call("lkg_adam_step_f32", p.numel(), N.ptr(p), N.ptr(g), N.ptr(m), N.ptr(v), 1e-3, 0.9, 0.999, 1e-8, 0.0, 3, ops._stream()))
print(f"fused adam                     {t:.3f} ms  {7*gb/t*1e3:.0f} GB/s")
