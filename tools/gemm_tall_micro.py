#!/usr/bin/env python3
"""Tall GEMM engines side by side on the path's shapes (GPU box): f16 x 2 (lkg_gemm_tall_f32) vs bf16 x 3 (round 1)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd import ops
dev = torch.device("cuda:0")
n, d = 1_000_000, 256
def tm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
x = torch.randn(n, d, device=dev); gy = torch.randn(n, d, device=dev)
w = torch.randn(d, d, device=dev) * 0.06; b = torch.randn(d, device=dev)
out = torch.empty(n, d, device=dev)
rm = ops.row_absmax(x)
print(f"row_absmax {n}x{d}: {tm(lambda: ops.row_absmax(x, rm)):.3f} ms")
for eng in ("f16x2-all", "bf16x3"):
    ops._ENGINE = eng
    f = tm(lambda: ops.gemm(x, w, trans_b=True, bias=b, out=out))
    g = tm(lambda: ops.gemm(gy, w, out=out))
    print(f"{eng}: Linear fwd {f:.3f} ms ({2*n*d*d/f/1e9:.0f} TF f32-eq) | dgrad {g:.3f} ms ({2*n*d*d/g/1e9:.0f} TF)")
ops._ENGINE = "f16x2"
f = tm(lambda: ops.gemm_tall((x,), ((w,),), True, b, out=out, rowmax=rm))
print(f"f16x2 fwd with a given row scale: {f:.3f} ms ({2*n*d*d/f/1e9:.0f} TF; x read + y write = {2*n*d*4/f/1e6:.0f} GB/s)")
gate = L.GateMul(d, 2, 300).to(dev)
num, txt = torch.rand(n, 2, device=dev), torch.randn(n, 300, device=dev)
xx = x * 0.05
with torch.no_grad():
    for eng in ("f16x2", "bf16x3"):
        ops._ENGINE = eng
        t = tm(lambda: gate(xx, num, txt, out), 10)
        fl = 2.0 * n * (d + 302) * d * 2
        print(f"{eng}: GateMul forward {t:.3f} ms ({fl/t/1e9:.0f} TF f32-eq, algorithmic {4.0*n*(2*d+302)/t/1e6:.0f} GB/s)")
ops._ENGINE = "f16x2"
xg = xx.clone().requires_grad_(True)
def fb():
    gate.zero_grad(set_to_none=True); xg.grad = None
    gate(xg, num, txt).backward(gy)
for eng in ("f16x2", "bf16x3"):
    ops._ENGINE = eng
    print(f"{eng}: GateMul forward+backward {tm(fb, 5):.3f} ms")
