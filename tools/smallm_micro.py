#!/usr/bin/env python3
"""lkg_gemm_smallm_f32 (narrow-dY weight gradients) over its slice count (GPU box only; tuning aid: run with LKG_SMALLM_BLOCKS=n)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from literalkg_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
k = 1_000_000
for m, n in ((32, 32), (32, 300), (64, 64)):
    a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
    t = timeit(lambda: ops.gemm(a, b, trans_a=True))
    print(f"blocks={os.environ.get('LKG_SMALLM_BLOCKS', 'default')}: dW[{m} x {n}] over {k} rows {t:.3f} ms  ({(m + n) * k * 4 / t / 1e6:.0f} GB/s of operands)")
