#!/usr/bin/env python3
"""lkg_gemm_f32 on other widths (512, 64, 32, 128, 1024) against torch.  GPU box only; not part of the tests."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
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
for N, d in ((1_000_000, 512), (1_000_000, 64), (1_000_000, 32), (2_000_000, 128), (500_000, 1024)):
    x = torch.randn(N, d, device=dev); w = torch.randn(d, d, device=dev); gy = torch.randn(N, d, device=dev)
    for name, fn, ref in (("fwd  ", lambda: ops.gemm(x, w, trans_b=True), lambda: torch.matmul(x, w.t())),
                          ("dgrad", lambda: ops.gemm(gy, w), lambda: torch.matmul(gy, w)),
                          ("wgrad", lambda: ops.gemm(gy, x, trans_a=True), lambda: torch.matmul(gy.t(), x))):
        ms, rs = timeit(fn), timeit(ref)
        fl = 2.0 * N * d * d
        floor = (2 * N * d * 4) / 6.0e9 if name != "wgrad" else (2 * N * d * 4) / 6.0e9
        print(f"N={N} d={d} {name}: {ms:7.3f} ms {fl/ms/1e9:6.1f} TF/s | torch {rs:7.3f} ms {fl/rs/1e9:6.1f} TF/s | memory floor ~{floor:.3f} ms", flush=True)
    del x, w, gy
