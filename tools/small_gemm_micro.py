#!/usr/bin/env python3
"""Small-m GEMM calls: GPU time vs host-side issue time.  GPU box only; not part of the tests."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from literalkg_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
import time
for m in (256, 500, 512, 1024, 2049, 4096, 8192, 16384, 65536):
    x = torch.randn(m, 256, device=dev); w = torch.randn(256, 256, device=dev)
    t = timeit(lambda: ops.gemm(x, w, trans_b=True))
    t0 = time.perf_counter()
    for _ in range(200): ops.gemm(x, w, trans_b=True)
    host = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize()
    tt = timeit(lambda: torch.matmul(x, w.t()))
    print(f"m={m:6d}: lkg {t:8.1f} us (host issue {host:6.1f} us/call)  torch {tt:8.1f} us")
