#!/usr/bin/env python3
"""Where the ~12 us host-side cost of one op goes (GPU box only; not part of the tests)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.build()
from literalkg_amd import ops, _native as N
dev = torch.device("cuda:0")
x = torch.randn(256, 256, device=dev); w = torch.randn(256, 256, device=dev); out = torch.empty(256, 256, device=dev)
def t(fn, n=20000):
    for _ in range(100): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print(f"torch.cuda.current_stream().cuda_stream   {t(lambda: torch.cuda.current_stream().cuda_stream):6.2f} us")
print(f"ops._stream()                             {t(ops._stream):6.2f} us")
print(f"N.call('lkg_version')                     {t(lambda: N.load().lkg_version()):6.2f} us")
print(f"x.data_ptr()                              {t(x.data_ptr):6.2f} us")
print(f"torch.empty((256,256))                    {t(lambda: torch.empty((256, 256), dtype=torch.float32, device=dev)):6.2f} us")
print(f"ops._f32_rows(x)                          {t(lambda: ops._f32_rows(x)):6.2f} us")
print(f"ops.gemm(x, w, trans_b, out=out)          {t(lambda: ops.gemm(x, w, trans_b=True, out=out), 5000):6.2f} us")
print(f"ops.gemm(x, w, trans_b)                   {t(lambda: ops.gemm(x, w, trans_b=True), 5000):6.2f} us")
torch.cuda.synchronize()
print(f"torch.matmul(x, w.t())                    {t(lambda: torch.matmul(x, w.t()), 5000):6.2f} us")
torch.cuda.synchronize()
