#!/usr/bin/env python3
"""lkg_gemm_f32 throughput on the shapes the hot path uses (GPU box only)."""
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
N = 1_000_000
shapes = [  # name, (a shape), (b shape), ta, tb
    ("(warm-up) x[N,256] @ W[256,256]^T", (N, 256), (256, 256), False, True),
    ("linear fwd  x[N,256] @ W[256,256]^T", (N, 256), (256, 256), False, True),
    ("linear fwd  x[N,128] @ W[128,128]^T", (N, 128), (128, 128), False, True),
    ("dgrad       gy[N,256] @ W[256,256]", (N, 256), (256, 256), False, False),
    ("wgrad       gy[N,256]^T @ x[N,256]", (N, 256), (N, 256), True, False),
    ("gate txt    t[N,300] @ W[256,300]^T", (N, 300), (256, 300), False, True),
    ("gate wgrad  gy[N,256]^T @ t[N,300]", (N, 256), (N, 300), True, False),
    ("square 4096", (4096, 4096), (4096, 4096), False, False),
    ("square 4096 NT", (4096, 4096), (4096, 4096), False, True),
]
import os as _os
for name, sa, sb, ta, tb in shapes[:int(_os.environ.get('GEMM_MICRO_SHAPES', '99'))]:
    a = torch.randn(sa, device=dev); b = torch.randn(sb, device=dev)
    m, k = (sa[1], sa[0]) if ta else sa
    n = sb[0] if tb else sb[1]
    ms = timeit(lambda: ops.gemm(a, b, ta, tb))
    ref = timeit(lambda: torch.matmul(a.t() if ta else a, b.t() if tb else b))
    fl = 2.0 * m * n * k
    # accuracy against f64 on a slice of the output (ours and torch's f32 result)
    if ta:
        a64, sl = a[:, :512].double().t(), (slice(0, 512), slice(None))
    else:
        a64, sl = a[:4096].double(), (slice(0, 4096), slice(None))
    want = a64 @ (b.double().t() if tb else b.double())
    scale = float(want.abs().max())
    e_own = float((ops.gemm(a, b, ta, tb)[sl].double() - want).abs().max()) / scale
    e_ref = float((torch.matmul(a.t() if ta else a, b.t() if tb else b)[sl].double() - want).abs().max()) / scale
    print(f"{name:40s} {ms:8.3f} ms {fl/ms/1e9:7.1f} TF/s (err {e_own:.1e}) | rocBLAS/hipBLASLt via torch {ref:8.3f} ms "
          f"{fl/ref/1e9:7.1f} TF/s (err {e_ref:.1e})", flush=True)
