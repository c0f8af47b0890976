#!/usr/bin/env python3
"""Times of the dense products a training step of the reference's default architecture launches (shapes from
tools/step_gemm_shapes.py), each through the library's own routing (ops.gemm / ops.linear), 1 M rows (GPU box).
    python tools/product_times.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ge.build()
from literalkg_amd import ops
from literalkg_amd.transport import install_drain_excepthook

install_drain_excepthook()
dev = torch.device("cuda:0")
n = 1_000_000


def timed(fn, reps=7):
    fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


cases = [("x W^T   1M x 300 -> 32", (n, 300), (32, 300), False, True),
         ("x W^T   1M x 556 -> 300", (n, 556), (300, 556), False, True),
         ("x W^T   1M x 32 -> 32", (n, 32), (32, 32), False, True),
         ("g W     1M x 32 -> 32", (n, 32), (32, 32), False, False),
         ("g W     1M x 32 -> 300", (n, 32), (32, 300), False, False),
         ("g W     1M x 300 -> 556", (n, 300), (300, 556), False, False),
         ("g^T x   32 x 1M x 32", (n, 32), (n, 32), True, False),
         ("g^T x   32 x 1M x 300", (n, 32), (n, 300), True, False),
         ("g^T x   300 x 1M x 556", (n, 300), (n, 556), True, False)]
for name, sa, sb, ta, tb in cases:
    a = torch.randn(sa, device=dev)
    b = torch.randn(sb, device=dev) * 0.05
    ms = timed(lambda: ops.gemm(a, b, ta, tb))
    m_ = sa[1] if ta else sa[0]
    k_ = sa[0] if ta else sa[1]
    n_ = sb[0] if tb else sb[1]
    byt = 4.0 * (a.numel() + b.numel() + m_ * n_)
    print(f"{name:28s} {ms:8.3f} ms   {2.0 * m_ * n_ * k_ / ms / 1e9:8.1f} TFLOP/s (f32)   {byt / ms / 1e6:8.0f} GB/s algorithmic")
    del a, b
