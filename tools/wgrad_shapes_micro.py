#!/usr/bin/env python3
"""lkg_gemm_wgrad_f32 (f16 x 2 long-k engine) by shape: the gate's weight gradients of C3 (512 x 558) and of the reference's
default architecture (600 x 300 twice), and padded / aligned variants of them (GPU box).
    python tools/wgrad_shapes_micro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ge.build()
from literalkg_amd import ops

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


for m, k, lda, ldb in ((512, 558, 512, 558), (512, 256, 512, 256), (600, 300, 600, 300), (600, 300, 640, 320), (640, 320, 640, 320),
                       (600, 300, 600, 602), (600, 602, 600, 602), (512, 384, 512, 384), (768, 384, 768, 384), (256, 128, 256, 128)):
    g = torch.randn(n, lda, device=dev)[:, :m]
    x = torch.randn(n, ldb, device=dev)[:, :k]
    ca, cb = ops.col_absmax(g), ops.col_absmax(x)
    ms = timed(lambda: ops.gemm_wgrad(g, x, ca, cb))
    tm_, tn_ = -(-m // 256), -(-k // 128)
    print(f"dW[{m:4d} x {k:4d}] (row strides {lda}, {ldb}): {ms:7.3f} ms  useful {2.0 * n * m * k * 3 / ms / 1e9:7.0f} TF/s of fp16 MFMA, "
          f"padded ({tm_} x {tn_} tiles) {2.0 * n * tm_ * 256 * tn_ * 128 * 3 / ms / 1e9:7.0f}; operands read once = "
          f"{4.0 * n * (m + k) / ms / 1e6:6.0f} GB/s")
    del g, x
