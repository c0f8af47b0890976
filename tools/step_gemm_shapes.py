#!/usr/bin/env python3
"""Which dense products does one pre_training step of a configuration launch, on which engine?  (GPU box; counts per shape)
    python tools/step_gemm_shapes.py --config default"""
import collections
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import literalkg_amd as L
from literalkg_amd import ops

seen = collections.Counter()
real_gemm, real_tall, real_call = ops.gemm, ops.gemm_tall, ops.N.call


def gemm(a, b, trans_a=False, trans_b=False, **kw):
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    seen[("gemm", f"ta={int(trans_a)} tb={int(trans_b)}", m, n, k)] += 1
    return real_gemm(a, b, trans_a=trans_a, trans_b=trans_b, **kw)


def gemm_tall(a_panels, b_blocks, trans_b, *args, **kw):
    rows = b_blocks[0][0].shape[0] if trans_b else b_blocks[0][0].shape[1]
    seen[("gemm_tall", f"groups={len(b_blocks)} tb={int(trans_b)}", a_panels[0].shape[0], rows * len(b_blocks),
          tuple(a.shape[1] for a in a_panels))] += 1
    return real_tall(a_panels, b_blocks, trans_b, *args, **kw)


def call(name, *a):
    if name in ("lkg_gemm_longk_f32", "lkg_gemm_smallm_f32", "lkg_gemm_skinny_f32", "lkg_gemm_wgrad_f32"):
        seen[(name, "", a[0], a[1], a[2])] += 1
    return real_call(name, *a)


ops.gemm, ops.gemm_tall, ops.N.call = gemm, gemm_tall, call
sys.argv = [sys.argv[0]] + sys.argv[1:] + ["--iters", "1"]
exec(compile(open(os.path.join(os.path.dirname(__file__), "step_profile.py")).read(), "step_profile.py", "exec"))
tot = sum(seen.values())
print(f"\n{tot} dense products over the script's passes (3 warm-up + timed forwards / steps); by shape:")
for (eng, flags, m, n, k), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"  {c:4d} x {eng:22s} {flags:18s} m={m} n={n} k={k}")
