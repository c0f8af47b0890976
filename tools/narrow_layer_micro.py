#!/usr/bin/env python3
"""A 32 -> 32 aggregation layer's dense backward, fused (lkg_narrow_layer_bwd_f32) against the unfused passes, 1 M rows (GPU box):
forward + backward times of ops.narrow_layer vs ops.linear + ops.act_layernorm, and the backward alone by subtraction.
    python tools/narrow_layer_micro.py [--n 1000000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ge.build()
from literalkg_amd import ops
from literalkg_amd.transport import install_drain_excepthook

install_drain_excepthook()
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
args = ap.parse_args()
dev = torch.device("cuda:0")
n = args.n
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, 32, generator=gen, device=dev).requires_grad_(True)
w = (torch.randn(32, 32, generator=gen, device=dev) * 0.2).requires_grad_(True)
b = torch.zeros(32, device=dev, requires_grad=True)
gamma = torch.ones(32, device=dev, requires_grad=True)
beta = torch.zeros(32, device=dev, requires_grad=True)
gy = torch.randn(n, 32, generator=gen, device=dev)
gyn = torch.randn(n, 32, generator=gen, device=dev)


def timed(fn, reps=9):
    fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, c in evs:
        a.record()
        fn()
        c.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(c) for a, c in evs)
    return ts[len(ts) // 2]


def run(fused, grads, backward=True):
    for t in (x, w, b, gamma, beta):
        t.grad = None
    if fused:
        y, yn = ops.narrow_layer(x, w, b, gamma, beta, drop_p=0.1, seed=5)
    else:
        y, yn = ops.act_layernorm(ops.linear(x, w, b), gamma, beta, drop_p=0.1, seed=5)
    if backward:
        outs, gs = zip(*[(o, g) for o, g in ((y, grads[0]), (yn, grads[1])) if g is not None])
        torch.autograd.backward(outs, gs)


for label, grads in (("g_y and g_yn", (gy, gyn)), ("g_y only", (gy, None))):
    fwd = timed(lambda: run(False, grads, backward=False))
    for fused in (False, True):
        ms = timed(lambda: run(fused, grads))
        print(f"{label:14s} {'one launch' if fused else 'unfused   '}  forward + backward {ms:7.3f} ms   backward alone {ms - fwd:7.3f} ms")
