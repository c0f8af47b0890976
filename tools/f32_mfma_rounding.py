#!/usr/bin/env python3
"""Does the f32 MFMA engine (v_mfma_f32_32x32x2_f32 chains) round its accumulations to nearest?  Sums of k POSITIVE products
through lkg_grouped_gemm_f32's k mode (one group) against float64: a round-to-nearest chain errs like sqrt(k) eps with either
sign, a truncating one like -k eps / 2 (GPU box).   python tools/f32_mfma_rounding.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from literalkg_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for k in (256, 2048, 16384, 131072):
    x = torch.rand(k, 64, device=dev) + 0.5
    g = torch.rand(k, 8, device=dev) + 0.5
    out = torch.empty(1, 64, 8, device=dev)
    seg = torch.tensor([0, k], dtype=torch.int32, device=dev)
    ops._grouped(2, seg, k, x, g, out, 64, 8, 0, True, False, 0.0, stride_c=64 * 8)
    want = x.double().t() @ g.double()
    rel = (out[0].double() - want) / want
    cpu = (x.cpu().t() @ g.cpu()).double()
    rel_cpu = (cpu - want.cpu()) / want.cpu()
    print(f"k = {k:7d}: device mean relative error {float(rel.mean()):+.3e}  max |.| {float(rel.abs().max()):.3e}   (k eps / 2 = {k * 2 ** -24 / 2:.3e}; "
          f"torch CPU f32 matmul: mean {float(rel_cpu.mean()):+.3e} max {float(rel_cpu.abs().max()):.3e})")
