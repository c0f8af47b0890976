#!/usr/bin/env python3
"""Where the wave-specialised tall GEMM's waves spend their cycles (DIAGNOSTIC build: rebuilds the library with
-DLKG_WS_STAMPS, runs one 1 M x K x 256 product per K, prints cycles per k step and wave kind: waiting for memory, waiting at
the step barrier, working; rebuild WITHOUT the flag before anything else uses the library).
    LKG_EXTRA_HIPCC_FLAGS=-DLKG_WS_STAMPS python tools/ws_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "LKG_WS_STAMPS" in os.environ.get("LKG_EXTRA_HIPCC_FLAGS", ""), "run with LKG_EXTRA_HIPCC_FLAGS=-DLKG_WS_STAMPS"
from literalkg_amd.build import build

build(force=False, verbose=False)      # (the flag is part of the object names: a library built without it is re-made)
from literalkg_amd import ops

dev = torch.device("cuda:0")
n, d = 1_000_000, 256
for k in (64, 256, 558):
    x = torch.randn(n, k, device=dev)
    w = torch.randn(d, k, device=dev) * 0.06
    out = torch.empty(n, d, device=dev)
    rm = ops.row_absmax(x)
    dbg = torch.zeros((n, d), device=dev)
    for _ in range(2):
        ops.gemm_tall((x,), ((w,),), True, None, out=out, rowmax=rm, variant="ws")
    torch.cuda.synchronize()
    ops.gemm_tall((x,), ((w,),), True, None, out=out, rowmax=rm, variant="ws", keep=(None, dbg))
    torch.cuda.synchronize()
    v = dbg.view(-1)[:32].view(torch.int64).cpu().tolist()
    c_items, l_items = max(v[4], 1), max(v[12], 1)      # summed over waves: items x waves
    print(f"K={k}: per wave and k step, cycles (s_memtime):")
    print(f"   compute waves: wait for B {v[0] / c_items:8.0f}  barrier {v[1] / c_items:8.0f}  step (reads + 24 MFMAs) {v[2] / c_items:8.0f}  "
          f"epilogue per step {v[3] / c_items:8.0f} (per tile {v[3] / c_items * ((k + 15) // 16):8.0f})")
    print(f"   loader waves : barrier {v[8] / l_items:8.0f}  issue {v[9] / l_items:8.0f}  wait for A {v[10] / l_items:8.0f}  stage {v[11] / l_items:8.0f}")
    for variant in ("256x1", "256x1w", "256r"):
        dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_tall((x,), ((w,),), True, None, out=out, rowmax=rm, variant=variant, keep=(None, dbg))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        v = dbg.view(-1)[:64].view(torch.int64).cpu().tolist()
        ns = max(v[22], 1)
        kt = (k + 15) // 16
        print(f"   {variant:7s} per wave and k step: wait (vmcnt + lgkmcnt) {v[16] / ns:7.0f}  barrier {v[17] / ns:7.0f}  ring read + requests {v[18] / ns:7.0f}  "
              f"MFMAs + split {v[19] / ns:7.0f} | per tile: epilogue {v[20] / ns * kt:8.0f}  tile opening {v[21] / ns * kt:8.0f}"
              f" | of the epilogue: scale + bias {v[23] / ns * kt:7.0f}  LDS writes {v[24] / ns * kt:7.0f}  LDS reads + stores {v[25] / ns * kt:7.0f}"
              f" | k loop end -> epilogue {v[26] / ns * kt:7.0f}  barrier behind the epilogue {v[27] / ns * kt:7.0f}"
              f" | whole kernel per wave {v[28] / max(v[29], 1):9.0f} cycles, stamped {sum(v[16:22]) / max(v[29], 1) + (v[26] + v[27]) / max(v[29], 1):9.0f}; launch {ms:.3f} ms (with the planes pass)"
              f" | in-kernel clock {v[28] / max(v[30], 1) * 0.1:.2f} GHz (s_memtime / s_memrealtime)")
