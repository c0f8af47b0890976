#!/usr/bin/env python3
"""The clock the chip holds inside lkg_gemm_tall_f32's default tiling (MI355X_MICROARCH.md 'DVFS give-back' item 6): a build whose
ONLY stamps are s_memtime / s_memrealtime at kernel entry and exit, ~2 s of back-to-back launches on random data, then one
stamped launch per shape (GPU box; rebuild without the flag before anything else uses the library).
    LKG_EXTRA_HIPCC_FLAGS=-DLKG_CLOCK_STAMP python tools/tall_clock.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "LKG_CLOCK_STAMP" in os.environ.get("LKG_EXTRA_HIPCC_FLAGS", ""), "run with LKG_EXTRA_HIPCC_FLAGS=-DLKG_CLOCK_STAMP"
from literalkg_amd.build import build

build(force=False, verbose=False)
from literalkg_amd import ops

dev = torch.device("cuda:0")
n = 1_000_000
for k, d in ((256, 256), (64, 256), (558, 256)):
    x = torch.randn(n, k, device=dev)
    w = torch.randn(d, k, device=dev) * 0.06
    out = torch.empty(n, d, device=dev)
    rm = ops.row_absmax(x)
    dbg = torch.zeros((n, d), device=dev)
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(50):
            ops.gemm_tall((x,), ((w,),), True, None, out=out, rowmax=rm, variant="256x1")
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_tall((x,), ((w,),), True, None, out=out, rowmax=rm, variant="256x1", keep=(None, dbg))
    e1.record()
    torch.cuda.synchronize()
    v = dbg.view(-1)[:64].view(torch.int64).cpu().tolist()
    print(f"1 M x {k} x {d}: in-kernel clock {v[28] / max(v[30], 1) * 0.1:.2f} GHz over {v[29]} waves "
          f"({v[28] / max(v[29], 1):.0f} cycles per wave), launch {e0.elapsed_time(e1):.3f} ms incl. the planes pass; "
          f"3-MFMA product at THIS clock: {2.0 * n * k * d * 3 / (256 * 4 * 1024 * (v[28] / max(v[30], 1) * 1e8)) * 1e3:.3f} ms at a full matrix pipe")
