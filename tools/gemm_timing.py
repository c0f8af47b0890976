#!/usr/bin/env python3
"""Per-workgroup cycle stamps of the split GEMM (build with LKG_EXTRA_HIPCC_FLAGS=-DLKG_TIMING).  Debug only."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from literalkg_amd import ops, _native as N
dev = torch.device("cuda:0")
a = torch.randn(1_000_000, 256, device=dev); b = torch.randn(256, 256, device=dev)
for _ in range(3): ops.gemm(a, b, trans_b=True)
torch.cuda.synchronize()
lib = N.load()
n = 15626
buf = np.zeros(4 * n, np.uint64)
lib.lkg_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.lkg_debug_timing(buf.ctypes.data, 4 * n)
t = buf.reshape(n, 4)
t0 = t[:, 0].min()
start, loop, end = (t[:, 0] - t0).astype(np.int64), (t[:, 1] - t0).astype(np.int64), (t[:, 2] - t0).astype(np.int64)
print("rc", rc, "kernel span (cycles of the 100 MHz? counter)", end.max())
print("per-block: total median", np.median(end - start), " main loop median", np.median(loop - start), " epilogue median", np.median(end - loop))
print("first 8 blocks start/loop/end:", [(int(start[i]), int(loop[i]), int(end[i])) for i in range(8)])
order = np.argsort(start)
print("start times percentiles:", np.percentile(start, [0, 10, 50, 90, 100]).astype(int))
print("blocks in flight at mid-kernel:", int(((start < end.max() // 2) & (end > end.max() // 2)).sum()))
