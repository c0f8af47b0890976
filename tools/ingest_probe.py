#!/usr/bin/env python3
"""Ingestion at BASELINE scale (dataloader.py:186-190, 449-495): a generated "h r t" text file of --e triples
is parsed (lkg_triples_count / _read, mmap'd, threaded), de-duplicated (lkg_triples_dedup), turned into the device structure
(radix-sort build + transpose) and into the loader's initial A_in (device Laplacian).  GPU box only; prints triples/s per stage.
    python tools/ingest_probe.py [--n 5000000] [--e 100000000] [--json out.json]"""
import argparse, json, os, subprocess, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import literalkg_amd as L
from literalkg_amd import io
from literalkg_amd.synth import make_kg_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=5_000_000)
ap.add_argument("--e", type=int, default=100_000_000)
ap.add_argument("--writers", type=int, default=16)
ap.add_argument("--json", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
h, t, r = (x.cpu().numpy() for x in make_kg_device(args.n, args.e, "zipf", 2022, dev))
tmp = tempfile.mkdtemp(prefix="lkg_ingest_")
path = os.path.join(tmp, "kg_final.txt")


def write_chunk(i, lo, hi):          # (plain "%d %d %d\n" lines, the format dataloader.py:186 reads)
    import pandas as pd
    pd.DataFrame({"h": h[lo:hi], "r": r[lo:hi], "t": t[lo:hi]}).to_csv(f"{path}.{i:03d}", sep=" ", header=False, index=False)


t0 = time.perf_counter()
from concurrent.futures import ThreadPoolExecutor
cuts = np.linspace(0, args.e, args.writers + 1).astype(np.int64)
import multiprocessing as mp
procs = []
ctx = mp.get_context("fork")
for i in range(args.writers):
    p = ctx.Process(target=write_chunk, args=(i, int(cuts[i]), int(cuts[i + 1])))
    p.start()
    procs.append(p)
for p in procs:
    p.join()
    assert p.exitcode == 0
with open(path, "wb") as out:
    for i in range(args.writers):
        with open(f"{path}.{i:03d}", "rb") as f:
            while True:
                buf = f.read(1 << 26)
                if not buf:
                    break
                out.write(buf)
        os.remove(f"{path}.{i:03d}")
t_write = time.perf_counter() - t0
size = os.path.getsize(path)
del h, t, r
res = {"entities": args.n, "triples_in_file": args.e, "file_bytes": size, "file_write_s (not part of ingestion)": round(t_write, 2)}

t0 = time.perf_counter()
hh, rr, tt = io.load_triples(path)
t_load = time.perf_counter() - t0
res["load_triples_s"] = t_load
res["load_triples_per_s"] = args.e / t_load
res["load_triples_MB_per_s"] = size / t_load / 1e6
res["triples_after_dedup"] = int(len(hh))
torch.cuda.synchronize()
t0 = time.perf_counter()
hd, td, rd = (torch.from_numpy(x).to(dev) for x in (hh, tt, rr))
torch.cuda.synchronize()
t_up = time.perf_counter() - t0
res["upload_s"] = t_up
t0 = time.perf_counter()
g = L.KGStructure.from_triples(args.n, hd, td, rd, device=dev)
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
res["device_structure_build_s"] = t_build
res["stored_entries"] = g.nnz
t0 = time.perf_counter()
val = io.laplacian_values(g, "random-walk")
torch.cuda.synchronize()
t_lap = time.perf_counter() - t0
res["device_laplacian_s"] = t_lap
total = t_load + t_up + t_build + t_lap
res["total_ingestion_s"] = total
res["total_triples_per_s"] = args.e / total
res["row_sum_check"] = float(val.sum()) / max(1, int((g.rowptr[1:] > g.rowptr[:-1]).sum()))
for k, v in res.items():
    print(f"{k}: {v}")
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
os.remove(path)
os.rmdir(tmp)
