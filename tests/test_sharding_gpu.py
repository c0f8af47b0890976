"""GPU: the two multi-GPU schemes of literalkg_amd/sharding.py with the REAL HIP SpMM, rehearsed as 2 ranks sharing
the one GPU of the test box over gloo (host-staged exchange).  Each rank checks its slab / row range against the
single-device result it computes itself.  A second test runs the same checks over RCCL with one GPU per rank; it is
skipped where fewer than two GPUs are visible (the driver's scaling run exercises RCCL in any case)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    nccl = backend == "nccl"
    if nccl:      # one GPU per rank, RCCL over xGMI: the exchange paths bench.py uses at N > 1
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import literalkg_amd as L
        from literalkg_amd import ops
        from literalkg_amd.sharding import FeatureShardedAggregation, ShardedAggregation, shard_bounds
        from literalkg_amd.synth import make_kg
        dev = torch.device("cuda", rank if nccl else 0)
        n, e, d = 30_000, 400_000, 128
        h, t, r = make_kg(n, e, seed=3)
        g = L.KGStructure.from_triples(n, h, t, r, device=dev)
        gen = torch.Generator(device=dev).manual_seed(1)
        ent = torch.randn((n, d), generator=gen, device=dev) * 0.2
        rel = torch.randn((16, d), generator=gen, device=dev) * 0.2
        gside = torch.randn((n, d), generator=gen, device=dev)
        val, _ = ops.edge_softmax(g, ent, rel)
        want_side = ops.spmm_raw(g.rowptr, g.col, val, ent, n, long_rows=g.long_rows(False))
        want_grad = ops.spmm_raw(g.t_rowptr, g.t_col, ops.permute_values(val, g.t_perm), gside, n,
                                 long_rows=g.long_rows(True))
        cuts = shard_bounds(g, world)
        lo, hi = cuts[rank], cuts[rank + 1]
        ok = True
        # --- feature sharding: slab forward (+ pipelined exchange), exchange back, slab backward
        fs = FeatureShardedAggregation(g, val, rank, world, d, cuts)
        cols = slice(rank * fs.dg, (rank + 1) * fs.dg)
        side_slab, block = fs.forward_to_row_block(fs.column_slab(ent))
        ok &= torch.allclose(side_slab, want_side[:, cols], rtol=1e-5, atol=1e-5)
        rows = torch.cat([block[i] for i in range(world)], dim=1)
        ok &= torch.allclose(rows, want_side[lo:hi], rtol=1e-5, atol=1e-5)
        ok &= torch.equal(fs.to_row_block(side_slab), block)
        gblock = torch.stack([gside[lo:hi, i * fs.dg:(i + 1) * fs.dg] for i in range(world)]).contiguous()
        gslab = fs.to_column_slab(gblock)
        ok &= torch.equal(gslab, gside[:, cols].contiguous())
        ok &= torch.allclose(fs.backward(gslab), want_grad[:, cols], rtol=1e-5, atol=1e-4)
        ok &= torch.allclose(fs.backward_from_row_block(gblock, pieces=2), want_grad[:, cols], rtol=1e-5, atol=1e-4)
        # --- row-range sharding: own rows forward, all-reduced transpose backward (attention computed per shard)
        keep = (h >= lo) & (h < hi)
        mine = L.KGStructure.from_triples(n, h[keep], t[keep], r[keep], device=dev)
        val_mine, _ = ops.edge_softmax(mine, ent, rel, row_lo=lo, row_hi=hi)
        rp = g.host("rowptr")
        ok &= torch.allclose(val_mine, val[rp[lo]:rp[hi]], rtol=1e-5, atol=1e-7)
        sh = ShardedAggregation(mine, val_mine, lo, hi, n_chunks=3)
        ok &= torch.allclose(sh.forward(ent), want_side[lo:hi], rtol=1e-5, atol=1e-5)

        grad = sh.backward(gside[lo:hi].contiguous())      # (the chunked all-reduce goes through transport.Transport)
        ok &= torch.allclose(grad, want_grad, rtol=1e-5, atol=1e-4)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu(gpu_device):
    import __graft_entry__ as ge
    ge.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)], res


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (the test boxes have one)")
def test_two_ranks_over_rccl(gpu_device):
    """The same checks with one GPU per rank and backend "nccl" (RCCL): all-to-all with uneven splits, batched
    point-to-point sends behind the SpMM, the async column-piece exchange of the backward, chunked all-reduce."""
    import __graft_entry__ as ge
    ge.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "nccl")) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)], res


@pytest.mark.timeout(600)
def test_bench_two_rank_rehearsal_starts_its_own_workers(gpu_device):
    """`python bench.py --gpus 2` without a torch.distributed environment starts its two workers itself (here both on
    the one GPU, over gloo: a rehearsal of the N > 1 code path, not a measurement) and prints ONE JSON line carrying
    both exchange forms of the feature scheme, the north star's row-range scheme and the integrated sharded step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu",
                          "--steps", "2", "--warmup", "1", "--entities", "40000", "--edges", "400000"],
                         capture_output=True, text=True, timeout=580, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and "REHEARSAL" in j["data"]
    assert j["config"]["spot_check"] == "ok" and j["config"]["exchange"] in ("all_to_all", "pipelined")
    fx = j["features_exchanges"]
    assert fx["features"]["spot_check"] == "ok" and fx["features_pipelined"]["spot_check"] == "ok"
    assert j["rows_scheme"]["spot_check"] == "ok" and j["rows_scheme"]["value"] > 0
    st = j["sharded_pre_training_step"]
    assert "error" not in st and st["rows"]["ms_per_step"] > 0 and st["features"]["ms_per_step"] > 0
    assert "note" not in j                                   # no phase was abandoned by the watchdog


# ----------------------------------------------------------------------------- drawn shapes of the feature-sharded aggregation
def _sweep_worker(rank, world, port, seeds, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import literalkg_amd as L
        from literalkg_amd import ops
        from literalkg_amd.sharding import FeatureShardedAggregation, ShardedAggregation, shard_bounds
        from literalkg_amd.synth import make_kg
        dev = torch.device("cuda", 0)
        n_done = 0
        for seed in seeds:
            rng = np.random.default_rng(seed)
            pick = lambda xs: xs[int(rng.integers(len(xs)))]
            n = int(pick([world, 7, 50, 3000, 30_000]))
            e = int(max(1, n * pick([0.3, 1, 5, 15])))
            dg = int(pick([3, 4, 8, 16, 32, 64]))           # columns per rank: the scalar path (3) and the 16-byte paths
            d = dg * world
            h, t, r = make_kg(n, min(e, n * n), pick(["zipf", "uniform"]), seed=seed)
            g = L.KGStructure.from_triples(n, h, t, r, device=dev)
            gen = torch.Generator(device=dev).manual_seed(seed)
            val = torch.rand(g.nnz, generator=gen, device=dev)
            x = torch.randn((n, d), generator=gen, device=dev)
            gx = torch.randn((n, d), generator=gen, device=dev)
            want_side = ops.spmm_raw(g.rowptr, g.col, val, x, n, long_rows=g.long_rows(False))
            want_grad = ops.spmm_raw(g.t_rowptr, g.t_col, ops.permute_values(val, g.t_perm), gx, n, long_rows=g.long_rows(True))
            kind = pick(["entries", "equal", "lopsided"])
            if kind == "entries":
                cuts = shard_bounds(g, world)
            elif kind == "equal":
                cuts = [min(n, -(-n // world) * i) for i in range(world + 1)]
            else:                                           # one rank owns (almost) everything, another nothing
                cuts = [0] + sorted(int(v) for v in rng.choice([0, n // 3, n], world - 1)) + [n]
            lo, hi = cuts[rank], cuts[rank + 1]
            what = (seed, world, n, g.nnz, dg, kind, cuts)
            fs = FeatureShardedAggregation(g, val, rank, world, d, cuts)
            cols = slice(rank * dg, (rank + 1) * dg)
            tol = dict(rtol=1e-4, atol=1e-4)
            panels = lambda tab: torch.stack([tab[lo:hi, i * dg:(i + 1) * dg] for i in range(world)]).contiguous()
            # the bench's forms: slab forward with the pipelined return exchange, row block -> slab backward in pieces
            side_slab, block = fs.forward_to_row_block(fs.column_slab(x), pieces=int(pick([1, 2, 4, 7])))
            torch.testing.assert_close(side_slab, want_side[:, cols], msg=lambda m_: f"forward slab {what}: {m_}", **tol)
            torch.testing.assert_close(block, panels(want_side), msg=lambda m_: f"forward row block {what}: {m_}", **tol)
            got = fs.backward_from_row_block(panels(gx), pieces=int(pick([1, 2, 3])))
            torch.testing.assert_close(got, want_grad[:, cols], msg=lambda m_: f"backward from row block {what}: {m_}", **tol)
            # the module's form: both exchanges folded into the SpMM, forward and transposed, with and without the self term
            for transposed, src, want in ((False, x, want_side), (True, gx, want_grad)):
                plus_self = bool(rng.random() < 0.5)
                slab_out, out = fs.exchange_aggregate(transposed, block_in=panels(src), plus_self=plus_self,
                                                      n_batches=pick([None, 1, 2, 5]), pieces=pick([None, 1, 3]))
                ref = want + src if plus_self else want
                torch.testing.assert_close(slab_out, ref[:, cols], msg=lambda m_: f"exchange_aggregate slab T={transposed} {what}: {m_}", **tol)
                torch.testing.assert_close(out, panels(ref), msg=lambda m_: f"exchange_aggregate block T={transposed} {what}: {m_}", **tol)
            # the north star's row-range form: own head rows forward, chunked all-reduce behind the transpose SpMM
            keep = (h >= lo) & (h < hi)
            mine = L.KGStructure.from_triples(n, h[keep], t[keep], r[keep], device=dev)
            rp = g.host("rowptr")
            val_mine = val[int(rp[lo]):int(rp[hi])].contiguous()
            assert mine.nnz == val_mine.numel(), what
            sh = ShardedAggregation(mine, val_mine, lo, hi, n_chunks=int(pick([1, 2, 3, 5])))
            if hi > lo:
                torch.testing.assert_close(sh.forward(x), want_side[lo:hi], msg=lambda m_: f"row-range forward {what}: {m_}", **tol)
            grad = sh.backward(gx[lo:hi].contiguous())         # (chunked all-reduce through transport.Transport)
            torch.testing.assert_close(grad, want_grad, msg=lambda m_: f"row-range backward {what}: {m_}", **tol)
            dist.barrier()
            n_done += 1
        assert n_done == len(seeds)
        q.put((rank, "ok"))
    except Exception as exc:   # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-3000:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_feature_sharded_aggregation_over_drawn_shapes(gpu_device, world):
    """The exchange forms bench.py times at N > 1 and the module uses (slab forward with the pipelined return exchange, the
    transpose from a row block in pieces, exchange_aggregate with its incoming sub-ranges and outgoing pieces) on drawn
    graphs, widths (scalar and 16-byte paths), row cuts (balanced by entries, equal, lopsided with an empty rank), batch and
    piece counts, and the row-range form with its chunked all-reduce -- 2, 3 and 4 ranks on the one GPU over gloo, every rank
    against the single-device product.
    LKG_FUZZ_FS_CASES cases per world size (default 30)."""
    import __graft_entry__ as ge
    ge.build()
    k = int(os.environ.get("LKG_FUZZ_FS_CASES", "30"))
    seeds = [51000 + 100 * world + i for i in range(k)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, seeds, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=840) for _ in procs]
    for p in procs:
        p.join(60)
    bad = [(r, msg) for r, msg in res if msg != "ok"]
    for r, msg in bad:
        print(f"---- rank {r}\n{msg}")
    assert not bad, f"{len(bad)} of {world} ranks failed (their tracebacks are in the captured output)"
