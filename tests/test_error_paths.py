"""Error paths of the multi-stream / multi-transfer passes (literalkg_amd/transport.py: InFlight).

The reference's only failure path is a clean exit (main_pretraining.py:112-116: NaN loss -> sys.exit); round 3's record
holds a run that ended in a GPU memory access fault instead (gpurun_out/r03_s1/n8_shape.log, root cause in DESIGN.md:
``lkg_permute_f32`` driven past the end of an index list).  Two guarantees are tested here:

  * CPU, gloo world 2: an exception raised between two launches of the pipelined exchange leaves the pass with every queued
    transfer waited for -- the SAME process group then still works (a collective right after the failed pass completes and
    is correct on both ranks);
  * GPU (child process, 2 ranks on the one GPU over gloo, the real HIP kernels, side streams in use): the same injection
    ends the child with a non-zero exit code, the injected error in its stderr and NO "Memory access fault" -- one run, no
    retry loop;
  * GPU: an index list that does not belong to its value array (round 3's defect, reproduced on purpose at a small size)
    yields NaNs, not a fault; a structure whose offsets or column ids point outside its operands is refused by
    ``ops.check_csr``."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Injected(RuntimeError):
    pass


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_sharding_gloo import cpu_permute, cpu_spmm
        from literalkg_amd import KGStructure
        from literalkg_amd.sharding import FeatureShardedAggregation, shard_bounds
        rng = np.random.default_rng(21)
        n, d, e = 600, 8, 5000
        h = (n * rng.random(e) ** 2.0).astype(np.int64)
        g = KGStructure.from_triples(n, h, rng.integers(0, n, e), rng.integers(0, 3, e))
        val = torch.from_numpy(rng.random(g.nnz).astype(np.float32))
        cuts = shard_bounds(g, world)
        calls = {"n": 0, "fail_at": None}

        def spmm(*a, **k):
            calls["n"] += 1
            if calls["fail_at"] is not None and calls["n"] == calls["fail_at"]:
                raise Injected("host-side error between two launches")
            return cpu_spmm(*a, **k)

        fs = FeatureShardedAggregation(g, val, rank, world, d, cuts, spmm=spmm, permute=cpu_permute)
        x = torch.from_numpy(np.random.default_rng(5).standard_normal((n, d)).astype(np.float32))
        lo, hi = cuts[rank], cuts[rank + 1]
        block = torch.stack([x[lo:hi, i * fs.dg:(i + 1) * fs.dg] for i in range(world)]).contiguous()
        want = fs.exchange_aggregate(False, block_in=block, plus_self=True)[1].clone()
        raised = []
        for fail_at in (2, 4):               # inside the part-wise passes / inside the owner-range rounds of the last part
            calls.update(n=0, fail_at=fail_at)
            try:
                fs.exchange_aggregate(False, block_in=block, plus_self=True)
                raised.append(False)
            except Injected:
                raised.append(True)
            # the group is still usable and in step on both ranks: a collective right behind the failed pass
            probe = torch.full((4,), float(rank + 1))
            dist.all_reduce(probe)
            raised.append(bool((probe == sum(range(1, world + 1))).all()))
        calls.update(n=0, fail_at=None)
        again = fs.exchange_aggregate(False, block_in=block, plus_self=True)[1]
        q.put((rank, raised, bool(torch.equal(again, want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_exception_inside_the_pipelined_exchange_leaves_the_group_usable():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(r, [True, True, True, True], True) for r in range(world)], res


def test_inflight_waits_every_transfer_on_an_exception():
    from literalkg_amd.transport import InFlight, Pending
    waited = []

    class Work:
        def __init__(self, k):
            self.k = k

        def wait(self, *a):
            waited.append(self.k)

    with pytest.raises(Injected):
        with InFlight(None) as fl:
            fl.add(Pending([Work(0), Work(1)], None, lambda: waited.append("landed")))
            fl.add(Pending([Work(2)]))
            raise Injected("x")
    assert waited == [0, 1, "landed", 2]


CHILD = textwrap.dedent('''
    import os, sys, socket
    sys.path.insert(0, %(root)r)
    import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp

    class Injected(RuntimeError):
        pass

    def worker(rank, world, port):
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            import literalkg_amd as L
            from literalkg_amd import ops
            from literalkg_amd.sharding import FeatureShardedAggregation, shard_bounds
            from literalkg_amd.synth import make_kg
            dev = torch.device("cuda", 0)
            n, e, d = 400_000, 6_000_000, 128                 # launches long enough to be in flight when the error comes
            h, t, r = make_kg(n, e, seed=3)
            g = L.KGStructure.from_triples(n, h, t, r, device=dev)
            val = torch.rand(g.nnz, device=dev)
            cuts = shard_bounds(g, world)
            calls = {"n": 0}

            def spmm(*a, **k):
                calls["n"] += 1
                if calls["n"] == %(fail_at)d:
                    raise Injected("host-side error between two launches")
                return ops.spmm_raw(*a, **k)

            fs = FeatureShardedAggregation(g, val, rank, world, d, cuts, spmm=spmm, permute=ops.permute_values)
            x = torch.randn((n, d), device=dev)
            lo, hi = cuts[rank], cuts[rank + 1]
            block = torch.stack([x[lo:hi, i * fs.dg:(i + 1) * fs.dg] for i in range(world)]).contiguous()
            fs.exchange_aggregate(False, block_in=block, plus_self=True)     # raises Injected mid-pass
        finally:
            dist.destroy_process_group()

    if __name__ == "__main__":
        import __graft_entry__ as ge
        ge.build()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
        codes = [p.exitcode for p in procs]
        print("rank exit codes", codes, file=sys.stderr)
        sys.exit(0 if all(c == 0 for c in codes) else 3)
''')


@pytest.mark.gpu
@pytest.mark.timeout(420)
def test_host_error_between_launches_ends_the_process_cleanly(gpu_device, tmp_path):
    """ONE run of the pipelined pass in a child process with an error injected between two of its launches (the 6th SpMM
    call: inside the owner-range rounds on the side streams, transfers queued): non-zero exit, the injected error reported,
    no device fault."""
    script = tmp_path / "child.py"
    script.write_text(CHILD % {"root": ROOT, "fail_at": 6})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    err = res.stderr
    assert res.returncode != 0, err[-2000:]
    assert "Injected" in err, err[-2000:]
    assert "Memory access fault" not in err and "core dump" not in err.lower(), err[-3000:]


@pytest.mark.gpu
def test_foreign_index_list_gives_nans_not_a_fault(gpu_device):
    """round 3's defect on purpose, small: an index list longer than / foreign to its value array"""
    from literalkg_amd import _native as N, ops
    val = torch.rand(1000, device=gpu_device)
    perm = torch.randint(0, 1000, (4000,), device=gpu_device, dtype=torch.int32)
    perm[::7] = 2_000_000_000            # far outside the value array
    perm[3] = -5
    out = ops.permute_values(val, perm)
    torch.cuda.synchronize()
    bad = (perm < 0) | (perm >= 1000)
    assert bool(torch.isnan(out[bad]).all()) and bool(torch.equal(out[~bad], val[perm[~bad].long()]))


@pytest.mark.gpu
def test_check_csr_refuses_structures_that_point_outside_their_operands(gpu_device):
    import literalkg_amd as L
    from literalkg_amd import _native as N, ops
    from literalkg_amd.synth import make_kg
    n = 5000
    h, t, r = make_kg(n, 40_000, seed=5)
    g = L.KGStructure.from_triples(n, h, t, r, device=gpu_device)
    ops.check_csr(g.rowptr, g.col, n, n)                       # the library's own structures pass
    ops.check_csr(g.t_rowptr, g.t_col, n, n)
    ops.check_csr(g.rowptr[100:1101].contiguous(), g.col, 1000, n)
    with pytest.raises(N.LkgError):                            # a source table with fewer rows than the column ids reach
        ops.check_csr(g.rowptr, g.col, n, n - 1000)
    with pytest.raises(N.LkgError):                            # offsets beyond the entry arrays (a part's offsets, the whole's length)
        ops.check_csr(g.rowptr, g.col[: g.nnz // 2].contiguous(), n, n)
    rp = g.rowptr.clone()
    rp[10] = rp[11] + 5                                        # a descending pair of offsets
    with pytest.raises(N.LkgError):
        ops.check_csr(rp, g.col, n, n)
    with pytest.raises(ValueError):                            # and the host-side extent check of the launch itself
        ops.spmm_raw(g.rowptr, g.col, torch.rand(g.nnz - 1, device=gpu_device), torch.rand((n, 8), device=gpu_device), n)
