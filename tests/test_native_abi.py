"""CPU: the C-ABI library loads, exports every symbol include/literalkg_hip.h declares, and its HOST
entry points (KG structure build) are bit-exact against a numpy restatement.  No device calls."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as ge
    ge.build()
    from literalkg_amd import _native
    return _native


def header_symbols():
    text = open(os.path.join(ROOT, "include", "literalkg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lkg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(native):
    lib = native.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/literalkg_hip.h but not exported"
    assert sorted(native.PROTOTYPES) == names        # the ctypes table mirrors the header exactly
    assert lib.lkg_version() >= 100


def test_error_convention(native):
    h = np.array([0, 5], np.int64)
    t = np.array([1, 1], np.int64)
    out = [np.zeros(8, np.int32) for _ in range(4)] + [np.zeros(8, np.int64), np.zeros(1, np.int64)]
    with pytest.raises(native.LkgError, match="outside"):
        native.call("lkg_csr_build", 3, 2, native.ptr(h), native.ptr(t), None, *[native.ptr(o) for o in out])
    with pytest.raises(native.LkgError, match="null"):
        native.call("lkg_csr_build", 3, 2, None, native.ptr(t), None, *[native.ptr(o) for o in out])
    with pytest.raises(native.LkgError, match="d must be positive"):
        native.call("lkg_spmm_csr_f32", 4, 0, None, None, None, None, 0, None, 0, None, 0, None, 0, 0, None)


def numpy_csr(n, h, t, r):
    """Restatement of coalesce()'s index work (model.py:468-470): sort by (h,t), merge equal pairs."""
    order = np.lexsort((np.arange(len(h)), t, h))
    hs, ts = h[order], t[order]
    new = np.r_[True, (hs[1:] != hs[:-1]) | (ts[1:] != ts[:-1])] if len(h) else np.zeros(0, bool)
    eptr = np.r_[np.flatnonzero(new), len(h)]
    col = ts[new]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, hs[new] + 1, 1)
    return np.cumsum(rowptr), col, eptr, r[order], order


@pytest.mark.parametrize("seed,n,e", [(0, 50, 400), (1, 1000, 20000), (2, 7, 0), (3, 300, 3000)])
def test_csr_build_bit_exact(native, seed, n, e):
    from literalkg_amd import KGStructure
    rng = np.random.default_rng(seed)
    h = (n * rng.random(e) ** 2).astype(np.int64)
    t = rng.integers(0, max(n // 3, 1), e)                  # many duplicate (h,t) pairs
    r = rng.integers(0, 4, e)
    g = KGStructure.from_triples(n, h, t, r)
    rowptr, col, eptr, rel, order = numpy_csr(n, h, t, r)
    assert g.nnz == len(col) and g.n_raw == e
    assert np.array_equal(g.host("rowptr"), rowptr)
    assert np.array_equal(g.host("col"), col)
    assert np.array_equal(g.host("rel"), rel)
    assert np.array_equal(g.order, order)
    if g.nnz != e:
        assert np.array_equal(g.host("eptr"), eptr)
    else:
        assert g.host("eptr") is None
    # transpose: same entries, sorted by (t, h); t_perm points back at the CSR entry
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    tp = g.host("t_perm")
    assert np.array_equal(np.sort(tp), np.arange(g.nnz))
    assert np.array_equal(g.host("t_col"), rows[tp])
    tr = np.repeat(np.arange(n), np.diff(g.host("t_rowptr")))
    assert np.array_equal(tr, col[tp])
    assert np.all(np.diff(tr.astype(np.int64) * n + rows[tp]) > 0)
    assert g.coo_indices().dtype == torch.int64
    assert np.array_equal(g.coo_indices().numpy(), np.stack([rows, col]))


@pytest.mark.parametrize("name", ["attention_toy5", "attention_rand300", "attention_testslice"])
def test_structure_equals_reference_coalesced_indices(native, name):
    from literalkg_amd import KGStructure
    gd = load_golden(name)
    g = KGStructure.from_triples(int(gd["n"]), gd["h"], gd["t"], gd["r"])
    assert np.array_equal(g.coo_indices().numpy(), gd["a_indices"])       # bit-exact vs torch coalesce
    assert g.has_dups == (g.nnz < len(gd["h"]))


def test_row_partition_balances_entries(native):
    from literalkg_amd import KGStructure
    rng = np.random.default_rng(4)
    n, e = 5000, 60000
    h = (n * rng.random(e) ** 3).astype(np.int64)
    g = KGStructure.from_triples(n, h, rng.integers(0, n, e), None)
    rp = g.host("rowptr")
    for parts in (1, 2, 3, 8):
        cuts = g.row_cuts(parts)
        assert cuts[0] == 0 and cuts[-1] == n and np.all(np.diff(cuts) >= 0)
        loads = np.diff(rp[cuts])
        assert loads.sum() == g.nnz
        assert loads.max() - g.nnz / parts <= np.diff(rp).max()       # within one row of perfect balance


def test_from_coo_keeps_value_order(native):
    from literalkg_amd import KGStructure
    gd = load_golden("encoder_gcn_l1")
    n = int(gd["n"])
    a = torch.sparse_coo_tensor(torch.from_numpy(gd["a_indices"]), torch.from_numpy(gd["a_values"]), (n, n)).coalesce()
    g = KGStructure.from_coo(a)
    assert np.array_equal(g.coo_indices().numpy(), gd["a_indices"]) and not g.has_dups


# ----------------------------------------------------------------------------- f3 ingestion
def test_triples_file_roundtrip_and_dedup(native, tmp_path):
    from literalkg_amd import io
    gd = load_golden("attention_testslice")           # a slice of the reference's shipped KG file (data)
    h, t, r = gd["h"], gd["t"], gd["r"]
    rows = np.stack([h, r, t], 1)
    rows = np.concatenate([rows, rows[5:25], rows[:3]])            # inject duplicate rows
    path = tmp_path / "pre_training_train.txt"
    with open(path, "w") as f:
        for i, (a, b, c) in enumerate(rows):
            f.write(f"{a} {b} {c}" + ("\r\n" if i % 7 == 0 else "\n"))
        f.write("\n")                                               # trailing blank line
    h2, r2, t2 = io.load_triples(str(path), drop_duplicates=False)
    assert np.array_equal(np.stack([h2, r2, t2], 1), rows)
    h3, r3, t3 = io.load_triples(str(path))
    import pandas as pd
    want = pd.DataFrame(rows, columns=list("hrt")).drop_duplicates()  # the loader's call (dataloader.py:189)
    assert np.array_equal(np.stack([h3, r3, t3], 1), want.to_numpy())
    bad = tmp_path / "bad.txt"
    bad.write_text("1 2 3\n4 x 6\n")
    with pytest.raises(native.LkgError, match="malformed"):
        io.load_triples(str(bad))
    with pytest.raises(native.LkgError, match="cannot open"):
        io.load_triples(str(tmp_path / "missing.txt"))


@pytest.mark.parametrize("kind", ["random-walk", "symmetric"])
def test_initial_a_in_matches_loader_restatement(native, kind):
    from literalkg_amd import io
    from oracle import literalkg_oracle as O
    for name in ("encoder_gcn_l1", "attention_testslice"):
        gd = load_golden(name)
        n = int(gd["n"])
        h, t, r = gd["h"], gd["t"], gd["r"]
        a = io.initial_a_in(n, h, t, r, kind)
        want = O.laplacian_a_in(n, torch.from_numpy(h), torch.from_numpy(t), torch.from_numpy(r), kind)
        assert a.is_coalesced() and torch.equal(a.indices(), want.indices())
        torch.testing.assert_close(a.values(), want.values(), rtol=1e-6, atol=1e-7)
    gd = load_golden("encoder_gcn_l1")              # and against the scipy-built A_in stored in the fixture
    a = io.initial_a_in(int(gd["n"]), gd["h"], gd["t"], gd["r"], "random-walk")
    assert np.array_equal(a.indices().numpy(), gd["a_indices"])
    np.testing.assert_allclose(a.values().numpy(), gd["a_values"], rtol=1e-6)
    with pytest.raises(NotImplementedError):
        io.initial_a_in(5, np.array([0]), np.array([1]), np.array([0]), "other")


REF_KG = "/root/reference/data/Test/pre_training_train.txt"


@pytest.mark.skipif(not os.path.exists(REF_KG), reason="reference mount absent (GPU box): build-container check only")
def test_triples_file_parsed_and_deduplicated_by_many_threads(native, tmp_path):
    """Files beyond a few MB are parsed over line-aligned byte ranges by several threads and de-duplicated bucket by bucket
    (lkg_triples_read / lkg_triples_dedup): same answer as pandas' drop_duplicates(keep='first') in file order
    (dataloader.py:186-190), blank lines and a CRLF ending included; a malformed line anywhere is still an error."""
    import pandas as pd
    from literalkg_amd import io
    rng = np.random.default_rng(1)
    n = 800_000
    h, r, t = rng.integers(0, 40_000, n), rng.integers(0, 8, n), rng.integers(0, 40_000, n)
    dup = rng.integers(0, n, 60_000)
    h[dup], r[dup], t[dup] = h[(dup * 7) % n], r[(dup * 7) % n], t[(dup * 7) % n]
    df = pd.DataFrame({"h": h, "r": r, "t": t})
    path = tmp_path / "kg_final.txt"
    df.to_csv(path, sep=" ", header=False, index=False)
    with open(path, "a") as f:
        f.write("\n\n7 1 9\r\n")
    assert os.path.getsize(path) > 8 << 20                    # (several parser threads)
    hh, rr, tt = io.load_triples(str(path))
    want = pd.concat([df, pd.DataFrame({"h": [7], "r": [1], "t": [9]})], ignore_index=True).drop_duplicates(keep="first")
    assert len(want) < n and len(hh) == len(want)
    assert np.array_equal(hh, want["h"].to_numpy()) and np.array_equal(rr, want["r"].to_numpy())
    assert np.array_equal(tt, want["t"].to_numpy())
    h2, r2, t2 = io.load_triples(str(path), drop_duplicates=False)
    assert len(h2) == n + 1 and np.array_equal(h2[:n], h) and np.array_equal(t2[:n], t) and (h2[-1], r2[-1], t2[-1]) == (7, 1, 9)
    with open(path, "r+") as f:                               # damage a line in the middle of the file
        f.seek(os.path.getsize(path) // 2)
        f.readline()
        f.write("x")
    with pytest.raises(native.LkgError, match="malformed"):
        io.load_triples(str(path))


def test_ingest_the_reference_kg_file(native):
    """The reference's only complete KG file: parse, drop duplicates and build A_in like its DataLoader
    (dataloader.py:186-190, 449-495), against pandas / the oracle's restatement."""
    import pandas as pd
    from literalkg_amd import io
    from oracle import literalkg_oracle as O
    h, r, t = io.load_triples(REF_KG)
    want = pd.read_csv(REF_KG, sep=" ", names=["h", "r", "t"]).drop_duplicates()
    assert np.array_equal(np.stack([h, r, t], 1), want.to_numpy())
    assert len(h) == 217463 and len(set(r.tolist())) == 15
    n = int(max(h.max(), t.max())) + 1
    a = io.initial_a_in(n, h, t, r)
    want_a = O.laplacian_a_in(n, torch.from_numpy(h), torch.from_numpy(t), torch.from_numpy(r))
    assert torch.equal(a.indices(), want_a.indices())
    torch.testing.assert_close(a.values(), want_a.values(), rtol=1e-6, atol=1e-7)
    rs = torch.zeros(n).index_add_(0, a.indices()[0], a.values())
    assert abs(float(rs[0]) - 9.0) < 1e-5                            # row 0 sums to 9
    assert a._nnz() == len(h) - 33                                   # its 33 duplicate (h,t) pairs are merged


@pytest.mark.parametrize("kind", ["random-walk", "symmetric"])
def test_initial_a_in_matches_the_reference_loader(native, kind):
    """Fixture produced by the reference's own create_adjacency_dict / create_laplacian_dict
    (dataloader.py:449-495, oracle/gen_golden.py --only-laplacian)."""
    from literalkg_amd import io
    from oracle import literalkg_oracle as O
    gd = load_golden("laplacian_rand260")
    n = int(gd["n"])
    key = kind.replace("-", "_")
    a = io.initial_a_in(n, gd["h"], gd["t"], gd["r"], kind)
    assert np.array_equal(a.indices().numpy(), gd[key + "_indices"])
    np.testing.assert_allclose(a.values().numpy(), gd[key + "_values"], rtol=1e-6, atol=1e-7)
    want = O.laplacian_a_in(n, *(torch.from_numpy(gd[k]) for k in "htr"), kind)     # pins the oracle's restatement too
    assert np.array_equal(want.indices().numpy(), gd[key + "_indices"])
    np.testing.assert_allclose(want.values().numpy(), gd[key + "_values"], rtol=1e-6, atol=1e-7)
