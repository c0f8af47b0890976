"""CPU: the drop-in nn.Module surface (ctor signature, state_dict keys/shapes, sparse A_in parameter,
mode dispatch) and the no-CPU-fallback guarantee."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, golden_cfg, golden_params, load_golden


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge
    ge.build()
    import literalkg_amd
    return literalkg_amd


def test_state_dict_manifest_matches_reference(L):
    manifest = json.load(open(os.path.join(GOLDEN, "statedict_manifest.json")))
    assert len(manifest) >= 13
    for name, want in manifest.items():
        gd = load_golden("encoder_" + name)
        m = L.LiteralKG(golden_cfg(gd), int(gd["n"]), int(gd["n_rel"]))
        got = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert got == want, name


def test_load_reference_weights_and_roundtrip(L):
    gd = load_golden("encoder_gcn_l2_scale")
    n = int(gd["n"])
    a = torch.sparse_coo_tensor(torch.from_numpy(gd["a_indices"]), torch.from_numpy(gd["a_values"]), (n, n)).coalesce()
    m = L.LiteralKG(golden_cfg(gd), n, int(gd["n_rel"]), a)
    res = m.load_state_dict(golden_params(gd), strict=False)
    assert res.missing_keys == ["A_in"] and not res.unexpected_keys
    assert m.A_in.is_sparse and not m.A_in.requires_grad and m.A_in._nnz() == gd["a_values"].size
    assert all(p.requires_grad for k, p in m.named_parameters() if k != "A_in")
    m2 = L.LiteralKG(golden_cfg(gd), n, int(gd["n_rel"]))
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m2.A_in.data.coalesce().indices(), a.indices())
    assert torch.equal(m2.entity_embed.weight, m.entity_embed.weight)
    att = m._attention()
    assert att.graph.nnz == a._nnz() and att.val.data_ptr() == m.A_in.data._values().data_ptr()
    assert "LiteralKG" in repr(m)


def test_transe_variant_has_no_projection(L):
    gd = load_golden("transe_gcn_l1")
    m = L.LiteralKG(golden_cfg(gd), int(gd["n"]), int(gd["n_rel"]), scoring="transe")
    assert "gat_trans_M" not in m.state_dict()


def test_unknown_aggregator_raises(L):
    gd = load_golden("encoder_gcn_l1")
    cfg = golden_cfg(gd)
    cfg.aggregation_type = "nope"
    with pytest.raises(NotImplementedError):
        L.LiteralKG(cfg, 10, 2)


def test_no_cpu_fallback(L):
    gd = load_golden("encoder_gcn_l1")
    n = int(gd["n"])
    a = torch.sparse_coo_tensor(torch.from_numpy(gd["a_indices"]), torch.from_numpy(gd["a_values"]), (n, n)).coalesce()
    m = L.LiteralKG(golden_cfg(gd), n, int(gd["n_rel"]), a)
    ids = torch.zeros(3, dtype=torch.long)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(ids, ids, ids, ids, device=torch.device("cpu"), mode="pre_training")
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.from_numpy(gd["h"]), torch.from_numpy(gd["t"]), torch.from_numpy(gd["r"]), [0, 1, 2, 3],
          device=torch.device("cpu"), mode="update_att")
    assert m(ids, device=torch.device("cpu"), mode="bogus") is None


def test_product_code_never_imports_the_oracle():
    root = os.path.dirname(GOLDEN.rstrip("/"))
    pkg = os.path.join(os.path.dirname(root), "literalkg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "/root/reference" not in text, f
