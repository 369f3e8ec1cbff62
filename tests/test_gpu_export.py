"""f2 interoperability on real factors: layers factorised by the HIP path are exported with `save_asvd_repo` and the repo is checked
against what the REFERENCE's remote-code model classes create for the same truncation_ranks (tests/golden/hf_export_ref.json)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fam", ["llama", "opt"])
def test_hip_factors_export_in_the_reference_layout(gpu, fam, tmp_path, golden):
    from asvd4llm_amd.export import load_asvd_repo, save_asvd_repo
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    from tests.test_export_format import _build_from_fixture, check_repo_against_reference_fixture
    rec = golden.json("hf_export_ref.json")["families"][fam]
    model = _build_from_fixture(rec, fam).to(gpu)
    ids = torch.randint(0, model.config.vocab_size, (1, 12), generator=torch.Generator().manual_seed(1)).to(gpu)
    g = torch.Generator().manual_seed(0)
    for full, r in rec["truncation_ranks"].items():
        parent_name, _, child = full.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        lin = getattr(parent, child)
        assert isinstance(lin, nn.Linear)
        lin.scaling_diag_matrix = (torch.rand(lin.in_features, generator=g) + 0.1).to(gpu)
        ratio = r * (lin.in_features + lin.out_features) / (lin.in_features * lin.out_features)
        new = SVDLinear.from_linear(lin, ratio, act_aware=True, alpha=0.5)   # the hand-written kernels
        assert isinstance(new, SVDLinear) and new.truncation_rank == r
        setattr(parent, child, new)
    with torch.no_grad():
        want = model(input_ids=ids)[0].float().cpu()
    path = str(tmp_path / "repo")
    assert save_asvd_repo(model.cpu(), path) == rec["truncation_ranks"]
    check_repo_against_reference_fixture(path, rec)
    m = load_asvd_repo(path, dtype=torch.float32)
    with torch.no_grad():
        got = m(input_ids=ids.cpu())[0]
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5)
