"""BASELINE configs[2] as a whole: every Linear of a Llama-2-7B-shaped model (225: 128 x 4096^2, 64 x 11008x4096, 32 x 4096x11008, the
32000x4096 lm_head) decomposed through the product path (prefactorize -> from_linear, ratio 0.9, alpha 0.5) on one GPU — the leg bench.py
prints as "full_model", the stage the reference times as `decompose time` (binary_search.py:111-131) — with one layer of every distinct
shape checked against the CPU oracle (torch.linalg.svd on the same scaled weights): sigma <= 1e-4 relative over the retained rank,
reconstruction <= 1e-3 |W|_F."""
import os
import sys

import pytest
import torch

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_llama2_7b_shaped_model_decomposes_with_parity(gpu):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["ASVD_STRICT"] = "1"
    samples = []
    rec = bench.sharded_model_leg("llama-2-7b", 0, 1, gpu, samples=samples)
    assert "error" not in rec, rec
    assert rec["linears"] == 225 and rec["layers_per_rank"] == [225]
    assert rec["plan_identical_on_all_ranks"] and 0.85 < rec["plan_param_ratio"] <= 0.9001
    assert rec["sweeps_min_max_rank0"][1] <= 12
    assert 0 < rec["decompose_s"] < 30.0                      # 4.6-4.8 s measured (round 4-5); a fallback to anything slower is a defect
    assert rec["achieved_TFLOPs_whole_job"] > 20.0
    assert sorted(tuple(s["shape"]) for s in samples) == [(4096, 4096), (4096, 11008), (11008, 4096), (32000, 4096)]
    par = bench.full_model_parity(samples, "llama-2-7b", gpu, min(16, os.cpu_count() or 1), rec["decompose_s"])
    assert par["parity_ok"], par["parity_per_shape"]
    for p in par["parity_per_shape"]:
        assert p["sigma_rel_err_top_r"] <= 1e-4 and p["recon_fro_err_vs_oracle"] <= 1e-3 and p["recon_fro_err_scaled_norm"] <= 1e-3, p
    assert sum(p["count_in_model"] for p in par["parity_per_shape"]) == 225
    assert par["speedup_vs_cpu_reference"] > 20.0             # north_star asks for >= 20x on 8 GPUs; one GPU delivers it alone


@pytest.mark.timeout(2400)
def test_llama2_13b_shaped_model_ratio_095_decomposes_with_parity(gpu):
    """BASELINE configs[4]'s model as a whole on one GPU: the 281 Linears of a Llama-2-13B-shaped model (160 x 5120^2, 80 x 13824x5120,
    40 x 5120x13824, the 32000x5120 lm_head — 80 super-panels: the grouped schedule) at --param_ratio_target 0.95, one layer of every distinct
    shape against the CPU oracle.  (The sensitivity sweep of configs[4] is covered on the tiny LM — tests/test_gpu_pipeline.py — and as a
    builder-run record, profiles/r*_e2e_llama2_13b_ratio095_*.json: 13B forwards take minutes per layer.)"""
    sys.path.insert(0, ROOT)
    import bench
    os.environ["ASVD_STRICT"] = "1"
    samples = []
    rec = bench.sharded_model_leg("llama-2-13b", 0, 1, gpu, ratio=0.95, samples=samples)
    assert "error" not in rec, rec
    assert rec["linears"] == 281 and rec["layers_per_rank"] == [281]
    assert rec["plan_identical_on_all_ranks"] and 0.9 < rec["plan_param_ratio"] <= 0.951   # the search stops at the first cut whose ratio exceeds the target (binary_search.py:29-110)
    assert rec["sweeps_min_max_rank0"][1] <= 12
    assert 0 < rec["decompose_s"] < 60.0                      # 10.8-11.6 s measured (round 5)
    assert sorted(tuple(s["shape"]) for s in samples) == [(5120, 5120), (5120, 13824), (13824, 5120), (32000, 5120)]
    for smp in samples:   # the rank arithmetic at ratio 0.95 (svd_linear.py:39-44)
        o, i = smp["shape"]
        assert smp["rank"] == int(o * i * 0.95) // (o + i)
    par = bench.full_model_parity(samples, "llama-2-13b", gpu, min(16, os.cpu_count() or 1), rec["decompose_s"])
    assert par["parity_ok"], par["parity_per_shape"]
    for p in par["parity_per_shape"]:
        assert p["sigma_rel_err_top_r"] <= 1e-4 and p["recon_fro_err_vs_oracle"] <= 1e-3 and p["recon_fro_err_scaled_norm"] <= 1e-3, p
    assert sum(p["count_in_model"] for p in par["parity_per_shape"]) == 281
