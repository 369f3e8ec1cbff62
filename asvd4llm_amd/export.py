"""On-disk format of a compressed model — interoperable with the reference's exported repos.

The reference's exporter (huggingface_repos/build_asvd_repo.py:58-92) writes an ordinary `save_pretrained` checkpoint whose
factorised layers appear under the state-dict keys `<layer>.ALinear.weight` / `.ALinear.bias` / `<layer>.BLinear.weight`,
plus `config.json` extended with
    "truncation_ranks": {<layer full name>: rank},  "auto_map": {...},  "architectures": ["ASVD<Family>ForCausalLM"]
and a pair of remote-code files whose model class rebuilds the two-Linear modules from `truncation_ranks` before the weights
are loaded (huggingface_repos/modeling_asvd_llama.py:5-41).  `save_asvd_repo` writes exactly that layout; the loader files
it emits are generated from the templates below (subclassing the stock HF config/model classes), so the directory loads
with `AutoModelForCausalLM.from_pretrained(path, trust_remote_code=True)` as the published `hahnyuan/*-asvd*` repos do.
`load_asvd_repo` is a family-agnostic loader that needs no remote code; it also reads repos exported by the reference.
"""
import json
import os

import torch
import torch.nn as nn

from .modules.svd_linear import SVDLinear

_FAMILIES = {
    # family tag -> (HF config class, HF model class, file stem)
    "llama": ("LlamaConfig", "LlamaForCausalLM", "llama"),
    "opt": ("OPTConfig", "OPTForCausalLM", "opt"),
}

_CONFIG_TEMPLATE = '''"""Config of an ASVD-compressed {cls_model}: the stock config plus `truncation_ranks` ({{layer name: rank}})."""
from transformers import {cls_config}


class ASVD{Fam}Config({cls_config}):
    model_type = "{model_type}"

    def __init__(self, truncation_ranks=None, **kwargs):
        super().__init__(**kwargs)
        self.truncation_ranks = truncation_ranks if truncation_ranks is not None else {{}}
'''

_MODELING_TEMPLATE = '''"""ASVD-compressed {cls_model}: every layer named in config.truncation_ranks is a pair BLinear (in -> r, no bias),
ALinear (r -> out, original bias) — the module/key names of ASVD4LLM's SVDLinear."""
import torch.nn as nn
from transformers import {cls_model}

from .configuration_asvd_{stem} import ASVD{Fam}Config


class ASVDLinear(nn.Module):
    def __init__(self, in_features, out_features, rank, bias=True):
        super().__init__()
        self.BLinear = nn.Linear(in_features, rank, bias=False)
        self.ALinear = nn.Linear(rank, out_features, bias=bias)
        self.truncation_rank = rank

    def forward(self, x):
        return self.ALinear(self.BLinear(x))


class ASVD{Fam}ForCausalLM({cls_model}):
    config_class = ASVD{Fam}Config

    def __init__(self, config):
        super().__init__(config)
        self.truncation_ranks = dict(config.truncation_ranks or {{}})
        for full_name, rank in self.truncation_ranks.items():
            parent_name, _, child = full_name.rpartition(".")
            parent = self.get_submodule(parent_name) if parent_name else self
            lin = getattr(parent, child)
            setattr(parent, child, ASVDLinear(lin.in_features, lin.out_features, int(rank), bias=lin.bias is not None))
'''


def _family_of(model_or_config):
    cfg = getattr(model_or_config, "config", model_or_config)
    mt = getattr(cfg, "model_type", "")
    for fam in _FAMILIES:
        if fam in mt:
            return fam
    raise ValueError(f"no ASVD repo template for model_type={mt!r} (supported: {sorted(_FAMILIES)})")


def truncation_ranks_of(model):
    """{full layer name: rank} of every SVDLinear, in named_modules order (build_asvd_repo.py:66-69)."""
    return {name: int(m.truncation_rank) for name, m in model.named_modules() if isinstance(m, SVDLinear)}


def replace_with_factor_pairs(model, truncation_ranks, dtype=None):
    """Swap each named nn.Linear for an (uninitialised) SVDLinear-shaped module so a checkpoint with ALinear/BLinear keys fits."""
    for full_name, rank in truncation_ranks.items():
        parent_name, _, child = full_name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        lin = getattr(parent, child)
        if not isinstance(lin, nn.Linear):
            raise ValueError(f"{full_name} is not an nn.Linear in the base architecture")
        dt = dtype or lin.weight.dtype
        A = torch.empty(lin.out_features, int(rank), dtype=dt, device=lin.weight.device)
        B = torch.empty(int(rank), lin.in_features, dtype=dt, device=lin.weight.device)
        bias = torch.empty(lin.out_features, dtype=dt, device=lin.weight.device) if lin.bias is not None else None
        setattr(parent, child, SVDLinear._from_factors(A, B, bias, int(rank)))
    return model


def save_asvd_repo(model, save_path, tokenizer=None, family=None):
    """Write `save_path/` in the reference's exported-repo layout.  Returns the truncation_ranks dict."""
    fam = family or _family_of(model)
    cls_config, cls_model, stem = _FAMILIES[fam]
    Fam = {"llama": "Llama", "opt": "OPT"}[fam]
    os.makedirs(save_path, exist_ok=True)
    if tokenizer is not None:
        tokenizer.save_pretrained(save_path)
    model.save_pretrained(save_path)
    config = model.config.to_dict()
    ranks = truncation_ranks_of(model)
    config["truncation_ranks"] = ranks
    config["auto_map"] = {
        "AutoConfig": f"configuration_asvd_{stem}.ASVD{Fam}Config",
        "AutoModelForCausalLM": f"modeling_asvd_{stem}.ASVD{Fam}ForCausalLM",
    }
    config["architectures"] = [f"ASVD{Fam}ForCausalLM"]
    fmt = dict(cls_config=cls_config, cls_model=cls_model, stem=stem, Fam=Fam, model_type=config.get("model_type", fam))
    with open(os.path.join(save_path, f"configuration_asvd_{stem}.py"), "w") as f:
        f.write(_CONFIG_TEMPLATE.format(**fmt))
    with open(os.path.join(save_path, f"modeling_asvd_{stem}.py"), "w") as f:
        f.write(_MODELING_TEMPLATE.format(**fmt))
    with open(os.path.join(save_path, "config.json"), "w") as f:
        json.dump(config, f, indent=2)
    return ranks


def _read_state_dict(path):
    st = {}
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if files:
        from safetensors.torch import load_file
        for f in files:
            st.update(load_file(os.path.join(path, f)))
        return st
    files = sorted(f for f in os.listdir(path) if f.endswith(".bin") and f.startswith("pytorch_model"))
    for f in files:
        st.update(torch.load(os.path.join(path, f), map_location="cpu"))
    if not st:
        raise FileNotFoundError(f"no weights (*.safetensors / pytorch_model*.bin) under {path}")
    return st


def load_asvd_repo(path, dtype=torch.float16, device="cpu"):
    """Load an exported ASVD repo (ours or the reference's) without remote code: stock architecture from config.json,
    factor pairs from `truncation_ranks`, then the checkpoint (strict, except tied / re-creatable buffers)."""
    from transformers import AutoConfig, AutoModelForCausalLM
    with open(os.path.join(path, "config.json")) as f:
        raw = json.load(f)
    ranks = raw.pop("truncation_ranks", {}) or {}
    raw.pop("auto_map", None)
    fam = next((k for k in _FAMILIES if k in raw.get("model_type", "")), None)
    if fam is None:
        raise ValueError(f"unsupported model_type {raw.get('model_type')!r}")
    raw["architectures"] = [_FAMILIES[fam][1]]
    cfg = AutoConfig.for_model(**raw)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        model = AutoModelForCausalLM.from_config(cfg)
    finally:
        torch.set_default_dtype(prev)
    replace_with_factor_pairs(model, ranks, dtype=dtype)
    state = _read_state_dict(path)
    missing, unexpected = model.load_state_dict(state, strict=False)
    tied_ok = {"lm_head.weight"} if getattr(cfg, "tie_word_embeddings", False) else set()
    bad_missing = [k for k in missing if k not in tied_ok and not k.endswith("rotary_emb.inv_freq")]
    if bad_missing or unexpected:
        raise RuntimeError(f"checkpoint does not match truncation_ranks: missing={bad_missing[:5]} unexpected={list(unexpected)[:5]}")
    if tied_ok and "lm_head.weight" in missing:
        model.tie_weights()
    model.truncation_ranks = ranks
    return model.to(device).eval()
