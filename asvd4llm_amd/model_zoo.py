"""Shape-faithful, randomly initialised stand-ins for the models BASELINE.json names (no checkpoints / network in this image).
Only shapes matter for the compression path; weights are N(0, 0.02^2) as HF initialises them."""
import torch

NAMED = {
    # name -> (family, kwargs)
    "opt-125m": ("opt", dict(hidden_size=768, ffn_dim=3072, num_hidden_layers=12, num_attention_heads=12, vocab_size=50272,
                             word_embed_proj_dim=768, max_position_embeddings=2048)),
    "llama-2-7b": ("llama", dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                                 num_key_value_heads=32, vocab_size=32000, max_position_embeddings=4096)),
    "llama-2-13b": ("llama", dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                                  num_key_value_heads=40, vocab_size=32000, max_position_embeddings=4096)),
    "tiny-llama": ("llama", dict(hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=4,
                                 num_key_value_heads=4, vocab_size=512, max_position_embeddings=2048)),
    "tiny-opt": ("opt", dict(hidden_size=128, ffn_dim=512, num_hidden_layers=2, num_attention_heads=4, vocab_size=512,
                             word_embed_proj_dim=128, max_position_embeddings=2048)),
}


def _match(name):
    n = name.lower().split("/")[-1].replace("_", "-")
    for key in NAMED:
        if n.startswith(key) or key in n:
            return key
    raise KeyError(f"no shape-faithful config for '{name}'; known: {sorted(NAMED)}")


def named_config(name, **overrides):
    from transformers import LlamaConfig, OPTConfig
    fam, kw = NAMED[_match(name)]
    kw = {**kw, **overrides}
    cfg = LlamaConfig(**kw) if fam == "llama" else OPTConfig(**kw)
    cfg._name_or_path = name
    return cfg


def random_init_model(name, dtype=torch.float16, seed=233, device=None, **overrides):
    """device: where the parameters are created and initialised (None = CPU, the default the tests pin; "cuda" builds a 7B-shaped model
    in seconds instead of ~2 minutes of fp16 normal_() on the host — a different, equally valid random stream)."""
    import contextlib
    from transformers import AutoModelForCausalLM
    cfg = named_config(name, **overrides)
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with (torch.device(device) if device is not None else contextlib.nullcontext()):
            model = AutoModelForCausalLM.from_config(cfg)
    finally:
        torch.set_default_dtype(prev)
    model.config._name_or_path = name
    model.eval()
    return model


def hide_lm_head(model):
    """keep lm_head out of the isinstance(nn.Linear) selection without changing its arithmetic"""
    import torch.nn as nn

    class _Head(nn.Module):
        def __init__(self, lin):
            super().__init__()
            self.weight, self.bias = lin.weight, lin.bias

        def forward(self, x):
            return nn.functional.linear(x, self.weight, self.bias)

    if hasattr(model, "lm_head") and isinstance(model.lm_head, nn.Linear):
        model.lm_head = _Head(model.lm_head)
    return model
