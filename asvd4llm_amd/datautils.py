"""Calibration data.  The reference samples wikitext2/c4/ptb/alpaca/selfgen through HF `datasets` (datautils.py:106-160);
this image has no network, so the build adds a `synthetic` dataset (seeded uniform token ids of the same [1, 2048] shape and
the same list-of-dict layout / cache file convention).  Other names are attempted through `datasets` and fail loudly."""
import os

import torch


def get_calib_data(name, tokenizer, model_id, nsamples, seqlen=2048, seed=3, use_bos=False, vocab_size=None):
    cache_file = f"cache/{name}_{model_id.replace('/','_')}_{nsamples}_{seqlen}_{seed}_bos{use_bos}.pt"
    from . import parallel
    if parallel.cache_exists(cache_file):  # one answer for all ranks; the file is complete (written under a temporary name, then renamed)
        return parallel.load_cache(cache_file)
    if name != "synthetic":
        raise RuntimeError(f"calibration dataset '{name}' needs HF datasets + network, unavailable here; use --calib_dataset synthetic")
    if vocab_size is None:
        vocab_size = getattr(tokenizer, "vocab_size", None) or 32000
    g = torch.Generator().manual_seed(seed)
    traindataset = []
    for _ in range(nsamples):
        inp = torch.randint(0, vocab_size, (1, seqlen), generator=g)
        if use_bos and getattr(tokenizer, "bos_token_id", None) is not None:
            inp[0, 0] = tokenizer.bos_token_id
        traindataset.append({"input_ids": inp, "attention_mask": torch.ones_like(inp)})
    parallel.save_cache(traindataset, cache_file)  # the samples are a function of the seed: every rank built the same list, rank 0 writes it
    return traindataset
