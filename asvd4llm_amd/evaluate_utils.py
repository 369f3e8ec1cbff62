"""Calibration perplexity used inside the sweep and the ppl-target search (reference: evaluate_utils.py:90-115).

Own implementation of the same quantity, including the reference's normalisation quirk: each sample
contributes  mean-over-(T-1)-predicted-tokens x T  and the total is divided by  n x T  — not by the number of predicted tokens.
The arithmetic keeps the reference's rounding points (mean cross entropy in the logits' dtype, up-cast, times T; one fp32
vector sum over the samples; one exp), so the value is the reference's to the last bit on the same logits
(tests/test_host_logic.py::test_perplexity_quirk_matches_reference).  Model forwards are ordinary PyTorch-ROCm execution."""
import torch
import torch.nn.functional as F


def _sample_nll(model, row, seqlen):
    """T x (mean next-token cross entropy of one calibration row), fp32 scalar on the model's device"""
    dev = model.device
    logits = model(input_ids=row[:, :-1].to(dev))[0]
    targets = row[:, 1:].to(dev).reshape(-1)
    return F.cross_entropy(logits.reshape(-1, logits.size(-1)), targets).float() * seqlen


@torch.no_grad()
def evaluate_perplexity(model, dataset, limit):
    """dataset: [n, T] token ids; the first `limit` rows are used (all of them when limit is negative or larger than n)."""
    n_rows, seqlen = dataset.size()
    n_used = n_rows if (limit < 0 or limit > n_rows) else limit
    per_sample = torch.stack([_sample_nll(model, dataset[i:i + 1], seqlen) for i in range(n_used)])  # empty -> RuntimeError, as the reference
    return torch.exp(per_sample.sum() / (n_used * seqlen)).item()


@torch.no_grad()
def evaluate_model(model, tokenizer, model_name, tasks, eval_ppl="", num_fewshot=0, limit=-1, batch_size=1, use_bos=False, eval_ids=None):
    """Accuracy evaluation (wikitext2/ptb ppl, MMLU, lm-eval tasks; evaluate_utils.py:118-226) needs `lm_eval` and network
    datasets — out of scope here.  When `eval_ids` ([n, T] token ids) is given its perplexity is reported instead."""
    results = {}
    if eval_ids is not None:
        results["synthetic_ppl"] = evaluate_perplexity(model, eval_ids, limit if limit > 0 else eval_ids.size(0))
    return results
