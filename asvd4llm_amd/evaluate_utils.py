"""evaluate_perplexity — the calibration perplexity used inside the sweep and the ppl-target search (evaluate_utils.py:90-115).
Reproduced verbatim including the mean-over-(T-1)-times-T quirk.  Model forwards are ordinary
PyTorch-ROCm execution; only linalg is hand-written in this build."""
import torch
import torch.nn as nn


@torch.no_grad()
def evaluate_perplexity(model, dataset, limit):
    """dataset: input ids tensor of shape [batch, sequence length]"""
    nsamples, seqlen = dataset.size()
    nlls = []
    for i in range(nsamples):
        if i == limit:
            break
        input_ids = dataset[i:i + 1, :-1].to(model.device)
        labels = dataset[i:i + 1, 1:].contiguous()
        logits = model(input_ids=input_ids)[0]
        shift_logits = logits[:, :, :]
        shift_labels = labels.to(model.device)
        loss_fct = nn.CrossEntropyLoss()
        loss = loss_fct(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1))
        neg_log_likelihood = loss.float() * seqlen
        nlls.append(neg_log_likelihood)
    ppl = torch.exp(torch.stack(nlls).sum() / (len(nlls) * seqlen))
    return ppl.item()


@torch.no_grad()
def evaluate_model(model, tokenizer, model_name, tasks, eval_ppl="", num_fewshot=0, limit=-1, batch_size=1, use_bos=False, eval_ids=None):
    """Accuracy evaluation (wikitext2/ptb ppl, MMLU, lm-eval tasks; evaluate_utils.py:118-226) needs `lm_eval` and network
    datasets — out of scope here.  When `eval_ids` ([n, T] token ids) is given its perplexity is reported instead."""
    results = {}
    if eval_ids is not None:
        results["synthetic_ppl"] = evaluate_perplexity(model, eval_ids, limit if limit > 0 else eval_ids.size(0))
    return results
