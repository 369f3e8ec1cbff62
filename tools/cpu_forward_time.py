"""CPU forward time of a model-shaped network on the GPU box's HOST cores: the other half of the denominator of BASELINE.json's ">= 20x the
CPU-reference wall-clock" (VERDICT r3 8b).  The reference's ppl sweep runs #Linears x 6 x n_calib full-model forwards of 2047 tokens
(sensitivity.py:43-59, evaluate_utils.py:90-115); this measures ONE such forward (no_grad, logits for all tokens, the cross-entropy the
evaluator takes) of a shape-faithful random-init model built on the GPU (fast) and moved to the host, at the best of a few thread counts.
fp32 would be the faithful CPU dtype of an fp16 checkpoint (CPU fp16 GEMMs are not a thing); bf16 halves memory traffic and is what a
patient user would pick — both are reported when --dtypes says so.  Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-2-7b")
    ap.add_argument("--tokens", type=int, default=2047)
    ap.add_argument("--dtypes", default="bfloat16,float32")
    ap.add_argument("--threads", default="64,128,32")
    ap.add_argument("--layers", type=int, default=None, help="build only this many decoder layers and scale (memory-bounded hosts)")
    a = ap.parse_args()
    from asvd4llm_amd.model_zoo import random_init_model, NAMED, _match
    full_layers = NAMED[_match(a.model)][1]["num_hidden_layers"]
    kw = {} if a.layers is None else {"num_hidden_layers": a.layers}
    dev = "cuda" if torch.cuda.is_available() else None
    model = random_init_model(a.model, dtype=torch.float16, device=dev, **kw).to("cpu")
    ids = torch.randint(0, model.config.vocab_size, (1, a.tokens), generator=torch.Generator().manual_seed(3))
    out = {"model": a.model, "tokens": a.tokens, "layers_built": a.layers or full_layers, "layers_full": full_layers, "host_cpu_count": os.cpu_count(), "results": []}
    for dt in a.dtypes.split(","):
        m = model.to(getattr(torch, dt))
        best = None
        for th in [int(x) for x in a.threads.split(",")]:
            if th > (os.cpu_count() or th):
                continue
            torch.set_num_threads(th)
            with torch.no_grad():
                t0 = time.perf_counter()
                logits = m(input_ids=ids)[0]
                loss = torch.nn.functional.cross_entropy(logits[0, :-1].float(), ids[0, 1:])
                dt_s = time.perf_counter() - t0
            if best is None or dt_s < best[1]:
                best = (th, dt_s)
        scale = full_layers / (a.layers or full_layers)
        out["results"].append({"dtype": dt, "threads": best[0], "seconds_per_forward": best[1], "seconds_per_forward_full_model_est": best[1] * scale if a.layers else best[1],
                               "loss": float(loss)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
