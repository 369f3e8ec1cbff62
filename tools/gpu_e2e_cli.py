"""End-to-end CLI run on a shape-faithful random-init HF model (BASELINE configs[0] analogue on the GPU): prints stage timings."""
import os, sys, time, types, io, contextlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asvd
name = sys.argv[1] if len(sys.argv) > 1 else "tiny-llama"
n_calib = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.makedirs("gpurun_out/e2e", exist_ok=True); os.chdir("gpurun_out/e2e")
args = asvd.build_parser().parse_args(["--model_id", name, "--random_init", "--act_aware", "--alpha", "0.5", "--n_calib_samples", str(n_calib),
                                       "--calib_dataset", "synthetic", "--param_ratio_target", "0.9", "--scaling_method", "abs_mean"] + sys.argv[3:])
from asvd4llm_amd.act_aware_utils import calib_input_distribution
from asvd4llm_amd.binary_search import binary_search_truncation_rank
from asvd4llm_amd.datautils import get_calib_data
from asvd4llm_amd.sensitivity import calib_sensitivity_ppl
from asvd4llm_amd.evaluate_utils import evaluate_perplexity
from asvd4llm_amd.modules.svd_linear import SVDLinear
os.environ["ASVD_STRICT"] = "1"
torch.manual_seed(args.seed)
t = {}
t0 = time.time(); model, tok = asvd.build_model(args); torch.cuda.synchronize(); t["build_model"] = time.time() - t0
calib = get_calib_data("synthetic", tok, name, n_calib, seed=args.seed, vocab_size=model.config.vocab_size)
ids = torch.cat([c["input_ids"] for c in calib], 0)
buf = io.StringIO()
with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
    t0 = time.time(); ppl0 = evaluate_perplexity(model, ids, n_calib); torch.cuda.synchronize(); t["ppl_raw"] = time.time() - t0
    t0 = time.time(); calib_input_distribution(model, calib, "abs_mean", False); torch.cuda.synchronize(); t["hook_pass"] = time.time() - t0
    t0 = time.time(); sens = calib_sensitivity_ppl(model, calib, args, False); torch.cuda.synchronize(); t["sweep_incl_factorize"] = time.time() - t0
    t0 = time.time(); binary_search_truncation_rank(model, sens, calib, args); torch.cuda.synchronize(); t["search_and_decompose"] = time.time() - t0
    ppl1 = evaluate_perplexity(model, ids, n_calib)
nsvd = sum(1 for m in model.modules() if isinstance(m, SVDLinear))
tot = sum(p.numel() for p in model.parameters())
print(json.dumps({"model": name, "n_calib": n_calib, "fused_sweep": args.fused_sweep, "linears": len(sens), "svd_linears_after": nsvd, "ppl_raw": ppl0, "ppl_after": ppl1, "params_after": tot,
                  "timings_s": t, "trace_tail": [l for l in buf.getvalue().splitlines() if l.startswith("low=") or l.startswith("decompose")][-3:]}))
