"""Full-model CPU baseline of the decomposition stage: the oracle pipeline — scale + torch.linalg.svd (LAPACK gesdd) +
truncate/split, i.e. what the reference's binary_search.py:111-131 loop costs with the exact SVD — on the GPU box's HOST cores.
  * opt-125m: every one of its 73 Linears measured (synthetic weights of the real shapes);
  * Llama-2-7b: one measurement per distinct shape (median of --reps) x the number of Linears of that shape (225 in total).
Writes one JSON (default profiles/r2_cpu_full_model.json).  Test/measurement infrastructure: imports oracle/, never the product path."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import asvd_oracle as O
from bench import synth


def one(m, n, ratio, seed):
    W, scal = synth(m, n, seed)
    s = O.make_scale(scal, 0.5)
    r = max(1, int(m * n * ratio) // (m + n))
    t0 = time.perf_counter()
    ws = O.scaled_weight(W, s)
    U, S, V = O.exact_svd(ws)
    O.truncate_split(U, S, V, s, r, "UV", torch.float16)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="profiles/r2_cpu_full_model.json")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    one(768, 768, 0.9, 1)  # warm-up
    res = {"host_cpu_count": os.cpu_count(), "threads": a.threads, "ratio": 0.9,
           "what": "oracle scale + torch.linalg.svd(full_matrices=False) + truncate/split per Linear, synthetic weights of the model's shapes"}
    # opt-125m: 12 layers x (4 x 768x768, 3072x768, 768x3072) + lm_head 50272x768 — every layer measured
    t_opt, n_opt = 0.0, 0
    for layer in range(12):
        for (m, n) in [(768, 768)] * 4 + [(3072, 768), (768, 3072)]:
            t_opt += one(m, n, 0.9, 1000 + n_opt); n_opt += 1
    t_head = one(50272, 768, 0.9, 4242); t_opt += t_head; n_opt += 1
    res["opt-125m"] = {"linears": n_opt, "seconds_total_measured": t_opt, "seconds_lm_head": t_head}
    # Llama-2-7b: per-shape median x count
    shapes = {"4096x4096": (4096, 4096, 128), "11008x4096": (11008, 4096, 64), "4096x11008": (4096, 11008, 32), "32000x4096": (32000, 4096, 1)}
    tot, per = 0.0, {}
    for name, (m, n, cnt) in shapes.items():
        ts = sorted(one(m, n, 0.9, 7 + i) for i in range(a.reps if cnt > 1 else 1))
        med = ts[len(ts) // 2]
        per[name] = {"seconds_median": med, "runs": ts, "count": cnt}
        tot += med * cnt
    res["llama-2-7b"] = {"linears": sum(c for _, _, c in shapes.values()), "per_shape": per, "seconds_total_per_shape_times_count": tot}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
