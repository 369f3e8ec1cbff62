"""CPU restatement ("oracle") of the ASVD4LLM activation-aware SVD compression path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the
product package `asvd4llm_amd`, which fails loudly without its HIP library instead of falling back to this file.

Each function restates one piece of the reference (hahnyuan/ASVD4LLM @ 2024-10-24) and cites the lines it follows.
Arithmetic: numpy for the integer / rounding-sensitive pieces, torch-CPU (MKL LAPACK gesdd) for the exact SVD that
BASELINE.json names as the parity oracle: `torch.linalg.svd(W.float() * s, full_matrices=False)`.

Pinning: the reference ships no tests or golden vectors for this path.  This file is pinned
against outputs of the reference itself, produced in the build container by importing /root/reference
(oracle/make_golden.py) and committed under tests/golden/; tests/test_oracle_golden.py replays them.
Where the reference calls the randomized torch.svd_lowrank (modules/svd_linear.py:65) the oracle deliberately uses the
exact SVD (north_star); against the stock call only the one-sided Eckart-Young bound is checked.
"""
import math

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------------
# a2  rank arithmetic — modules/svd_linear.py:39-44
def synth_linear_numpy(out_features, in_features, seed, n_calib=16):
    """Seed-regenerated 'LLM-like' Linear for the mid-size fixtures (tests/golden/svd_mid.*): numpy PCG64 streams are
    platform-stable, so the fixture stores only a checksum of the inputs instead of megabytes of weights.
    Returns (W fp16 [out,in], scaling_diag_matrix fp16 [in]) as torch tensors."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W = (rng.standard_normal((out_features, in_features)) * 0.02).astype(np.float32)
    k = max(1, int(0.005 * in_features))
    W[:, rng.permutation(in_features)[:k]] *= 20
    scal = (n_calib * np.abs(rng.standard_normal(in_features))).astype(np.float32)
    k = max(1, int(0.01 * in_features))
    scal[rng.permutation(in_features)[:k]] *= 30
    return torch.from_numpy(W).to(torch.float16), torch.from_numpy(scal).to(torch.float16)


def tensor_checksum(*tensors):
    import hashlib
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def rank_from_ratio(out_features, in_features, param_ratio, rank_align=1):
    n_params = out_features * in_features
    compressed_params = int(n_params * param_ratio)
    rank = compressed_params // (in_features + out_features)
    rank = int(np.ceil(rank / rank_align) * rank_align)
    return rank


# ---------------------------------------------------------------------------------------------------------------
# a1  calibration hook — act_aware_utils.py:64-74 (accumulator starts as python int 0, :80)
def hook_update(acc, x, method):
    """acc: None (first call) or torch tensor [C]; x: torch tensor [..., T, C] with leading dims of size 1.
    Returns the new accumulator exactly as the reference hook computes it (same torch-CPU ops)."""
    if "abs_mean" in method:
        abs_mean = x.abs().mean(dim=-2).detach().view(-1)
        return abs_mean if acc is None else acc + abs_mean  # 0 + t == t
    elif "abs_max" in method:
        abs_max = x.abs().amax(dim=-2).detach().view(-1)
        if acc is None:
            acc = torch.zeros_like(abs_max)  # where(abs_max > 0, abs_max, 0) on the python int 0
        return torch.where(abs_max > acc, abs_max, acc)
    raise ValueError(method)


def fisher_update(acc, grad):
    """act_aware_utils.py:30: `module.fisher_info += module.weight.grad.detach().pow(2).mean(0)` (accumulator starts as python 0)"""
    g2 = grad.detach().pow(2).mean(0)
    return g2 if acc is None else acc + g2


def fisher_finish(acc, n_batches):
    """act_aware_utils.py:33-35: `module.fisher_info = module.fisher_info.div(len(calib_loader)).sqrt()`"""
    return acc.div(n_batches).sqrt()


def hook_update_numpy(acc, x, method):
    """Independent numpy restatement of the same update (float64 column sums, one rounding to the activation dtype
    for .mean(), one for the += ), used to bound the rounding freedom of the fp32-accumulating device kernel."""
    x2 = np.asarray(x).reshape(-1, x.shape[-1])
    dt = x2.dtype
    a = np.abs(x2.astype(np.float64))
    if "abs_mean" in method:
        mean = (a.sum(axis=0) / x2.shape[0]).astype(dt)
        return mean if acc is None else (acc.astype(np.float32) + mean.astype(np.float32)).astype(dt)
    mx = a.max(axis=0).astype(dt)
    if acc is None:
        acc = np.zeros_like(mx)
    return np.where(mx > acc, mx, acc)


# ---------------------------------------------------------------------------------------------------------------
# a3  scale vector + scaled weight — modules/svd_linear.py:47-60
def make_scale(scaling_diag_matrix, alpha, fisher_info=None):
    s = 1
    s = s * scaling_diag_matrix ** alpha
    if fisher_info is not None:
        s = s * fisher_info ** alpha
    s = s + 1e-6
    return s


def scaled_weight(weight, s):
    w = weight.float()
    if s is not None:
        w = w * s.view(1, -1)
    return w


# ---------------------------------------------------------------------------------------------------------------
# a4  factorisation (exact oracle) — replaces modules/svd_linear.py:65
def exact_svd(w):
    """economy SVD on CPU fp32 (LAPACK gesdd through torch).  Returns U [m,k], S [k] descending, V [n,k]."""
    U, S, Vh = torch.linalg.svd(w.cpu().float(), full_matrices=False)
    return U, S, Vh.transpose(0, 1).contiguous()


# ---------------------------------------------------------------------------------------------------------------
# a5  un-scale, sigma fusion, cast — modules/svd_linear.py:69-70, :8-24, :101-102
def truncate_split(U, S, V, s, rank, sigma_fuse, out_dtype):
    U, S, V = U[:, :rank], S[:rank], V[:, :rank]
    if s is not None:
        V = V / s.view(-1, 1)
    if sigma_fuse == "UV":
        A = U.mul(S.sqrt()).contiguous()
        B = V.t().mul(S.sqrt().view(-1, 1)).contiguous()
    elif sigma_fuse == "U":
        A = U.mul(S).contiguous()
        B = V.t().contiguous()
    elif sigma_fuse == "V":
        A = U.contiguous()
        B = V.t().mul(S.view(-1, 1)).contiguous()
    else:
        raise ValueError(sigma_fuse)
    nan = [bool((S != S).any()), bool((U != U).any()), bool((V != V).any())]
    return A.to(out_dtype), B.to(out_dtype), nan


def from_linear_oracle(weight, scaling_diag_matrix, param_ratio, alpha=1, act_aware=False, sigma_fuse="UV", rank_align=1,
                       fisher_info=None):
    """The whole of SVDLinear.from_linear (svd_linear.py:26-103) with the exact SVD.  Returns dict."""
    out_f, in_f = weight.shape
    rank = rank_from_ratio(out_f, in_f, param_ratio, rank_align)
    s = None
    if act_aware and (scaling_diag_matrix is not None or fisher_info is not None):
        # svd_linear.py:48-59: s = 1 * scaling**alpha (* fisher**alpha), then += 1e-6
        s = 1
        if scaling_diag_matrix is not None:
            s = s * scaling_diag_matrix ** alpha
        if fisher_info is not None:
            s = s * fisher_info ** alpha
        s = s + 1e-6
    w = scaled_weight(weight, s)
    U, S, V = exact_svd(w)
    A, B, nan = truncate_split(U, S, V, s, rank, sigma_fuse, weight.dtype)
    return {"rank": rank, "s": s, "w_scaled": w, "U": U, "S": S, "V": V, "A": A, "B": B, "nan": nan}


# ---------------------------------------------------------------------------------------------------------------
# a8  stable-rank sensitivity — sensitivity.py:96-107 (on the UNSCALED weight; result dtype follows torch promotion)
STABLE_RANK_RATIOS = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]


def stable_rank_sensitivity(weight):
    w = weight
    w_fro = torch.norm(w, p="fro") ** 2
    singular_values = torch.linalg.svdvals(w.float())
    spectral_norm = torch.max(singular_values)
    w_spec = spectral_norm ** 2
    sr = (w_fro / w_spec) ** 0.5
    return {r: -sr * r ** 0.1 for r in STABLE_RANK_RATIOS}


# ---------------------------------------------------------------------------------------------------------------
# a9  calibration perplexity — evaluate_utils.py:90-115 (mean over T-1 tokens times seqlen=T; reproduce verbatim)
def perplexity_from_logits(logits_list, dataset, limit):
    nsamples, seqlen = dataset.shape
    nlls = []
    for i in range(nsamples):
        if i == limit:
            break
        labels = dataset[i:i + 1, 1:].contiguous()
        logits = logits_list[i]
        loss = torch.nn.functional.cross_entropy(logits.view(-1, logits.size(-1)), labels.view(-1))
        nlls.append(loss.float() * seqlen)
    return torch.exp(torch.stack(nlls).sum() / (len(nlls) * seqlen)).item()


# ---------------------------------------------------------------------------------------------------------------
# a10  rank allocation by binary search — binary_search.py:29-110 (ratio-target branch; pure python arithmetic)
def binary_search_ratios(sensitivity_dict, numel, param_ratio_target=-1.0, compress_kv_cache=False, kv_cache_ratio_target=-1.0):
    """sensitivity_dict {layer: {ratio: ppl}} in insertion order; numel {layer: weight.numel()}.
    Returns (layers_min_ratio, trace_lines) exactly as the reference prints / decides (incl. the last-`mid` quirk)."""
    if compress_kv_cache:
        ratio_target = kv_cache_ratio_target
        sensitivity_dict = {k: v for k, v in sensitivity_dict.items() if "k_proj" in k or "v_proj" in k}
        default_param_ratio = 2
    else:
        ratio_target = param_ratio_target
        default_param_ratio = 1
    sensitivity_list = []
    for layername, v in sensitivity_dict.items():
        for param_ratio, ppl in v.items():
            if not compress_kv_cache and param_ratio >= 1:
                continue
            sensitivity_list.append((layername, param_ratio, ppl))
    sorted_sensitive_list = sorted(sensitivity_list, key=lambda x: -x[2])
    high = len(sorted_sensitive_list) - 1
    low = 0
    trace = []
    mid = None
    while low < high:
        mid = (low + high) // 2
        layers_min_ratio = {layername: default_param_ratio for layername in sensitivity_dict.keys()}
        for layername, param_ratio, ppl in sorted_sensitive_list[mid:]:
            layers_min_ratio[layername] = min(layers_min_ratio[layername], param_ratio)
        tot_params = 0
        compress_params = 0
        for layername, param_ratio in layers_min_ratio.items():
            tot_params += numel[layername]
            compress_params += numel[layername] * param_ratio
        now_ratio = compress_params / tot_params
        if compress_kv_cache:
            now_ratio /= 2
        trace.append(f"low={low} mid={mid}, high={high}, now_ratio={now_ratio}, params=({compress_params}/{tot_params})")
        if now_ratio > ratio_target:
            high = mid
        else:
            low = mid + 1
    layers_min_ratio = {layername: default_param_ratio for layername in sensitivity_dict.keys()}
    for layername, param_ratio, ppl in sorted_sensitive_list[mid:]:
        layers_min_ratio[layername] = min(layers_min_ratio[layername], param_ratio)
    return layers_min_ratio, trace


# ---------------------------------------------------------------------------------------------------------------
# parity metrics used by the tests
def sigma_rel_err(S_test, S_ref, r):
    S_test = torch.as_tensor(S_test).double().cpu()[:r]
    S_ref = torch.as_tensor(S_ref).double().cpu()[:r]
    return ((S_test - S_ref).abs() / S_ref).max().item()


def recon_rel_err(A, B, R_ref, W):
    """|A B - R_ref|_F / |W|_F in float64"""
    P = A.double().cpu() @ B.double().cpu()
    return ((P - R_ref.double()).norm() / W.double().norm()).item()


def live_channels(s, rel=1e-3):
    """Input channels whose scale is not numerically dead.  For a dead channel (statistic 0 -> s = 1e-6) the scaled column
    W[:, i] * s_i lies below fp32 epsilon relative to sigma_1, so V[i, :] is rounding noise in ANY fp32 SVD (LAPACK included)
    and the reference's un-scaling `V / s` (svd_linear.py:70) amplifies that noise by 1e6: the reference's own B[:, i] is not
    reproducible across LAPACK builds.  Parity on those columns is therefore measured in the scaled norm only."""
    if s is None:
        return None
    s = torch.as_tensor(s).double().cpu().flatten()
    return s >= rel * s.max()


def recon_parity(A, B, A_ref, B_ref, W, s):
    """(unscaled error on live channels / |W|_F, scaled error on all channels / |W diag(s)|_F), float64"""
    P = A.double().cpu() @ B.double().cpu()
    P_ref = A_ref.double().cpu() @ B_ref.double().cpu()
    Wd = W.double().cpu()
    D = P - P_ref
    if s is None:
        e = (D.norm() / Wd.norm()).item()
        return e, e
    live = live_channels(s)
    sd = torch.as_tensor(s).double().cpu().flatten()
    e_live = (D[:, live].norm() / Wd.norm()).item()
    e_scaled = ((D * sd.view(1, -1)).norm() / (Wd * sd.view(1, -1)).norm()).item()
    return e_live, e_scaled
