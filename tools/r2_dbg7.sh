#!/bin/bash
env N=4096 B=6 ASVD_DBG_SELFTEST=1 $EXTRA ASVD_PIPE=0 timeout 600 python - <<'PY' 2>&1 | grep -E "selftest|^SW"
import torch, sys, os
sys.path.insert(0, ".")
from bench import synth
from asvd4llm_amd import ops
mats, scs = [], []
n = int(os.environ["N"])
for b in range(int(os.environ["B"])):
    W, scal = synth(n, n, 233 + b)
    mats.append(W.cuda()); scs.append(ops.make_scale(scal.cuda(), alpha=0.5))
U, S, V, infos = ops.svd_batched(mats, scs, max_sweeps=1)
PY
