#!/bin/bash
for cfg in "B=16" "B=15" "B=16 ASVD_PIPE=0"; do
echo "== $cfg"
env $cfg timeout 600 python - <<'PY' 2>&1 | grep -E "^SW"
import torch, sys, os
sys.path.insert(0, ".")
from bench import synth
from asvd4llm_amd import ops
mats, scs = [], []
for b in range(int(os.environ["B"])):
    W, scal = synth(4096, 4096, 233 + b)
    mats.append(W.cuda()); scs.append(ops.make_scale(scal.cuda(), alpha=0.5))
for rep in range(2):
    U, S, V, infos = ops.svd_batched(mats, scs)
    print("SW", [i.sweeps for i in infos], file=sys.stderr)
PY
done
