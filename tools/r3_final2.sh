#!/bin/bash
# Round-3 evidence refresh after the last library change (everything of tools/r3_final.sh except the two end-to-end pipeline runs, whose
# time is PyTorch forwards): tests, smoke, PMC + kernel-trace passes, bench (traffic of the same binary), shapes, full models, latency form.
set -u
O=gpurun_out/final4; mkdir -p $O
export ASVD_STRICT=1
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
PMC_BATCH=32 bash tools/prof_final.sh r3d > $O/prof.log 2>&1
cp gpurun_out/prof_r3d/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
bash tools/r2_job14.sh 2>&1 | tee $O/shapes.txt
python tools/full_model_bench.py --model llama-2-7b 2>/dev/null | tail -1 > $O/full_7b.json
python tools/full_model_bench.py --model llama-2-13b 2>/dev/null | tail -1 > $O/full_13b.json
python tools/full_model_bench.py --model opt-125m 2>/dev/null | tail -1 > $O/full_opt125m.json
python tools/bench_evd_wave.py > $O/evdw_micro.jsonl 2> /dev/null
for cfg in "4096 1" "768 16" "5120 2" "256 4" "1024 2"; do
  set -- $cfg
  for q in 0 1; do timeout 300 python tools/check_evdq.py run $q $1 $2 $O/r_$1_$2_$q.npz 2>&1 | grep evdq | cut -c1-110; done
  python tools/check_evdq.py cmp $O/r_$1_$2_0.npz $O/r_$1_$2_1.npz
done 2>&1 | tee $O/latency_form.txt
rm -f $O/*.npz
for f in full_7b full_13b full_opt125m; do echo "== $f"; python -c "
import json; x=json.load(open('$O/$f.json')); print(x['decompose_total_s'], x['all_layers_status_ok'], x['sweeps_min_max'])"; done
