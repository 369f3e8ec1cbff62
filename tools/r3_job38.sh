mkdir -p gpurun_out/r3_38
O=gpurun_out/r3_38
for cfg in "4096 1" "4096 4" "4096 8" "768 16" "5120 2" "256 4" "1024 2" "11008 1"; do
  set -- $cfg
  for q in 0 1; do timeout 300 python tools/check_evdq.py run $q $1 $2 $O/r_$1_$2_$q.npz 2>&1 | grep evdq | cut -c1-60; done
  python tools/check_evdq.py cmp $O/r_$1_$2_0.npz $O/r_$1_$2_1.npz
done
rm -f $O/*.npz
