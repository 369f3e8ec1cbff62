mkdir -p gpurun_out/r3_28
O=gpurun_out/r3_28
for cfg in "4096 1" "768 16"; do
  set -- $cfg
  for q in 0 1; do timeout 300 python tools/check_evdq.py run $q $1 $2 $O/r_$1_$2_$q.npz 2>&1 | grep evdq; done
  python tools/check_evdq.py cmp $O/r_$1_$2_0.npz $O/r_$1_$2_1.npz
done
rm -f $O/*.npz
