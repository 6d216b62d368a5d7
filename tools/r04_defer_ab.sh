# same-box A/B: the deferred arrival of the two-groups-per-workgroup recurrence (default) vs a drain of its own (XB_LSTM_DEFER_ARRIVE=0 build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
for rep in 1 2; do
  for v in libxnacall libxnacall_nodefer; do
    XNA_LIBXNACALL=$R/xna_basecaller_amd/$v.so timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-chunks 0 > $O/bench_${v}_$rep.json 2>> $O/bench.err
  done
done
for v in libxnacall libxnacall_nodefer; do
  XNA_LIBXNACALL=$R/xna_basecaller_amd/$v.so timeout -k 10 200 python bench.py --steps 4 --warmup 2 --cpu-chunks 0 --batch 2048 > $O/bench_n2048_${v}.json 2>> $O/bench.err
  XNA_LIBXNACALL=$R/xna_basecaller_amd/$v.so timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-chunks 0 --precision f16f8 > $O/bench_f16f8_${v}.json 2>> $O/bench.err
done
(export XNA_LIBXNACALL=$R/xna_basecaller_amd/libxnacall_diag.so PREC=2 XB_OVERLAP=0; N=1024 timeout -k 10 200 python tools/lstm_stamps.py > $O/lstm_stamps_dual.txt 2>&1)
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['ms_per_step'],2), r['kernel'], round(r['avg_launch_ms'],2), {k:round(v,1) for k,v in d['stage_ms_per_step'].items()})
"; done; cat $O/lstm_stamps_dual.txt
