# same-box A/B of the software-pipelined recurrence (xb_lstm_quad.h) against lstm_kernel<48, 2, DUAL> (XB_LSTM_QUAD=0):
# the default bench (two calls per device pass), batch 1024 / 2048 single calls, the serial schedule, and the cycle stamps
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04q; mkdir -p $O
cd $R
for q in 0 1 0 1; do
  XB_LSTM_QUAD=$q timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-chunks 0 > $O/bench_q${q}_$RANDOM.json 2>> $O/bench.err
done
for q in 0 1; do
  XB_LSTM_QUAD=$q XB_OVERLAP=0 timeout -k 10 200 python bench.py --steps 4 --warmup 2 --cpu-chunks 0 > $O/bench_serial_q${q}.json 2>> $O/bench.err
  XB_LSTM_QUAD=$q timeout -k 10 200 python bench.py --steps 4 --warmup 2 --cpu-chunks 0 --batch 2048 > $O/bench_n2048_q${q}.json 2>> $O/bench.err
  XB_LSTM_QUAD=$q timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-chunks 0 --precision f16f8 > $O/bench_f16f8_q${q}.json 2>> $O/bench.err
done
(export XNA_LIBXNACALL=$R/xna_basecaller_amd/libxnacall_diag.so PREC=2 XB_OVERLAP=0; N=1024 timeout -k 10 200 python tools/lstm_stamps.py > $O/lstm_stamps_quad.txt 2>&1; XB_LSTM_QUAD=0 N=1024 timeout -k 10 200 python tools/lstm_stamps.py > $O/lstm_stamps_dual.txt 2>&1)
echo done > $O/done_ab.txt
