set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v3; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > $O/bench_nb6.json 2> $O/bench_nb6.err
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --nbase 5 --cpu-chunks 0 > $O/bench_nb5.json 2>> $O/bench_nb6.err
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch 1024 --cpu-chunks 0 > $O/bench_n1024.json 2>> $O/bench_nb6.err
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --batch 2048 --cpu-chunks 0 > $O/bench_n2048.json 2>> $O/bench_nb6.err
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch 1024 --cpu-chunks 0 --precision f16f8i > $O/bench_n1024_f16f8i.json 2>> $O/bench_nb6.err
XB_LSTM_DUAL=0 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch 1024 --cpu-chunks 0 > $O/bench_n1024_single.json 2>> $O/bench_nb6.err
(export XNA_LIBXNACALL=$R/xna_basecaller_amd/libxnacall_diag.so PREC=2 XB_OVERLAP=0; N=1024 timeout -k 10 200 python tools/lstm_stamps.py > $O/lstm_stamps_dual.txt 2>&1; N=512 timeout -k 10 200 python tools/lstm_stamps.py > $O/lstm_stamps_single.txt 2>&1)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats512 -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-chunks 0 > $O/stats512.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1024 -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-chunks 0 --batch 1024 > $O/stats1024.log 2>&1
cd $R
PASSES=2 bash tools/pmc_gemm.sh v3/pmc512 > $O/pmc512.log 2>&1
echo done > $O/done.txt
