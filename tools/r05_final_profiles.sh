# The command list behind profiles/r05_* (run from the repo root on the GPU box; one call may run 1200 s at most, hence PART):
#   PART=bench : bench lines of the default build (configs[2], configs[1], the per-GPU workloads of configs[3] / [4], the peaky
#                model, ragged batches, the serial schedule, every call on its own, the f16f8 opt-in, the driver's --steps 20)
#   PART=prof  : rocprofv3 kernel-trace stats of the bench command + the PMC passes (FETCH_SIZE / WRITE_SIZE for roofline.traffic;
#                MFMA-busy / clock / L2 / LDS for the GEMM, the recurrence and the decode), default and serial schedule
#   PART=prof_serial: the PMC passes under the serial schedule (XB_OVERLAP=0)
#   PART=stamps: the recurrence's cycle stamps (diagnostic library)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
if [ "$PART" = "bench" ]; then
timeout -k 10 300 python bench.py --steps 6 --warmup 2 > $O/bench_nb6.json 2> $O/bench.err
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --nbase 5 --cpu-chunks 0 > $O/bench_nb5.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch 1024 --cpu-chunks 0 > $O/bench_n1024.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch 2048 --cpu-chunks 0 > $O/bench_n2048.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --weights peaky --cpu-chunks 0 > $O/bench_nb6_peaky.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch 448 --cpu-chunks 0 > $O/bench_n448.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch 98 --cpu-chunks 0 > $O/bench_n98.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch 640 --cpu-chunks 0 > $O/bench_n640.json 2>> $O/bench.err
XB_OVERLAP=0 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --cpu-chunks 0 > $O/bench_nb6_serial.json 2>> $O/bench.err
XB_FUSE=0 timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-chunks 0 > $O/bench_nb6_nofuse.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-chunks 0 --precision f16f8 > $O/bench_nb6_f16f8_optin.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-chunks 0 --precision f16x3 > $O/bench_nb6_f16x3.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-chunks 0 > $O/bench_nb6_steps20.json 2>> $O/bench.err
fi
if [ "$PART" = "stamps" ]; then
(export XNA_LIBXNACALL=$R/xna_basecaller_amd/libxnacall_diag.so PREC=4 XB_OVERLAP=0; N=512 timeout -k 10 200 python tools/lstm_stamps.py > $O/lstm_stamps_single.txt 2>&1; N=1024 timeout -k 10 200 python tools/lstm_stamps.py > $O/lstm_stamps_dual.txt 2>&1) || echo "stamps failed (stale diagnostic library? make -C xna_basecaller_amd/csrc diag)" >> $O/bench.err
fi
if [ "$PART" = "prof" ]; then
rm -rf $O/stats512
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats512 -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-chunks 0 > $O/stats512.log 2>&1
cd $R
bash tools/pmc_gemm.sh r05/pmc512 > $O/pmc512.log 2>&1
fi
if [ "$PART" = "prof_serial" ]; then
XB_OVERLAP=0 bash tools/pmc_gemm.sh r05/pmc512_serial > $O/pmc512_serial.log 2>&1
fi
echo done > $O/done_$PART.txt
