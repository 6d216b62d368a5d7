#!/bin/bash
# GPU tool: the schedule knobs again on the round-5 build (the paired recurrence got 6 % shorter: does the optimum move?)
mkdir -p gpurun_out/r05
run() { env "$@" timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-chunks 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s %7.2f ms/step  rec %.2f ms  %s' % ('$*', d['ms_per_step'], d['roofline']['avg_launch_ms'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}))"; }
{
run XB_TIME_SLABS=16
run XB_TIME_SLABS=8
run XB_TIME_SLABS=32 XB_SLAB_STEPS=60
run XB_GEMM_SHADOW_WGS=1
run XB_GEMM_SHADOW=8
run XB_DECODE_ASYNC=1
run XB_LSTM_DEFER=1
run XB_TIME_SLABS=16
} 2>&1 | tee gpurun_out/r05/schedule_knobs.txt
