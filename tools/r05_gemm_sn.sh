#!/bin/bash
# GPU tool: XB_GEMM_SN (N tiles per XCD super-tile of gemm4p_kernel) on the current build, serial order and default schedule
out=${1:-gpurun_out/r05/gemm_sn_s16.txt}
mkdir -p "$(dirname "$out")"
for sn in 2 4 6 2; do
  for mode in serial overlapped; do
    if [ $mode = serial ]; then export XB_OVERLAP=0; st=4; else unset XB_OVERLAP; st=8; fi
    XB_GEMM_SN=$sn timeout -k 10 200 python bench.py --steps $st --warmup 2 --cpu-chunks 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('SN=$sn %-10s %7.2f ms/step  %s' % ('$mode', d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}))"
  done
done 2>&1 | tee "$out"
