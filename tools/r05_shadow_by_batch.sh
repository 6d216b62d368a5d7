#!/bin/bash
# GPU tool: which GEMM kernel for the slabs beside the recurrence, by batch size, on the current build (XB_GEMM_SHADOW=0 auto / 4 gemm4p / 8 gemm8r)
out=${1:-gpurun_out/r05/shadow_by_batch.txt}
mkdir -p "$(dirname "$out")"
for n in 2048 1024; do
  for sh in 0 4 8; do
    XB_GEMM_SHADOW=$sh timeout -k 10 300 python bench.py --steps 4 --warmup 2 --batch $n --cpu-chunks 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $n XB_GEMM_SHADOW=$sh %8.2f ms/step  rec %.2f ms  %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}))"
  done
done 2>&1 | tee "$out"
