#!/bin/bash
# GPU tool: serial-order and default-schedule step of several builds of the library on one box:  tools/r05_variants.sh tag lib1 lib2 ...
tag=$1; shift
mkdir -p gpurun_out/r05
for lib in "$@"; do
  for mode in serial overlapped; do
    if [ $mode = serial ]; then export XB_OVERLAP=0; st=4; else unset XB_OVERLAP; st=8; fi
    XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall$lib.so timeout -k 10 200 python bench.py --steps $st --warmup 2 --cpu-chunks 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-14s %-10s %7.2f ms/step  %s %.3f ms  %s' % ('$lib', '$mode', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}))"
  done
done 2>&1 | tee gpurun_out/r05/variants_$tag.txt
