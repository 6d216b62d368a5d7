#!/bin/bash
# GPU tool: serial-order stage times of several builds of the library on one box:  tools/r05_variants_serial.sh tag lib1 lib2 ...
tag=$1; shift
mkdir -p gpurun_out/r05
for lib in "$@"; do
  XB_OVERLAP=0 XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall$lib.so timeout -k 10 200 python bench.py --steps 4 --warmup 2 --cpu-chunks 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-14s serial %7.2f ms/step  %s' % ('$lib', d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}))"
done 2>&1 | tee gpurun_out/r05/variants_serial_$tag.txt
