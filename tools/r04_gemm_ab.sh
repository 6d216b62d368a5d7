#!/bin/bash
# GPU tool: A/B of gemm4p_kernel's A-tile requests on ONE box: libxnacall_g0.so = the builtin (hipcc then waits lgkmcnt(0) for every
# A fragment), libxnacall.so = inline asm (counted waits); default precision (three-product GEMMs) and the f16f8 opt-in
out=gpurun_out/r04/gemm_dma_ab.txt
mkdir -p gpurun_out/r04
{
for rep in 1 2; do
for v in _g0 ""; do
 for prec in mixed f16f8; do
  XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall$v.so python bench.py --steps 8 --warmup 2 --cpu-chunks 0 --precision $prec 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('variant %-12s %-6s paired   %.2f ms/step  %.3e  stage in %.1f rec %.1f lin %.1f conv %.1f' % ('${v:-asm(default)}', '$prec', d['ms_per_step'], d['value'], s['lstm_in'], s['lstm_rec'], s['linear'], s['conv']))"
  XB_OVERLAP=0 XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall$v.so python bench.py --steps 4 --warmup 2 --cpu-chunks 0 --precision $prec 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('variant %-12s %-6s serial   %.2f ms/step  stage in %.1f rec %.1f lin %.1f conv %.1f' % ('${v:-asm(default)}', '$prec', d['ms_per_step'], s['lstm_in'], s['lstm_rec'], s['linear'], s['conv']))"
 done
done; done
} > $out 2>&1
cat $out
