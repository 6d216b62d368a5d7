#!/bin/bash
# GPU tool: A/B of the recurrence's MFMA-loop variants on ONE box (libxnacall_e{asm}{spread}.so built on the CPU side):
#   asm    = the exchange pieces' LDS-DMA requests as inline asm, so that hipcc keeps counting lgkmcnt (XB_LSTM_DMA_ASM)
#   spread = every request directly behind one MFMA instead of in pairs (XB_LSTM_DMA_SPREAD)
out=gpurun_out/r04/lstm_ab.txt
mkdir -p gpurun_out/r04
{
for rep in 1 2; do
for v in _e00 _e01 _e10 ""; do
  XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall$v.so python bench.py --steps 8 --warmup 2 --cpu-chunks 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('variant %-14s paired   %.2f ms/step  rec %.2f ms/launch (frac %.4f)  stage rec %.1f in %.1f' % ('${v:-_e11(default)}', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], s['lstm_rec'], s['lstm_in']))"
  XB_OVERLAP=0 XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall$v.so python bench.py --steps 4 --warmup 2 --cpu-chunks 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('variant %-14s serial   %.2f ms/step  rec %.2f ms/launch  stage rec %.1f in %.1f' % ('${v:-_e11(default)}', d['ms_per_step'], d['roofline']['avg_launch_ms'], s['lstm_rec'], s['lstm_in']))"
done; done
} > $out 2>&1
cat $out
