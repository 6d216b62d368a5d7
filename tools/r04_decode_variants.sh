#!/bin/bash
# GPU tool: decode alone under tuning builds of xb_decode.hip (ring depths XB_DEC_RDEPTH / XB_DEC_RDEPTH13, occupancy hint
# XB_DEC_WAVES_ATTR); the variants are built on the CPU side as xna_basecaller_amd/libxnacall_dv*.so
out=gpurun_out/r04/decode_variants.txt
mkdir -p gpurun_out/r04
{
for v in "" _dv1 _dv2 _dv3 _dv4 _dv5 _dv6; do
  for nb in 6 5; do for n in 512 1024; do
    XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall$v.so NB=$nb N=$n REPS=5 python tools/decode_sweeps.py 2>/dev/null | sed "s/^/variant ${v:-default} /"
  done; done
done
} > $out 2>&1
cat $out
