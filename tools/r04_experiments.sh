#!/bin/bash
# GPU tool (round 4): decode LDS-conflict bound, decode placement, schedule knobs under the mixed default.  Output: gpurun_out/r04/
out=gpurun_out/r04
mkdir -p $out
bench() { # label, env..., runs bench.py with the default (mixed) precision and prints step time + decode roofline
  label=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --cpu-chunks 0 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; q=d["roofline_decode"]; s=d["stage_ms_per_step"]
print("%-34s %7.2f ms/step %.3e samples/s | rec %.2f ms/launch frac %.4f | decode %.2f ms/launch frac %.3f | stage ms/step in %.1f rec %.1f lin %.1f conv %.1f dec %.1f" % (sys.argv[1], d["ms_per_step"], d["value"], r["avg_launch_ms"], r["frac"], q["avg_launch_ms"], q["frac"], s["lstm_in"], s["lstm_rec"], s["linear"], s["conv"], s["decode"]))' "$label"
}
{
  echo "# decode alone, diagnostic library: sweep 2 with lane-linear (conflict-free, WRONG results) LDS addresses vs the real ones"
  for n in 512 1024; do for lin in 0 1; do for stop in 0 2 1; do
    XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall_diag.so NB=6 N=$n XB_DECODE_LINEAR_LDS=$lin XB_DECODE_STOP=$stop python tools/decode_sweeps.py 2>/dev/null | sed "s/^/linear_lds=$lin /"
  done; done; done
} > $out/decode_lds_conflicts.txt 2>&1
{
  echo "# bench.py --steps 10 --warmup 2, default precision (mixed), batch 512, nb 6"
  bench "default" XB_NOP=1
  bench "XB_DECODE_ASYNC=0" XB_DECODE_ASYNC=0
  bench "default (again)" XB_NOP=1
  bench "XB_TIME_SLABS=8" XB_TIME_SLABS=8
  bench "XB_TIME_SLABS=32 XB_SLAB_STEPS=60" XB_TIME_SLABS=32 XB_SLAB_STEPS=60
  bench "XB_GEMM_SHADOW=8" XB_GEMM_SHADOW=8
  bench "XB_GEMM_SHADOW_WGS=1" XB_GEMM_SHADOW_WGS=1
  bench "XB_GEMM_SN=3" XB_GEMM_SN=3
  bench "XB_GEMM_SN=4" XB_GEMM_SN=4
  bench "XB_LSTM_SIGNAL=0" XB_LSTM_SIGNAL=0
  bench "XB_FUSE=0" XB_FUSE=0
  XB_X3_STAGES=0 bench "XB_X3_STAGES=0 (= f16f8 everywhere)" XB_NOP=1
} > $out/schedule_knobs.txt 2>&1
