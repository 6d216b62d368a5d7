#!/bin/bash
# GPU tool (round 4): which GEMM kernel under the mixed default (x3 feed-forward GEMMs are MFMA / L2-path balanced, where the
# 256 x 256 one-workgroup-per-CU kernel moves a third fewer operand bytes per flop), decode placement at the larger batches.
out=gpurun_out/r04
mkdir -p $out
bench() {
  label=$1; shift
  env "$@" python bench.py --steps 8 --warmup 2 --cpu-chunks 0 $BARGS 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; q=d["roofline_decode"]; s=d["stage_ms_per_step"]
print("%-40s %7.2f ms/step %.3e samples/s | rec %.2f ms/launch frac %.4f | decode %.2f ms/launch frac %.3f | stage ms/step in %.1f rec %.1f lin %.1f conv %.1f dec %.1f" % (sys.argv[1], d["ms_per_step"], d["value"], r["avg_launch_ms"], r["frac"], q["avg_launch_ms"], q["frac"], s["lstm_in"], s["lstm_rec"], s["linear"], s["conv"], s["decode"]))' "$label"
}
{
  echo "# bench.py --steps 8 --warmup 2, default precision (mixed), nb 6"
  bench "batch 512 default" XB_NOP=1
  bench "batch 512 XB_GEMM4=0 (gemm8r everywhere)" XB_GEMM4=0
  bench "batch 512 XB_GEMM4=0 XB_DECODE_ASYNC=0" XB_GEMM4=0 XB_DECODE_ASYNC=0
  bench "batch 512 XB_DECODE_ASYNC=0" XB_DECODE_ASYNC=0
  bench "batch 512 XB_OVERLAP=0 (serial)" XB_OVERLAP=0
  bench "batch 512 XB_OVERLAP=0 XB_GEMM4=0" XB_OVERLAP=0 XB_GEMM4=0
  BARGS="--batch 1024"
  bench "batch 1024 default" XB_NOP=1
  bench "batch 1024 XB_DECODE_ASYNC=0" XB_DECODE_ASYNC=0
  BARGS="--batch 2048"
  bench "batch 2048 default" XB_NOP=1
  bench "batch 2048 XB_DECODE_ASYNC=0" XB_DECODE_ASYNC=0
  BARGS="--nbase 5"
  bench "nb 5 batch 512 default" XB_NOP=1
  bench "nb 5 batch 512 XB_DECODE_ASYNC=0" XB_DECODE_ASYNC=0
} > $out/gemm_kernel_and_decode_placement.txt 2>&1
