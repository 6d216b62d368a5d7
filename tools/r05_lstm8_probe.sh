#!/bin/bash
# GPU tool: the M-split feasibility probe (tools/ubench/lstm8_probe.hip), variants built in the build container
out=${1:-gpurun_out/r05/lstm8_probe.txt}
mkdir -p "$(dirname "$out")"
{
  for v in s0 s1 s1_nodma s1_noread s1_nogate s1_nofp8 s1_gindirect s1_mfmaonly; do
    echo "== $v"
    timeout -k 10 120 tools/ubench/lstm8_probe_$v 192 300 || echo "FAILED $v"
  done
} > "$out" 2>&1
