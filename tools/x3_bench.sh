#!/bin/bash
# GPU tool: step time of the f16f8 encoder with chosen stages in the f16x3 arithmetic (XB_X3_STAGES masks as tools/x3_stages.py),
# BASELINE configs[2] (batch 512, nb 6), the driver's schedule.  usage: x3_bench.sh OUTFILE mask [mask ...]
out=$1; shift
for m in "$@"; do
  if [ "$m" = "f16x3" ]; then
    line=$(python bench.py --steps 8 --warmup 2 --cpu-chunks 0 --precision f16x3 2>/dev/null | tail -1)
  else
    line=$(XB_X3_STAGES=$m python bench.py --steps 8 --warmup 2 --cpu-chunks 0 --precision f16f8 2>/dev/null | tail -1)
  fi
  echo "$m $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step %.2f value %.3e stage_ms %s" % (d["ms_per_step"], d["value"], json.dumps(d.get("stage_ms_per_step", d.get("stages", "")))))')" >> "$out"
done
