#!/bin/bash
# GPU tool: the drop-in CLI end to end under the default schedule (two co-scheduled pairs in flight through four staging slots)
# and with every call on its own (XB_FUSE=0), beside bench.py's per-chunk rate on the same box.  -> profiles/r04_cli_e2e.txt
out=${1:-gpurun_out/r04/cli_e2e.txt}
mkdir -p "$(dirname "$out")"
{
  echo "# bench.py (device-resident chunks, default precision), same box"
  for f in 1 0; do
    XB_FUSE=$f python bench.py --steps 10 --warmup 2 --cpu-chunks 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("XB_FUSE='$f' bench: %.2f ms per step of 512 chunks, %.3e samples/s" % (d["ms_per_step"], d["value"]))'
  done
  echo "# CLI: 6000 reads x ~50 000 samples (6 chunks of 10 000 per read: chunk efficiency 83 %)"
  python tools/cli_e2e.py --reads 6000 --samples 50000 --fuse 1,0
  echo "# CLI: 1500 reads x ~200 000 samples (chunk efficiency 95 %)"
  python tools/cli_e2e.py --reads 1500 --samples 200000 --fuse 1,0
} > "$out" 2>&1
