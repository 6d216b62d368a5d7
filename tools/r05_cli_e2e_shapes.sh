#!/bin/bash
# GPU tool: the drop-in CLI end to end on the reference's REAL read shapes, through multi-read fast5 files
# (tests/h5write.py: classic HDF5 layout, VBZ), chunksize 10 000, batch 512, default precision, beside bench.py's per-chunk
# rate on the same box.  POC: ~3 000 samples per read (106-nt templates: ONE left-padded chunk per read), 4 000 reads per file;
# CPLX: ~25 000 samples per read (3 chunks).  -> profiles/r05_cli_e2e_shapes.txt
out=${1:-gpurun_out/r05/cli_e2e_shapes.txt}
mkdir -p "$(dirname "$out")"
{
  echo "# host: $(nproc) cores"
  echo "# bench.py (device-resident chunks, default precision), same box"
  python bench.py --steps 10 --warmup 2 --cpu-chunks 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("bench: %.2f ms per step of 512 chunks, %.3e samples/s = %.0f chunks/s" % (d["ms_per_step"], d["value"], d["value"] / 1e4))'
  echo "# CLI, POC shape: 40 000 reads x ~3 000 samples (2 000 - 4 000), multi-read fast5, 4 000 reads per file"
  python tools/cli_e2e.py --reads 40000 --samples 3000 --spread 0.333 --container fast5 --per-file 4000
  echo "# CLI, CPLX shape: 12 000 reads x ~25 000 samples (20 000 - 30 000), multi-read fast5, 4 000 reads per file"
  python tools/cli_e2e.py --reads 12000 --samples 25000 --spread 0.2 --container fast5 --per-file 4000
  echo "# CLI, POC shape through .xsig.npz bundles"
  python tools/cli_e2e.py --reads 40000 --samples 3000 --spread 0.333 --container npz --per-file 4000
} > "$out" 2>&1
