#!/bin/bash
# GPU tool: bit-identity of the working build against the round-4 library (tools/lib_ab.py) + the bench under both, same box
tag=${1:-lean}
mkdir -p gpurun_out/r05
python tools/lib_ab.py xna_basecaller_amd/libxnacall_r4.so xna_basecaller_amd/libxnacall.so > gpurun_out/r05/lib_ab_$tag.txt 2>&1
tail -16 gpurun_out/r05/lib_ab_$tag.txt
python bench.py --steps 12 --warmup 3 --cpu-chunks 0 > gpurun_out/r05/${tag}_bench.json 2> gpurun_out/r05/${tag}_bench.err
XNA_LIBXNACALL=$PWD/xna_basecaller_amd/libxnacall_r4.so python bench.py --steps 12 --warmup 3 --cpu-chunks 0 > gpurun_out/r05/${tag}_bench_r4lib.json 2>> gpurun_out/r05/${tag}_bench.err
XB_OVERLAP=0 python bench.py --steps 6 --warmup 2 --cpu-chunks 0 > gpurun_out/r05/${tag}_bench_serial.json 2>> gpurun_out/r05/${tag}_bench.err
python - <<PY
import json
for f in ("${tag}_bench","${tag}_bench_r4lib","${tag}_bench_serial"):
    try:
        d=json.load(open("gpurun_out/r05/%s.json"%f)); print(f, round(d["ms_per_step"],2), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"],3), round(d["roofline"]["frac"],4), {k: round(v,1) for k,v in d["stage_ms_per_step"].items()}, "decode", round(d["roofline_decode"]["frac"],3))
    except Exception as e:
        print(f, "FAILED", e)
PY
