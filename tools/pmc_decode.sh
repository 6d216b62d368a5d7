#!/bin/bash
# PMC passes over the CRF decode alone (tools/decode_sweeps.py as the workload; NB / N / T from the environment).
# usage: tools/pmc_decode.sh OUTDIR   (run on the GPU box from the repo root)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcdec}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/decode_sweeps.py > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/fail.log
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "crf_decode_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print("%-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
