#!/bin/bash
# PMC passes over two bench steps (one co-scheduled pair of calls in the default schedule) (each pass = its own rocprofv3 run; counters only, no traces).
# PASSES=2 limits the run to the first two sets (FETCH_SIZE, WRITE_SIZE: they do not fit one pass together).
# usage: tools/pmc_gemm.sh OUTDIR   (run on the GPU box from the repo root)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
rm -rf $OUT; mkdir -p $OUT      # a fresh directory: the summaries take every csv they find
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  if [ -n "$PASSES" ] && [ $i -gt $PASSES ]; then break; fi
  timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --cpu-chunks 0 $BENCH_ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/fail.log
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^(void )?\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in agg.items():
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write("   %-34s n=%-4d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open("$OUT/summary.txt").read()[:6000])
PY
