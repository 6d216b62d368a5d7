# LDS / MFMA counters of the two-groups-per-workgroup recurrence: lstm_quad_kernel (XB_LSTM_QUAD=1) vs lstm_kernel<.., DUAL>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04q/pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for q in 1 0; do
  export XB_LSTM_QUAD=$q
  i=0
  for set in "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $O/q${q}_p$i -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-chunks 0 > $O/q${q}_p$i.log 2>&1 || echo "q$q pass $i failed" >> $O/fail.log
  done
done
python3 - <<PY
import csv, glob, collections, re
for q in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/q%d_p*/**/*counter_collection.csv" % q, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^(void )?\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60]
            if "lstm" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print("QUAD=%d" % q, k)
        for c, v in sorted(d.items()):
            print("   %-34s n=%-4d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
