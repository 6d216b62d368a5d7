cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/beamprof6 -- python3 $R/tools/beam_time.py 512 6 3 > $R/gpurun_out/beamprof6.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/beamprof4 -- python3 $R/tools/beam_time.py 512 4 5 > $R/gpurun_out/beamprof4.log 2>&1
grep "scans" $R/gpurun_out/beamprof6.log $R/gpurun_out/beamprof4.log
head -4 $R/gpurun_out/beamprof6/*/*kernel_stats.csv | cut -c1-150
head -4 $R/gpurun_out/beamprof4/*/*kernel_stats.csv | cut -c1-150
