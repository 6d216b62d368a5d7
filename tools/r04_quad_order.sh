# diagnostic: cycle stamps of lstm_quad_kernel with the FP8 products reordered / removed (WRONG results; XB_Q_ORDER builds)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04q; mkdir -p $O; cd $R
for v in ${VARIANTS:-diag dq1 dq2 dq3}; do
  (export XNA_LIBXNACALL=$R/xna_basecaller_amd/libxnacall_$v.so PREC=2 XB_OVERLAP=0; N=1024 timeout -k 10 150 python tools/lstm_stamps.py > $O/stamps_$v.txt 2>&1)
  echo "== $v"; tail -11 $O/stamps_$v.txt
done
