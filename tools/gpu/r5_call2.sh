set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so
(timeout 300 python tests/micro/order_probe.py $L 2>&1 | grep -v "$F" | tail -14) > $OUT/r5b_order_probe.txt
(timeout 600 python tests/prof_ab.py $L $L@PVIO_HIP_DENSE_ROW_STRIDE=8 2>&1 | grep -v "$F") > $OUT/r5b_ab_row_stride.txt
(timeout 600 python tests/prof_ab.py $L $L@PVIO_HIP_DENSE_ROW_STRIDE=8 10 200 2>&1 | grep -v "$F") >> $OUT/r5b_ab_row_stride.txt
(timeout 300 python tests/prof_phases.py 2>&1 | grep -v "$F") > $OUT/r5b_prof_phases.txt
cat $OUT/r5b_order_probe.txt $OUT/r5b_ab_row_stride.txt; head -30 $OUT/r5b_prof_phases.txt
