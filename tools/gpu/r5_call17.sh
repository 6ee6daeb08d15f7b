set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so; V=tests/micro/variants/relaxed_signals.so; W=tests/micro/variants/relaxed_nosleep.so
(timeout 300 python tests/micro/order_probe.py $W 2>&1 | grep -v "$F" | tail -2)
(timeout 900 python tests/prof_ab.py $L $V@PVIO_HIP_DENSE_ROW_STRIDE=10 $W $W@PVIO_HIP_DENSE_ROW_STRIDE=10 2>&1 | grep -v "$F") > $OUT/r5j_ab_combo.txt; cat $OUT/r5j_ab_combo.txt
