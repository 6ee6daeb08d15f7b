set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so; V=tests/micro/variants/relaxed_signals.so
(timeout 400 python tests/micro/order_probe.py $V 2>&1 | grep -v "$F" | tail -4) > $OUT/r5i_relaxed_probe.txt; cat $OUT/r5i_relaxed_probe.txt
(timeout 600 python tests/prof_ab.py $L $V 2>&1 | grep -v "$F") > $OUT/r5i_ab_relaxed.txt; cat $OUT/r5i_ab_relaxed.txt
