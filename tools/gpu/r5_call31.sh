set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so
(timeout 300 python tests/micro/order_probe.py tests/micro/variants/lin_sums_unroll2.so tests/micro/variants/lin_sums_loads_first.so 2>&1 | grep -v "$F" | tail -3)
(timeout 900 python tests/prof_ab.py $L tests/micro/variants/lin_sums_unroll2.so tests/micro/variants/lin_sums_loads_first.so 2>&1 | grep -v "$F") > $OUT/r5r_ab_sums.txt; cat $OUT/r5r_ab_sums.txt
