set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so; V=tests/micro/variants/fin_copy.so
(timeout 300 python tests/micro/order_probe.py $V 2>&1 | grep -v "$F" | tail -14) > $OUT/r5m_fincopy_probe.txt; cat $OUT/r5m_fincopy_probe.txt
(timeout 900 python tests/prof_ab.py $L $V 2>&1 | grep -v "$F") > $OUT/r5m_ab_fincopy.txt; cat $OUT/r5m_ab_fincopy.txt
