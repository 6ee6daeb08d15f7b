set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so; V=tests/micro/variants/la_barrier.so
(timeout 300 python tests/micro/order_probe.py $V 2>&1 | grep -v "$F" | tail -3) > $OUT/r5k_barrier_probe.txt; cat $OUT/r5k_barrier_probe.txt
(timeout 900 python tests/prof_ab.py $L $V 2>&1 | grep -v "$F") > $OUT/r5k_ab_barrier.txt; cat $OUT/r5k_ab_barrier.txt
(timeout 900 python tests/prof_ab.py $L $V 10 200 2>&1 | grep -v "$F") >> $OUT/r5k_ab_barrier.txt; tail -2 $OUT/r5k_ab_barrier.txt
(PVIO_HIP_LIB=$R/$V timeout 200 python tests/prof_phases.py 2>&1 | grep -v "$F" | grep -A 22 "^vio per launch" | grep "factorization done\|wave 1: tiles\|back substitution\|per launch")
