set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so; V=tests/micro/variants/before_lm_prefetch.so
(timeout 300 python tests/micro/order_probe.py $L 2>&1 | grep -v "$F" | tail -2)
(timeout 900 python tests/prof_ab.py $L $V 2>&1 | grep -v "$F") > $OUT/r5q_ab_lm_prefetch.txt; cat $OUT/r5q_ab_lm_prefetch.txt
(timeout 900 python tests/prof_ab.py $L $V 10 200 2>&1 | grep -v "$F") >> $OUT/r5q_ab_lm_prefetch.txt; tail -2 $OUT/r5q_ab_lm_prefetch.txt
(timeout 300 python bench.py --steps 100 --warmup 10 --no-klt --no-cpu-baseline --no-scaling-window 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_rocprof'])")
(timeout 300 python bench.py --workload vision --steps 100 --warmup 10 --no-klt --no-cpu-baseline --no-scaling-window 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vision', d['value'], d['roofline']['kernel_us_rocprof'])")
