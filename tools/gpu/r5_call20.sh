set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -3) > $OUT/r5l_smoke.txt; cat $OUT/r5l_smoke.txt
(timeout 1500 python -m pytest tests -m gpu -q -x --durations=4 2>&1 | grep -v "$F" | tail -8) > $OUT/r5l_pytest_gpu.txt; tail -7 $OUT/r5l_pytest_gpu.txt
(timeout 400 python bench.py > $OUT/r5l_bench_vio.json 2> $OUT/r5l_bench_vio.err); python -c "
import json; d=json.load(open('$OUT/r5l_bench_vio.json')); print(d['value'], d['speedup_vs_cpu_baseline'], d['roofline']['kernel_us_rocprof'], d['identical_candidate_reuse']['value'], d['scaling_window']['value'])"
(timeout 300 python bench.py --workload vision --no-klt --no-scaling-window > $OUT/r5l_bench_vision.json 2>/dev/null); python -c "
import json; d=json.load(open('$OUT/r5l_bench_vision.json')); print('vision', d['value'], d['speedup_vs_cpu_baseline'])"
