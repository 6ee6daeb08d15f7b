set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -3) > $OUT/r5_final_smoke.txt; cat $OUT/r5_final_smoke.txt
(PVIO_SEQ_REPORT_LONG=$OUT/r5_final_seq_long_relief.json timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | grep -v "$F" | tail -12) > $OUT/r5_final_pytest_gpu.txt; tail -10 $OUT/r5_final_pytest_gpu.txt
(timeout 400 python bench.py > $OUT/r5_final_bench_vio.json 2> $OUT/r5_final_bench_vio.err); python -c "
import json; d=json.load(open('$OUT/r5_final_bench_vio.json')); print(d['value'], d['speedup_vs_cpu_baseline'], d['roofline']['kernel_us_rocprof'])"
