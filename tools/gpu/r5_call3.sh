set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(timeout 900 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | grep -v "$F" | tail -12) > $OUT/r5c_pytest_gpu.txt
(timeout 400 python bench.py > $OUT/r5c_bench_vio.json 2> $OUT/r5c_bench_vio.err)
L=pvio_amd/lib/libpvio_hip.so
(timeout 600 python tests/prof_ab.py $L $L@PVIO_HIP_LM_WGS=-1 2>&1 | grep -v "$F") > $OUT/r5c_ab_partials_default.txt
tail -3 $OUT/r5c_pytest_gpu.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c_bench_vio.json'))
print(d['value'], d['speedup_vs_cpu_baseline'], d['roofline']['traffic'], d['roofline']['kernel_us_rocprof'], d['cpu_baseline']['value'])
PY
cat $OUT/r5c_ab_partials_default.txt
