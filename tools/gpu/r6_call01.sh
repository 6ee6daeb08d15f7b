#!/bin/bash
# round 6, call 1: the level-per-wave LK form (k_lk_track_levels) against the wave-per-track and unit-queue forms on one box; KLT gpu tests; first bench line
cd /root/repo
mkdir -p gpurun_out
(timeout -s KILL 300 python tests/micro/klt_forms_check.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r6_klt_forms_check.txt
(timeout 900 python -m pytest tests/test_gpu_klt.py -x -q 2>&1 | tail -5) > gpurun_out/r6_pytest_gpu_klt.txt
(timeout 900 python bench.py > gpurun_out/r6_bench_a.json) 2> gpurun_out/r6_bench_a.err
tail -3 gpurun_out/r6_klt_forms_check.txt; cat gpurun_out/r6_pytest_gpu_klt.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6_bench_a.json'))
print(d['value'], d['speedup_vs_cpu_baseline'], d['klt']['value'], d['klt']['roofline']['kernel'], d['scaling_window'].get('value'))
PY
