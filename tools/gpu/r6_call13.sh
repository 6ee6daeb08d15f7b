#!/bin/bash
# round 6, call 13: k_linearize without the (1, 1) waves-per-EU pin (the small-window kernel is round 5's ISA again): BA gpu tests, default bench, 10 x 50 000
cd /root/repo
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_golden.py -x -q 2>&1 | tail -4) > gpurun_out/r6_pytest_gpu_ba_c13.txt; cat gpurun_out/r6_pytest_gpu_ba_c13.txt
(timeout 900 python bench.py --no-klt > gpurun_out/r6_bench_c.json) 2> gpurun_out/r6_bench_c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6_bench_c.json'))
print('headline', d['value'], d['speedup_vs_cpu_baseline'], d['roofline'].get('kernel_us_rocprof'), 'scaling window', d['scaling_window'].get('value'), d['scaling_window'].get('kernel_us_rank0'))
PY
