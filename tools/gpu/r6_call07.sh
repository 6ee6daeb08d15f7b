#!/bin/bash
# round 6, call 8: k_linearize_lm as a launch of its own, two workgroups per CU for N <= 15 (LDS 80 KB, 256 registers)
cd /root/repo
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_golden.py -x -q 2>&1 | tail -8) > gpurun_out/r6_pytest_gpu_ba_tp3.txt
cat gpurun_out/r6_pytest_gpu_ba_tp3.txt
python tests/prof_large_tp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_prof_large_tp3.txt; cat gpurun_out/r6_prof_large_tp3.txt
for w in 10x50000_vio 30x50000_vio; do
  (timeout 600 python bench.py --workload $w --no-klt --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r6_bench_${w}_tp3.json) 2> gpurun_out/r6_bench_${w}_tp3.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r6_bench_${w}_tp3.json'))
print('$w', d['value'], d['roofline']['kernel_us'], d['roofline'].get('kernel_us_rocprof'), d['roofline']['frac'], d['roofline']['traffic'])
PY
done
