#!/bin/bash
# round 6, call 6: the large-window landmark role (ba_lin_tp.h) on the GPU: BA gpu tests, then the large windows
cd /root/repo
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_golden.py -x -q 2>&1 | tail -8) > gpurun_out/r6_pytest_gpu_ba_tp.txt
cat gpurun_out/r6_pytest_gpu_ba_tp.txt
for w in 10x50000_vio 30x50000_vio 30x50000_vision; do
  (timeout 600 python bench.py --workload $w --no-klt --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r6_bench_${w}_tp.json) 2> gpurun_out/r6_bench_${w}_tp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r6_bench_${w}_tp.json'))
print('$w', d['value'], d['roofline']['kernel_us'], d['roofline'].get('kernel_us_rocprof'), d['roofline']['frac'], d['roofline']['traffic'])
PY
done
