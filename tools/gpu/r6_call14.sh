#!/bin/bash
# round 6, call 14: role_landmarks_tp -- L with 2 / 4 lanes per (landmark, column) for chunks of few landmarks, P without the unseen walk when every frame sees the landmark, the flush of many-frame
# windows straight to the row in HBM
cd /root/repo
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_golden.py -x -q 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl" | tail -3) > gpurun_out/r6_pytest_gpu_ba_c14.txt; cat gpurun_out/r6_pytest_gpu_ba_c14.txt
python tests/prof_large_tp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_prof_large_tp13.txt; cat gpurun_out/r6_prof_large_tp13.txt
for w in 10x50000_vio 30x50000_vio 30x50000_vision; do
  (timeout 600 python bench.py --workload $w --no-klt --no-cpu-baseline --no-pmc --steps 20 --warmup 5 > gpurun_out/r6_bench_${w}_tp13.json) 2> gpurun_out/r6_bench_${w}_tp13.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r6_bench_${w}_tp13.json'))
print('$w', d['value'], d['roofline']['kernel_us'])
PY
done
