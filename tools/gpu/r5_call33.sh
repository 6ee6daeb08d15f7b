#!/bin/bash
# A/B of the LK termination tests without FP64 in the loop (same box): old = HEAD~ klt.hip, new = product
cd /root/repo
for turn in 1 2; do
echo "== old (FP64 comparisons every iteration)"; PVIO_HIP_LIB=tests/micro/variants/libpvio_hip_klt_old.so python tests/prof_klt.py 2>&1 | grep -v amdgpu.ids
echo "== new (float wherever it decides the same)"; python tests/prof_klt.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r5_ab_klt_float_tests.txt
timeout 900 python -m pytest tests/test_gpu_klt.py -x -q 2>&1 | tail -3 >> gpurun_out/r5_ab_klt_float_tests.txt
timeout 600 python tests/sweep_random_klt.py 2>&1 | tail -2 >> gpurun_out/r5_ab_klt_float_tests.txt
