#!/bin/bash
# k_lk_track_units against k_lk_track on one box: correctness first, then launch times by track count and by waves of the queue
cd /root/repo
O=gpurun_out/r5_ab_klt_units.txt
(timeout 900 python -m pytest tests/test_gpu_klt.py -x -q 2>&1 | tail -3) > $O
(timeout 600 python tests/sweep_random_klt.py 2>&1 | tail -2) >> $O
for turn in 1 2; do
echo "== a wave per track (PVIO_HIP_LK_UNITS=0)"; PVIO_HIP_LK_UNITS=0 python tests/prof_klt.py 2>&1 | grep -v amdgpu.ids
echo "== default (units from a queue when tracks > SIMDs, two waves per SIMD)"; python tests/prof_klt.py 2>&1 | grep -v amdgpu.ids
done >> $O
for w in 1024 1536 2048 3072 4096; do
echo "== units always, $w waves"; PVIO_HIP_LK_UNITS=1 PVIO_HIP_LK_WAVES=$w python tests/prof_klt.py 2>&1 | grep -v amdgpu.ids
done >> $O
