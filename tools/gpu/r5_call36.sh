#!/bin/bash
# k_lk_track_units (LDS queue, a block of eight waves per CU) against k_lk_track on one box
cd /root/repo
O=gpurun_out/r5_ab_klt_units.txt
(timeout 900 python -m pytest tests/test_gpu_klt.py -x -q 2>&1 | tail -3) > $O
(timeout 600 python tests/sweep_random_klt.py 2>&1 | tail -2) >> $O
for turn in 1 2; do
echo "== a wave per track (PVIO_HIP_LK_UNITS=0)"; PVIO_HIP_LK_UNITS=0 python tests/prof_klt.py 2>&1 | grep -v amdgpu.ids
echo "== default (units when tracks > SIMDs, one block of eight waves per CU)"; python tests/prof_klt.py 2>&1 | grep -v amdgpu.ids
done >> $O
for w in 128 192 256 384 512; do
echo "== units always, $w blocks"; PVIO_HIP_LK_UNITS=1 PVIO_HIP_LK_BLOCKS=$w python tests/prof_klt.py 2>&1 | grep -v amdgpu.ids
done >> $O
