#!/bin/bash
# round 6, call 2: k_lk_track_levels with the step <-> wave mapping rotated by the block and eight waves per SIMD (64 VGPRs)
cd /root/repo
mkdir -p gpurun_out
(timeout -s KILL 300 python tests/micro/klt_forms_check.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r6_klt_forms_check_rot.txt
grep "mean of 50" gpurun_out/r6_klt_forms_check_rot.txt | grep "round 1"; tail -1 gpurun_out/r6_klt_forms_check_rot.txt
