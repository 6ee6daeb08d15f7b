#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
(timeout -s KILL 300 python tests/micro/klt_levels_ab.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r6_klt_levels_ab.txt
cat gpurun_out/r6_klt_levels_ab.txt | grep "round 1"
