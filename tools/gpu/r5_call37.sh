#!/bin/bash
# last GPU seconds of the round: k_lk_track_units (fixed loop head) against k_lk_track, every step under a hard limit
cd /root/repo
O=gpurun_out/r5_klt_units_check.txt
timeout -s KILL 25 python tests/micro/klt_units_check.py 2>&1 | grep -v amdgpu.ids > $O; echo "rc=${PIPESTATUS[0]}" >> $O
if grep -q "CHECK ok" $O; then
  (timeout -s KILL 45 python -m pytest tests/test_gpu_klt.py -x -q 2>&1 | tail -3) >> $O
fi
