#!/bin/bash
# the round's last GPU seconds: the new unit-queue parity test and smoke() on the final library, every step under a hard limit
cd /root/repo
O=gpurun_out/r5_final_klt_units_and_smoke.txt
(timeout -s KILL 35 python -m pytest tests/test_gpu_klt.py -x -q -k "unit_queue or large_batch" 2>&1 | tail -3) > $O
(timeout -s KILL 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -4) >> $O
