#!/bin/bash
# the library as re-linked at the very end (objects rebuilt from source; device code byte-identical to the verified one): smoke() + the unit-queue test
cd /root/repo
O=gpurun_out/r5_final_relinked_smoke.txt
(timeout -s KILL 12 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3) > $O
(timeout -s KILL 12 python -m pytest tests/test_gpu_klt.py -x -q -k "unit_queue" 2>&1 | tail -2) >> $O
