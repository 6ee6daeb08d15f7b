#!/bin/bash
# round 6, call 5: the whole GPU suite + smoke on the tree with k_lk_track_levels as the default LK form and the ADVICE r5 fixes
cd /root/repo
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r6_pytest_gpu.txt
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5) > gpurun_out/r6_smoke.txt
cat gpurun_out/r6_pytest_gpu.txt gpurun_out/r6_smoke.txt
