#!/bin/bash
# round 6, final: the GPU suite and smoke() on the final tree (the evidence the ISA ledger names)
cd /root/repo
mkdir -p gpurun_out
(PVIO_SEQ_REPORT_LONG=gpurun_out/r6_seq_long_relief.json PVIO_SEQ_REPORT_LONG_B=gpurun_out/r6_seq_long_relief_b.json timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -14) > gpurun_out/r6_pytest_gpu.txt
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path\|amdgpu.ids" | tail -3) > gpurun_out/r6_smoke.txt
cat gpurun_out/r6_pytest_gpu.txt gpurun_out/r6_smoke.txt
