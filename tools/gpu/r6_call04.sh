#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
(for f in 0 1; do PVIO_HIP_LK_FORM=$f timeout -s KILL 300 python tests/micro/klt_makespan_probe.py 2>&1 | grep -v amdgpu.ids; done) > gpurun_out/r6_klt_makespan_probe.txt
cat gpurun_out/r6_klt_makespan_probe.txt
