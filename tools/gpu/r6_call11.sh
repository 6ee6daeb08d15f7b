#!/bin/bash
# round 6, call 11: where the host time of a keyframe solve of the rendered sequence goes (PVIO_HIP_TIMING=1)
cd /root/repo
mkdir -p gpurun_out
PVIO_SEQ_IMAGE=hip PVIO_HIP_TIMING=1 PVIO_HIP_REUSE_CANDIDATES=0 timeout 600 python tests/chain_run.py oracle/_ref/libpvio_dropin.so /tmp/prof_seq 60 6 3 25.0 full_relief 2> gpurun_out/r6_seq_timing_raw.txt > /dev/null
grep "pvio-hip" gpurun_out/r6_seq_timing_raw.txt | grep -v "klt\|detect" | tail -40
