#!/bin/bash
# round 6, call 12: host share of the keyframe solves over the rendered sequences (60 frames and 360 frames)
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python tests/prof_sequence_solves.py > gpurun_out/r6_prof_sequence_solves.txt 2>&1
cat gpurun_out/r6_prof_sequence_solves.txt
