set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(PVIO_LONG_SEQUENCE_WALL=1 PVIO_SEQ_REPORT_LONG=$OUT/r5_seq_long.json timeout 1500 python -m pytest tests/test_dropin_sequence.py tests/test_host_headless.py -m gpu -q -x -s --durations=8 -k "long_sequence or reference_pvio or dataset_layout" 2>&1 | grep -v "$F" | grep -v "^Config::" | tail -40) > $OUT/r5e_pytest_seq.txt
tail -25 $OUT/r5e_pytest_seq.txt
