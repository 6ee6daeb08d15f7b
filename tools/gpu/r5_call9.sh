set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(timeout 600 python -m pytest tests/test_host_ransac.py tests/test_gpu_klt.py tests/test_chain_parity.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8) > $OUT/r5g_pytest_ransac.txt
tail -6 $OUT/r5g_pytest_ransac.txt
(PVIO_SEQ_KEEP=$OUT/seq_long PVIO_LONG_SEQUENCE_WALL=1 PVIO_SEQ_REPORT_LONG=$OUT/r5_seq_long.json timeout 1500 python -m pytest tests/test_dropin_sequence.py -m gpu -q -x -s -k "long_sequence" 2>&1 | grep -v "$F" | grep -v "^Config::" | tail -30) > $OUT/r5g_pytest_seq.txt
tail -12 $OUT/r5g_pytest_seq.txt
