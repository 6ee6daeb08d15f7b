set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(timeout 900 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | grep -v "$F" | tail -30) > $OUT/r5a_pytest_gpu.txt
(timeout 300 python bench.py > $OUT/r5a_bench_vio.json 2> $OUT/r5a_bench_vio.err)
L=pvio_amd/lib/libpvio_hip.so
(timeout 600 python tests/prof_ab.py $L $L@PVIO_HIP_LM_WGS=128 $L@PVIO_HIP_LM_WGS=64 $L@PVIO_HIP_LM_WGS=48 $L@PVIO_HIP_LM_WGS=32 2>&1 | grep -v "$F") > $OUT/r5a_ab_partials.txt
tail -3 $OUT/r5a_pytest_gpu.txt; cat $OUT/r5a_bench_vio.json; cat $OUT/r5a_ab_partials.txt
