set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(timeout 600 python tests/sweep_random_ransac.py 300 2>&1 | grep -v "$F" | tail -6) > $OUT/r5_sweep_random_ransac.txt; cat $OUT/r5_sweep_random_ransac.txt
(timeout 600 python tests/sweep_random_klt.py 2>&1 | grep -v "$F" | tail -4) > $OUT/r5_sweep_random_klt.txt; cat $OUT/r5_sweep_random_klt.txt
(timeout 600 python tests/sweep_random_detect.py 2>&1 | grep -v "$F" | tail -3) > $OUT/r5_sweep_random_detect.txt; cat $OUT/r5_sweep_random_detect.txt
(timeout 600 python tests/sweep_random_windows.py sharded 2>&1 | grep -v "$F" | tail -4) > $OUT/r5_sweep_random_windows_sharded.txt; cat $OUT/r5_sweep_random_windows_sharded.txt
(timeout 600 python tests/soak_gpu.py 2>&1 | grep -v "$F" | tail -3) > $OUT/r5_soak.txt; cat $OUT/r5_soak.txt
