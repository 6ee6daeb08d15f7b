set -u
# round 6, call 19: soak + random sweeps on the final library
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
(timeout 600 python tests/soak_gpu.py 2>&1 | grep -v amdgpu.ids | tail -3) > $OUT/r6_soak.txt
(timeout 900 python tests/sweep_random_klt.py 2>&1 | grep -v amdgpu.ids | tail -6) > $OUT/r6_sweep_random_klt.txt
(timeout 900 python tests/sweep_random_ransac.py 2>&1 | grep -v amdgpu.ids | tail -6) > $OUT/r6_sweep_random_ransac.txt
(timeout 900 python tests/sweep_random_detect.py 2>&1 | grep -v amdgpu.ids | tail -6) > $OUT/r6_sweep_random_detect.txt
for f in soak sweep_random_klt sweep_random_ransac sweep_random_detect; do tail -n 3 $OUT/r6_$f.txt; done
