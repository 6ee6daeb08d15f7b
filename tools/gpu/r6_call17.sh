set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 300 python tests/prof_large_tp.py 2>&1 | grep -v amdgpu.ids > $OUT/r6_prof_large_tp13.txt
cat $OUT/r6_prof_large_tp13.txt
