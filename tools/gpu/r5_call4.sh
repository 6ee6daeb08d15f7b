set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so
(timeout 900 python tests/prof_ab.py $L $L@PVIO_HIP_LM_WGS=192 $L@PVIO_HIP_LM_WGS=128 $L@PVIO_HIP_LM_WGS=96 $L@PVIO_HIP_LM_WGS=64 $L@PVIO_HIP_LM_WGS=32 2>&1 | grep -v "$F") > $OUT/r5d_ab_partials.txt
cat $OUT/r5d_ab_partials.txt
