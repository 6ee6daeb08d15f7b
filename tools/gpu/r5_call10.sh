set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
L=pvio_amd/lib/libpvio_hip.so
(timeout 900 python tests/prof_ab.py $L $L@PVIO_HIP_LM_WGS=300 $L@PVIO_HIP_LM_WGS=345 $L@PVIO_HIP_LM_WGS=460 $L@PVIO_HIP_LM_WGS=690 2>&1 | grep -v "$F") > $OUT/r5h_ab_more_wgs.txt
cat $OUT/r5h_ab_more_wgs.txt
