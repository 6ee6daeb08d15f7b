set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/klt_dump; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
cd /tmp
PVIO_KLT_DUMP=$OUT/klt_dump/a PVIO_SEQ_IMAGE=oracle timeout 600 python $R/tests/chain_run.py $R/oracle/_ref/libpvio_ref.so /tmp/ra 66 8 3 25.0 full_relief_sweep 2>&1 | tail -1
PVIO_KLT_DUMP=$OUT/klt_dump/b PVIO_SEQ_IMAGE=hip timeout 600 python $R/tests/chain_run.py $R/oracle/_ref/libpvio_dropin.so /tmp/rb 66 8 3 25.0 full_relief_sweep 2>&1 | grep -v "$F" | tail -1
ls -la $OUT/klt_dump
cd $R; (timeout 600 python -m pytest tests/test_gpu_klt.py -q -x -s -k "border_corner" 2>&1 | grep -v "$F" | tail -5)
