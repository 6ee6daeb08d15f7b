set -u
# round 6, call 16: LDS row padding of the large-window role, same-box A/B + phase stamps of both builds
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python tests/micro/tp_pad_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/r6_tp_pad_ab.txt
(echo "== unpadded"; PVIO_HIP_LIB=$R/tests/micro/variants/tp_nopad.so timeout 300 python tests/prof_large_tp.py 2>&1 | grep -v amdgpu.ids; echo "== padded"; timeout 300 python tests/prof_large_tp.py 2>&1 | grep -v amdgpu.ids) > $OUT/r6_tp_pad_stamps.txt
cat $OUT/r6_tp_pad_ab.txt; cat $OUT/r6_tp_pad_stamps.txt
