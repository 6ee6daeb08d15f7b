set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
: > $OUT/r5_ablation3_raw.txt
for V in pvio_amd/lib/libpvio_hip.so tests/micro/variants/abl_skeleton.so tests/micro/variants/abl_skel_norsq.so tests/micro/variants/abl_skel_norowio.so tests/micro/variants/abl_skel_bare.so; do
  echo "== $V" >> $OUT/r5_ablation3_raw.txt
  (PVIO_HIP_LIB=$R/$V timeout 200 python tests/prof_phases.py 2>&1 | grep -v "$F" | grep -A 22 "^vio per launch" | grep "factorization done\|wave 1: tiles\|back substitution\|    end ") >> $OUT/r5_ablation3_raw.txt
done
cat $OUT/r5_ablation3_raw.txt
