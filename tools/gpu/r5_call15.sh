set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
: > $OUT/r5_ablation_raw.txt
for V in pvio_amd/lib/libpvio_hip.so tests/micro/variants/abl_nofailtest.so tests/micro/variants/abl_norows.so tests/micro/variants/abl_noblock.so tests/micro/variants/abl_noarith.so tests/micro/variants/abl_nomfma.so tests/micro/variants/abl_skeleton.so; do
  echo "== $V" >> $OUT/r5_ablation_raw.txt
  (PVIO_HIP_LIB=$R/$V timeout 200 python tests/prof_phases.py 2>&1 | grep -v "$F" | grep -A 22 "^vio per launch" | grep "tiles scaled\|factorization done\|back substitution done\|    end \|wave 1: tiles\|per launch") >> $OUT/r5_ablation_raw.txt
done
cat $OUT/r5_ablation_raw.txt
