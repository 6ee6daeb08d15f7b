set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
: > $OUT/r5_ablation_lin_raw.txt
for V in pvio_amd/lib/libpvio_hip.so tests/micro/variants/abl_lin_noimu.so tests/micro/variants/abl_lin_nolm.so tests/micro/variants/abl_lin_noprior.so tests/micro/variants/abl_lin_onlyprologue.so; do
  rm -rf /tmp/abl_prof; PVIO_HIP_LIB=$R/$V timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_prof -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-klt --no-scaling-window > /tmp/abl.log 2>&1
  f=$(find /tmp/abl_prof -name '*kernel_stats.csv' | head -1)
  echo "== $V" >> $OUT/r5_ablation_lin_raw.txt
  grep "k_linearize\|k_reduce\|k_dense\|k_backsub" $f | awk -F, '{print $1, "calls", $2, "avg_ns", $4}' >> $OUT/r5_ablation_lin_raw.txt
done
cat $OUT/r5_ablation_lin_raw.txt
