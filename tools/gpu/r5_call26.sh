set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
: > $OUT/r5_ablation_lin2_raw.txt
for V in pvio_amd/lib/libpvio_hip.so tests/micro/variants/abl_lin_notiles.so tests/micro/variants/abl_lin_nosums.so; do
  rm -rf /tmp/abl_prof; PVIO_HIP_LIB=$R/$V timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_prof -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-klt --no-scaling-window > /tmp/abl.log 2>&1
  f=$(find /tmp/abl_prof -name '*kernel_stats.csv' | head -1)
  echo "== $V" >> $OUT/r5_ablation_lin2_raw.txt
  python - "$f" >> $OUT/r5_ablation_lin2_raw.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'k_linearize' in n: print('k_linearize calls',r['Calls'],'avg us %.2f'%(float(r['AverageNs'])/1000))
PY
done
cat $OUT/r5_ablation_lin2_raw.txt
cd $R; F='^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path'
(timeout 300 python tests/micro/order_probe.py tests/micro/variants/lin_tiles_unroll2.so tests/micro/variants/lin_tiles_unroll4.so 2>&1 | grep -v "$F" | tail -3)
(timeout 900 python tests/prof_ab.py pvio_amd/lib/libpvio_hip.so tests/micro/variants/lin_tiles_unroll2.so tests/micro/variants/lin_tiles_unroll4.so 2>&1 | grep -v "$F") > $OUT/r5p_ab_tiles_unroll.txt; cat $OUT/r5p_ab_tiles_unroll.txt
