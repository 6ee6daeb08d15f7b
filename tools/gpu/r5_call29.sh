set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
: > $OUT/r5_ablation_lin5_raw.txt
for V in tests/micro/variants/abl_lin_onlyimu.so tests/micro/variants/abl_lin_onlyprior.so tests/micro/variants/abl_lin_onlyprologue.so; do
  rm -rf /tmp/abl_prof; PVIO_HIP_LIB=$R/$V timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/abl_prof -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-klt --no-scaling-window > /tmp/abl.log 2>&1
  f=$(find /tmp/abl_prof -name '*kernel_trace.csv' | head -1)
  echo "== $V" >> $OUT/r5_ablation_lin5_raw.txt
  python - "$f" >> $OUT/r5_ablation_lin5_raw.txt <<'PY'
import csv,sys,numpy as np
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in csv.DictReader(open(sys.argv[1])) if 'k_linearize' in r['Kernel_Name']]
d=np.array(d); print('k_linearize launches',len(d),'quantiles us 10/25/50/75/90: %s'%np.round(np.percentile(d,[10,25,50,75,90]),2),' count<8us',int((d<8).sum()))
PY
done
cat $OUT/r5_ablation_lin5_raw.txt
