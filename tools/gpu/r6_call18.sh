set -u
# round 6, call 18: LDS bank-conflict counters of k_linearize on the 10 x 50 000 window, the round's first LDS layout (tests/micro/variants/tp_nopad.so) against the shipped one
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp OUT
for V in nopad product; do
  if [ $V = nopad ]; then export PVIO_HIP_LIB=$R/tests/micro/variants/tp_nopad.so; else unset PVIO_HIP_LIB; fi
  for C in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/r6_sqb_${V}_$C -- python $R/bench.py --workload 10x50000_vio --steps 4 --warmup 1 --no-cpu-baseline --no-klt --no-pmc > $OUT/r6_sqb_${V}_$C.log 2>&1
  done
done
python - <<'PY' > $OUT/r6_sq_lds_layout_10x50000.txt
import csv, glob, os, collections
out=os.environ['OUT']
print("Round 6: LDS counters of k_linearize per launch, 10 KF x 50 000 landmarks VIO (rocprofv3 --kernel-trace --pmc <one counter per pass>, bench.py --workload 10x50000_vio --steps 4)")
print("nopad = X rows of 14 pairs, U rows of 64 doubles, LMR rows of 8 (-DPVBA_TP_PAD=0); product = X as two arrays of 7 pairs per row, U rows of 65, LMR rows of 9 (same bytes of X)")
for d in sorted(glob.glob(out+'/r6_sqb_*')):
    if not os.path.isdir(d): continue
    C=os.path.basename(d)[7:]
    fs=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not fs: print(C,'no data'); continue
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if 'k_linearize' in r['Kernel_Name']]
    print('%-32s k_linearize %.0f (%d launches)'%(C,sum(v)/max(len(v),1),len(v)))
PY
cat $OUT/r6_sq_lds_layout_10x50000.txt
rm -rf $OUT/r6_sqb_* 2>/dev/null
