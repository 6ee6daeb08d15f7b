set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "LDS" | head -30 > $OUT/r5_lds_counters_available.txt
cat $OUT/r5_lds_counters_available.txt | cut -c1-200 | head -30
for LS in 8 10; do
  for C in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS; do
    PVIO_HIP_DENSE_ROW_STRIDE=$LS timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/r5_lds_${LS}_$C -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-klt --no-scaling-window > $OUT/r5_lds_${LS}_$C.log 2>&1
  done
done
python - <<'PY'
import csv, glob, os, collections
out=os.environ.get('OUT','/root/repo/gpurun_out')
for LS in (8,10):
    for C in ('SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_INSTS_LDS'):
        fs=glob.glob('%s/r5_lds_%d_%s/**/*counter_collection.csv'%(out,LS,C),recursive=True)
        if not fs: print(LS,C,'no file'); continue
        tot=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(fs[0])):
            k=r['Kernel_Name'].split('(')[0][-40:]
            if 'k_dense' in r['Kernel_Name']:
                tot[k][0]+=1; tot[k][1]+=float(r['Counter_Value'])
        for k,(n,v) in tot.items(): print('LS',LS,C,k,'launches',n,'per launch %.0f'%(v/max(n,1)))
PY
rm -rf $OUT/r5_lds_*/*/*kernel_trace.csv 2>/dev/null
