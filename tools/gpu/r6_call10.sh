set -u
# round 6, call 10: SQ counters of k_linearize on the 10 x 50 000 window (name the bound), one counter per pass, counters only
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp OUT
for C in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/r6_sq_$C -- python $R/bench.py --workload 10x50000_vio --steps 4 --warmup 1 --no-cpu-baseline --no-klt --no-pmc > $OUT/r6_sq_$C.log 2>&1
done
python - <<'PY' > $OUT/r6_sq_counters_10x50000.txt
import csv, glob, os, collections
out=os.environ['OUT']
print("Round 6: SQ counters per launch, 10 KF x 50 000 landmarks VIO (rocprofv3 --kernel-trace --pmc <one counter per pass>, bench.py --workload 10x50000_vio --steps 4); averages per launch")
for d in sorted(glob.glob(out+'/r6_sq_*')):
    if not os.path.isdir(d): continue
    C=os.path.basename(d)[6:]
    fs=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not fs: print(C,'no data'); continue
    by=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        n=r['Kernel_Name']
        for k in ('k_dense','k_linearize','k_reduce','k_backsub'):
            if k in n: by[k].append(float(r['Counter_Value']))
    print(C, '  '.join('%s %.0f (%d launches)'%(k,sum(v)/len(v),len(v)) for k,v in by.items()))
PY
cat $OUT/r6_sq_counters_10x50000.txt
rm -rf $OUT/r6_sq_SQ_* 2>/dev/null
