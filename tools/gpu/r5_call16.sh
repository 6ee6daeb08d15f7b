set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep "Counter_Name" | grep "SQ_" | awk '{print $3}' | tr '\n' ' ' > $OUT/r5_sq_counters_available.txt
cat $OUT/r5_sq_counters_available.txt | cut -c1-3000
for C in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_IFETCH SQ_INSTS_SENDMSG SQ_VALU_MFMA_BUSY_CYCLES; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/r5_sq_$C -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-klt --no-scaling-window > $OUT/r5_sq_$C.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out=os.environ['OUT']
res={}
for d in sorted(glob.glob(out+'/r5_sq_*')):
    if not os.path.isdir(d): continue
    C=os.path.basename(d)[6:]
    fs=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not fs: print(C,'no data'); continue
    by=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        n=r['Kernel_Name']
        for k in ('k_dense','k_linearize','k_reduce','k_backsub'):
            if k in n: by[k].append(float(r['Counter_Value']))
    line=C
    for k in ('k_dense','k_linearize','k_reduce','k_backsub'):
        v=by[k]
        if not v: continue
        if k=='k_dense':
            v=sorted(v); big=v[int(len(v)*0.45):]   # the factoring launches are the upper ~64 %: take the upper 55 % to be safe
            line+='  k_dense all %.0f / factoring %.0f'%(sum(v)/len(v), sum(big)/len(big))
        else: line+='  %s %.0f'%(k,sum(v)/len(v))
    print(line)
PY
rm -rf $OUT/r5_sq_*/*/*kernel_trace.csv 2>/dev/null
