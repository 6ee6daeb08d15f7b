#!/bin/bash
# round 6, call 9: whole GPU suite + smoke + default bench + large-window benches + stamps on the tree with ba_lin_tp.h (two workgroups per CU for N <= 15)
cd /root/repo
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r6_pytest_gpu.txt
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3) > gpurun_out/r6_smoke.txt
cat gpurun_out/r6_pytest_gpu.txt gpurun_out/r6_smoke.txt
python tests/prof_large_tp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_prof_large_tp7.txt; cat gpurun_out/r6_prof_large_tp7.txt
(timeout 900 python bench.py > gpurun_out/r6_bench_b.json) 2> gpurun_out/r6_bench_b.err
for w in 10x50000_vio 30x50000_vio; do
  (timeout 600 python bench.py --workload $w --no-klt --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r6_bench_${w}_tp7.json) 2> gpurun_out/r6_bench_${w}_tp7.err
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6_bench_b.json'))
print('headline', d['value'], d['speedup_vs_cpu_baseline'], 'klt', d['klt']['value'], d['klt']['roofline']['kernel'], 'scaling window', d['scaling_window'].get('value'))
for w in ('10x50000_vio','30x50000_vio'):
    d=json.load(open('gpurun_out/r6_bench_%s_tp7.json'%w))
    print(w, d['value'], d['roofline']['kernel_us'], d['roofline'].get('kernel_us_rocprof'), d['roofline']['frac'], d['roofline']['traffic'])
PY
