#!/bin/bash
# Collects the round's evidence on the GPU box: run from the repo root through gpurun, e.g.
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r1'
# Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench_vio.json 2> $OUT/${TAG}_bench_vio.err
python $R/bench.py --workload vision > $OUT/${TAG}_bench_vision.json 2> $OUT/${TAG}_bench_vision.err
(cd $R && python -m pytest tests -m gpu -q 2>&1 | tail -5) > $OUT/${TAG}_pytest_gpu.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_stats -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_prof_stats.log 2>&1
# counters in their own passes, kernel trace only (no sys / runtime traces)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_prof_fetch -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_prof_write -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_prof_write.log 2>&1
python $R/profiles/summarize_pmc.py $OUT/${TAG}_prof_fetch $OUT/${TAG}_prof_write $OUT/${TAG}_pmc_hbm.json > /dev/null
find $OUT/${TAG}_prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats_bench_vio.csv \;
# the raw traces are large: keep the summaries only
rm -rf $OUT/${TAG}_prof_fetch/*/*kernel_trace.csv $OUT/${TAG}_prof_write/*/*kernel_trace.csv 2>/dev/null
ls -la $OUT | head -40
