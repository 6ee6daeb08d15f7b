#!/bin/bash
# Collects the round's evidence on the GPU box: run from the repo root through gpurun, e.g.
#   gpurun --timeout 3300 -- 'bash profiles/collect.sh r6'
# Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -4) > $OUT/${TAG}_smoke.txt
(cd $R && PVIO_CHAIN_REPORT=$OUT/${TAG}_chain_parity.json PVIO_SEQ_REPORT=$OUT/${TAG}_seq_backend.json PVIO_SEQ_REPORT_FULL=$OUT/${TAG}_seq_full_product.json PVIO_SEQ_REPORT_LONG=$OUT/${TAG}_seq_long_relief.json PVIO_SEQ_REPORT_LONG_B=$OUT/${TAG}_seq_long_relief_b.json python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -16) > $OUT/${TAG}_pytest_gpu.txt
# large windows (k_linearize in its matrix-core form): bench lines + kernel stats + HBM counters of the 30 KF x 50k VIO window
for W in 30x50000_vio 30x50000_vision 10x50000_vio; do
  python $R/bench.py --workload $W --steps 10 --warmup 2 --no-klt --no-cpu-baseline > $OUT/${TAG}_bench_$W.json 2> $OUT/${TAG}_bench_$W.err
done
LARGE="python $R/bench.py --workload 30x50000_vio --steps 4 --warmup 1 --no-klt --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_large_stats -- $LARGE > $OUT/${TAG}_prof_large_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_prof_large_fetch -- $LARGE > $OUT/${TAG}_prof_large_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_prof_large_write -- $LARGE > $OUT/${TAG}_prof_large_write.log 2>&1
python $R/profiles/summarize_pmc.py $OUT/${TAG}_prof_large_fetch $OUT/${TAG}_prof_large_write $OUT/${TAG}_pmc_hbm_30x50000_vio.json > /dev/null
find $OUT/${TAG}_prof_large_stats -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats_bench_30x50000_vio.csv \;
# the same three passes for the 10 KF x 50k VIO window (the one north_star states the scaling target on; round 6: k_linearize in its two-workgroups-per-CU form)
MID="python $R/bench.py --workload 10x50000_vio --steps 8 --warmup 2 --no-klt --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_mid_stats -- $MID > $OUT/${TAG}_prof_mid_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_prof_mid_fetch -- $MID > $OUT/${TAG}_prof_mid_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_prof_mid_write -- $MID > $OUT/${TAG}_prof_mid_write.log 2>&1
python $R/profiles/summarize_pmc.py $OUT/${TAG}_prof_mid_fetch $OUT/${TAG}_prof_mid_write $OUT/${TAG}_pmc_hbm_10x50000_vio.json > /dev/null
find $OUT/${TAG}_prof_mid_stats -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats_bench_10x50000_vio.csv \;
rm -rf $OUT/${TAG}_prof_mid_fetch $OUT/${TAG}_prof_mid_write $OUT/${TAG}_prof_mid_stats 2>/dev/null
rm -rf $OUT/${TAG}_prof_large_fetch/*/*kernel_trace.csv $OUT/${TAG}_prof_large_write/*/*kernel_trace.csv $OUT/${TAG}_prof_large_stats/*/*kernel_trace.csv 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_stats -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-scaling-window > $OUT/${TAG}_prof_stats.log 2>&1
# counters in their own passes, kernel trace only (no sys / runtime traces)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_prof_fetch -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-scaling-window > $OUT/${TAG}_prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_prof_write -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-scaling-window > $OUT/${TAG}_prof_write.log 2>&1
python $R/profiles/summarize_pmc.py $OUT/${TAG}_prof_fetch $OUT/${TAG}_prof_write $OUT/${TAG}_pmc_hbm.json > /dev/null
find $OUT/${TAG}_prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats_bench_vio.csv \;
# the headline lines (round 3: bench.py collects roofline.traffic itself, with two counter passes of its own run; the summary above is the
# independent collection it can be compared with)
python $R/bench.py > $OUT/${TAG}_bench_vio.json 2> $OUT/${TAG}_bench_vio.err
python $R/bench.py --workload vision > $OUT/${TAG}_bench_vision.json 2> $OUT/${TAG}_bench_vision.err
# the multi-GPU code path with the one rank there is: process group, RCCL communicator, sharded iteration (captured in the hipGraph)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 $R/bench.py --gpus 1 --force-sharded --steps 100 --warmup 10 --no-klt --no-cpu-baseline > $OUT/${TAG}_bench_sharded_1rank.json 2> $OUT/${TAG}_bench_sharded_1rank.err
# end to end: rendered sequence -> front end -> PnP -> sliding-window BA (tests/test_host_headless.py prints the trajectory error)
(cd $R && python -m pytest tests/test_host_headless.py -m gpu -q -s 2>&1 | grep "^headless") > $OUT/${TAG}_headless.txt
# complete keyframe solves (upload + solve + download), LK launch time against the batch size, dense kernel of large windows
(cd $R && python tests/prof_sequence_solves.py full full_relief) > $OUT/${TAG}_prof_sequence_solves_60.txt 2>&1  # (the 360-frame sequences: tools/gpu/r6_call12.sh, ~12 min)
(cd $R && python tests/sweep_random_dropin.py gpu 24 2>&1 | tail -26) > $OUT/${TAG}_sweep_random_dropin_gpu.txt
(cd $R && python tests/prof_upload.py) > $OUT/${TAG}_prof_upload.txt 2>&1
(cd $R && python tests/prof_klt.py) > $OUT/${TAG}_prof_klt.txt 2>&1
(cd $R && python tests/prof_dense_large.py) > $OUT/${TAG}_prof_dense_large.txt 2>&1
(cd $R && python tests/prof_phases.py) > $OUT/${TAG}_prof_phases.txt 2>&1
(cd $R && python tests/prof_large_tp.py 2>&1 | grep -v amdgpu.ids) > $OUT/${TAG}_prof_large_tp.txt
(cd $R && timeout -s KILL 400 python tests/micro/klt_bench_ab.py 2>&1 | grep -v amdgpu.ids) > $OUT/${TAG}_klt_bench_ab_final.txt
# the raw traces are large: keep the summaries only
rm -rf $OUT/${TAG}_prof_fetch/*/*kernel_trace.csv $OUT/${TAG}_prof_write/*/*kernel_trace.csv 2>/dev/null
ls -la $OUT | head -40
