#!/bin/bash
# Collects the profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02
# -> gpurun_out/<tag>_*: bench JSONs (default, --parts 1, --graphs 8/16/32), rocprofv3 kernel stats of the sampling
#    pass (one stream) and of the training step, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) reduced to HBM bytes
#    per launch.  Every GPU step runs under its own time limit and the chain stops at the first failure.
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
( while true; do sleep 45; echo "... $(date +%T)"; done ) &      # a line a minute: long profiler passes print nothing
HB=$!
trap "kill $HB 2>/dev/null" EXIT
B="--train-steps 0 --no-cpu-baseline"
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench done"
timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --parts 1 $B > $OUT/${TAG}_bench_parts1.json 2>> $OUT/${TAG}_bench.err
for g in 8 16 32; do
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --graphs $g $B --no-roofline > $OUT/${TAG}_bench_graphs$g.json 2>> $OUT/${TAG}_bench.err
done
echo "small batches done"
cd /tmp && export TMPDIR=/tmp
S="--steps 1 --warmup 0 --parts 1 --train-steps 0 --no-cpu-baseline --no-roofline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_sample -o s -- python3 $ROOT/bench.py $S > /dev/null 2>&1
echo "kernel trace (sampling) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train -o t -- python3 $ROOT/tools/train_probe.py > /dev/null 2>&1 || echo "train probe profile failed (non-fatal)"
echo "kernel trace (training) done"
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 $ROOT/bench.py $S > /dev/null 2>&1
echo "pmc fetch done"
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 $ROOT/bench.py $S > /dev/null 2>&1
echo "pmc write done"
cd $ROOT
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_traffic.json
# the counter CSVs are large: keep only the reduction
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
find $OUT/${TAG}_prof_sample $OUT/${TAG}_prof_train -name '*kernel_trace.csv' -delete 2>/dev/null || true
ls -la $OUT | grep $TAG
