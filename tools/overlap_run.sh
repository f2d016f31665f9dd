#!/bin/bash
# kernel traces of a two-stream and a one-stream bench pass + the overlap report (through gpurun, from the repo root)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
S="--steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline --no-roofline"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/ov2 -o s -- python3 $ROOT/bench.py $S > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/ov1 -o s -- python3 $ROOT/bench.py $S --parts 1 > /dev/null 2>&1
cd $ROOT
F2=$(find $OUT/ov2 -name '*kernel_trace.csv' | head -1); F1=$(find $OUT/ov1 -name '*kernel_trace.csv' | head -1)
python3 tools/overlap_probe.py $F2 $F1 | tee $OUT/overlap_report.txt
rm -rf $OUT/ov2 $OUT/ov1
