#!/bin/bash
# One SQ counter pass over the fixed-state kernel bench (one reverse step at t = 100, 64 graphs) + the kbench timings at
# 64 and 8 graphs.  usage (through gpurun, repo root): bash tools/pmc_sq.sh TAG
set -e
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
for cfg in "100 64" "300 64" "100 8"; do set -- $cfg; timeout -k 10 90 python3 $ROOT/tools/kbench.py --t $1 --graphs $2 2>&1 | grep -v "phase shares\|inside B"; done | tee $OUT/${TAG}_kbench.txt
cd /tmp && export TMPDIR=/tmp
P="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $OUT/${TAG}_sqp -o p -- python3 $ROOT/tools/kbench.py --t 100 --reps 4 > /dev/null 2>&1
cd $ROOT
python3 tools/pmc_reduce.py $OUT/${TAG}_sqp k_layer k_graph_update | tee $OUT/${TAG}_sq_counters.txt
rm -rf $OUT/${TAG}_sqp
