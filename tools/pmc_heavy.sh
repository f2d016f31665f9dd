set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
i=1
for P in "$P1" "$P3"; do
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $OUT/st_p$i -o p -- python3 $ROOT/tools/kbench.py --t 100 --reps 4 > /dev/null 2>&1
  i=$((i+1))
done
cd $ROOT
for i in 1 2; do python3 tools/pmc_reduce.py $OUT/st_p$i k_layer k_graph_update k_pair k_tail k_select k_graph_update k_step_prepare k_row_popcount; done
rm -rf $OUT/st_p1 $OUT/st_p2
