#!/bin/bash
# Three counter passes over the fixed-state kernel bench (one reverse step at t = 100, 64 graphs): LDS and vector-memory
# instruction activity of the SQ, queue-full stalls, and the vector L1 / address unit.  One rocprofv3 --pmc run per pass.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD"
P2="SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $OUT/stall_p$i -o p -- python3 $ROOT/tools/kbench.py --t 100 --reps 4 > /dev/null 2>&1
  echo "pass $i done"
  i=$((i+1))
done
cd $ROOT
for i in 1 2 3; do python3 tools/pmc_reduce.py $OUT/stall_p$i k_layer k_graph_update; done > $OUT/r3_pmc_stalls.txt
rm -rf $OUT/stall_p1 $OUT/stall_p2 $OUT/stall_p3
cat $OUT/r3_pmc_stalls.txt
