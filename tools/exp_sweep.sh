#!/bin/bash
# Times experiment builds of the library against the default one on fixed states (tools/kbench.py: one reverse step at
# a given t, repeated).  An experiment build is gnn.hip compiled with extra -D switches (EH_EXP_*: timing only, wrong
# results by construction; EH_WALK_DEPTH, EH_HEAVY_DEG, EH_HEAVY_BLOCKS, EH_PAIR_BLOCKS, EH_CPL: tunables) and linked
# with the other objects into exp_libs/libedgehip_<NAME>.so (git-ignored); kbench picks it up through EDGEHIP_LIB.
# Round 3: the EH_EXP_* / EH_STAMP instrumentation is no longer in csrc/gnn.hip; apply
# tools/experiments/k_layer_timing_switches.patch to a scratch copy of it first.
#   bash tools/exp_sweep.sh NAME [NAME ...]      (through gpurun, from the repo root)
set -e
for V in base "$@"; do
  if [ $V = base ]; then unset EDGEHIP_LIB; else export EDGEHIP_LIB=$PWD/exp_libs/libedgehip_$V.so; fi
  for cfg in "100 64" "30 64" "300 64" "100 8"; do
    set -- $cfg
    echo "== $V t=$1 B=$2"
    timeout -k 10 90 python tools/kbench.py --t $1 --graphs $2 2>&1 | grep -E "layer|attention|graph_update|pair|total"
  done
done
