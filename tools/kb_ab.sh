#!/bin/bash
# fixed-state kernel timings (tools/kbench.py) of library variants: bash tools/kb_ab.sh "T B;T B" NAME [NAME..]
CF=$1; shift
IFS=';' read -ra CFGS <<< "$CF"
for cfg in "${CFGS[@]}"; do
  read T B <<< "$cfg"
  for V in base "$@"; do
    if [ $V = base ]; then unset EDGEHIP_LIB; else export EDGEHIP_LIB=$PWD/exp_libs/libedgehip_$V.so; fi
    echo "== $V t=$T B=$B"
    timeout -k 10 90 python tools/kbench.py --t $T --graphs $B 2>&1 | grep -E "layer|attention|graph_update|total"
  done
done
