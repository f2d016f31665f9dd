#!/bin/bash
# A/B of library variants over batch sizes / stream counts: bash tools/ab_sweep.sh "CFG;CFG;.." NAME [NAME ...]
# CFG = bench.py flags, e.g. "--graphs 8" or "--parts 1".  Variants: base = the default library.
set -e
IFS=';' read -ra CFGS <<< "$1"; shift
for cfg in "${CFGS[@]}"; do
  for V in base "$@"; do
    if [ $V = base ]; then unset EDGEHIP_LIB; else export EDGEHIP_LIB=$PWD/exp_libs/libedgehip_$V.so; fi
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --train-steps 0 --no-cpu-baseline --no-roofline $cfg > gpurun_out/ab_tmp.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$cfg | $V', round(d['value'], 2), 'graphs/s', round(d['ms_per_step'],1), 'ms')"
  done
done
