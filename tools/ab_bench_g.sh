#!/bin/bash
# like tools/ab_bench.sh at a given batch size: bash tools/ab_bench_g.sh GRAPHS NAME [NAME ...]
set -e
G=$1; shift
for i in 1 2 3; do
  for V in base "$@"; do
    if [ $V = base ]; then unset EDGEHIP_LIB; else export EDGEHIP_LIB=$PWD/exp_libs/libedgehip_$V.so; fi
    timeout -k 10 300 python bench.py --graphs $G --steps 4 --warmup 1 --train-steps 0 --no-cpu-baseline --no-roofline > gpurun_out/ab_tmp.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$V', $G, round(d['value'], 2))"
  done
done
