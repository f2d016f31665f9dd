#!/bin/bash
# End-to-end A/B of experiment builds (exp_libs/libedgehip_<NAME>.so, see tools/exp_sweep.sh) against the default
# library: bench.py's sampling throughput, the variants in turn, three rounds, all on the same box (between boxes the
# bench moves by ~2 %, inside one call by ~0.1 %).
#   bash tools/ab_bench.sh NAME [NAME ...]
set -e
for i in 1 2 3; do
  for V in base "$@"; do
    if [ $V = base ]; then unset EDGEHIP_LIB; else export EDGEHIP_LIB=$PWD/exp_libs/libedgehip_$V.so; fi
    timeout -k 10 300 python bench.py --steps 4 --warmup 1 --train-steps 0 --no-cpu-baseline --no-roofline > gpurun_out/ab_tmp.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$V', round(d['value'], 2))"
  done
done
