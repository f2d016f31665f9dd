# experiment builds (exp_libs/libedgehip_<V>.so) against the default library on fixed states
set -e
for V in base "$@"; do
  if [ $V = base ]; then unset EDGEHIP_LIB; else export EDGEHIP_LIB=$PWD/exp_libs/libedgehip_$V.so; fi
  for cfg in "100 64" "30 64" "300 64" "100 8"; do
    set -- $cfg
    echo "== $V t=$1 B=$2"
    timeout -k 10 90 python tools/kbench.py --t $1 --graphs $2 2>&1 | grep -E "layer|attention|graph_update|pair|total"
  done
done
