set -e
for D in 2 4 6 8; do
  if [ $D = 4 ]; then unset EDGEHIP_LIB; else export EDGEHIP_LIB=$PWD/exp_libs/libedgehip_d$D.so; fi
  for cfg in "100 64" "250 64" "100 8"; do
    set -- $cfg
    echo "== D=$D t=$1 B=$2"
    timeout -k 10 90 python tools/kbench.py --t $1 --graphs $2 2>&1 | grep -E "layer|attention|total"
  done
done
