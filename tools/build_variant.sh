#!/bin/bash
# Builds exp_libs/libedgehip_<NAME>.so: gnn.hip compiled with extra flags, linked with the default objects.
#   bash tools/build_variant.sh NAME "<extra hipcc flags>"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/exp_libs
cd $ROOT/graph-generation-edge_amd/csrc
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$ROOT/include -I. -Wall -Wno-unused-function \
  -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c gnn.hip -o $ROOT/exp_libs/gnn_$NAME.o
L=$ROOT/graph-generation-edge_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/exp_libs/libedgehip_$NAME.so $L/state.o $L/gemm.o \
  $ROOT/exp_libs/gnn_$NAME.o $L/train.o $L/stats.o $L/prof.o
rm -f $ROOT/exp_libs/gnn_$NAME.o
echo built exp_libs/libedgehip_$NAME.so
