#!/bin/bash
# A/B builds of the library with extra -D flags:  scripts/build_variant.sh <name> "<flags>"  -> rpt_amd/lib/librptgpu_<name>.so
# (use with RPTGPU_LIB=$PWD/rpt_amd/lib/librptgpu_<name>.so; only the strict kernels + host are rebuilt with the flags,
# the other objects are taken from the regular build)
set -e
NAME=$1; FLAGS=$2
cd $(dirname $0)/../rpt_amd/csrc
make -s -j8
B=build/var_$NAME; mkdir -p $B
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off $FLAGS -c kernels_strict.hip -o $B/kernels_strict.o &
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off $FLAGS -x hip -c api.cpp -o $B/api.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/librptgpu_$NAME.so $B/kernels_strict.o $B/api.o build/kernels_strict_ext.o build/host_scene.o
echo built ../lib/librptgpu_$NAME.so
