#!/bin/bash
# A/B builds of the library with extra -D flags:  scripts/build_variant.sh <name> "<flags>"  -> rpt_amd/lib/librptgpu_<name>.so
# (use with RPTGPU_LIB=$PWD/rpt_amd/lib/librptgpu_<name>.so; both kernel builds and the api_*.cpp files are rebuilt with the flags,
# host_scene.o is taken from the regular build).  NOEXT=1 skips the extended-shape build (takes the regular one).
set -e
NAME=$1; FLAGS=$2
cd $(dirname $0)/../rpt_amd/csrc
make -s -j8
B=build/var_$NAME; mkdir -p $B
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off $FLAGS -c kernels_strict.hip -o $B/kernels_strict.o &
if [ -z "$NOEXT" ]; then /opt/rocm/bin/hipcc $COMMON -ffp-contract=off $FLAGS -c kernels_strict_ext.hip -o $B/kernels_strict_ext.o & else cp build/kernels_strict_ext.o $B/; fi
API="api_common api_scene api_render api_comm api_buffer"
for a in $API; do /opt/rocm/bin/hipcc $COMMON -ffp-contract=off $FLAGS -x hip -c $a.cpp -o $B/$a.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/librptgpu_$NAME.so $B/kernels_strict.o $(for a in $API; do echo $B/$a.o; done) $B/kernels_strict_ext.o build/host_scene.o build/kdbuild.o
echo built ../lib/librptgpu_$NAME.so
