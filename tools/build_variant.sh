#!/bin/bash
# A variant of the library for a same-box A/B (tools/ab_libs.sh): the named translation units rebuilt with extra compiler flags, the rest
# taken from the current build.  bash tools/build_variant.sh <name> "<flags>" <tu> [<tu> ...]  ->  npbnn_amd/lib/variants/lib<name>.so
# (git-ignored like every .so, travels to the GPU box with the snapshot).
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=/tmp/npbnn_variant_$NAME
rm -rf $OBJ && mkdir -p $OBJ $ROOT/npbnn_amd/lib/variants
cp $ROOT/npbnn_amd/csrc/build/*.o $OBJ/
cd $ROOT/npbnn_amd/csrc
for TU in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=4 $FLAGS -c -o $OBJ/$TU.o $TU.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $ROOT/npbnn_amd/lib/variants/lib$NAME.so $OBJ/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la $ROOT/npbnn_amd/lib/variants/lib$NAME.so
