#!/bin/bash
# Variant build of the product library: tools/build/v_<name>/liblargesteps_hip.so = the product's objects with the listed sources
# recompiled under extra flags.   usage: tools/build_variant.sh <name> "<extra hipcc flags>" [sources = direct.hip]
# A/B on the GPU box: LD_LIBRARY_PATH=tools/build/v_<name> tools/build/nd_drive ... (or LARGESTEPS_HIP_LIB=... for the Python package)
set -e
name=$1; extra=$2; shift 2; srcs=${@:-direct.hip}
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/large-steps-pytorch_amd/csrc
out=$root/tools/build/v_$name
mkdir -p $out
make -s -C $csrc -j8 >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
objs=""
for o in $csrc/build/*.o; do
  b=$(basename $o .o); skip=0
  for s in $srcs; do [ "${s%.hip}" = "$b" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for s in $srcs; do
  ( cd $csrc && /opt/rocm/bin/hipcc $FLAGS $extra -c $s -o $out/${s%.hip}.o ) &
  objs="$objs $out/${s%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $out/liblargesteps_hip.so $objs -ldl
echo "built $out/liblargesteps_hip.so ($extra)"
