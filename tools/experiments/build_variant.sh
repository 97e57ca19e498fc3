#!/bin/bash
# usage: tools/experiments/build_variant.sh NAME "-DFLAG=1 ..." file1.hip [file2.hip ...]   -> easykv_amd/csrc/variants/lib_NAME.so
# (A/B builds for the experiment drivers in this directory: select one with EASYKV_HIP_LIB=...; *.so is git-ignored but travels with gpurun;
#  objects of the untouched sources are reused from csrc/obj)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT/easykv_amd/csrc
NAME=$1; FLAGS=$2; shift 2
mkdir -p /tmp/ekv_var/obj_$NAME $ROOT/easykv_amd/csrc/variants
OBJS=""
for o in obj/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do if [ "$b" == "$(basename $f .hip)" ]; then use=/tmp/ekv_var/obj_$NAME/$b.o; fi; done
  OBJS="$OBJS $use"
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $FLAGS -c $f -o /tmp/ekv_var/obj_$NAME/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/easykv_amd/csrc/variants/lib_$NAME.so
echo built easykv_amd/csrc/variants/lib_$NAME.so
