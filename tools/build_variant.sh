#!/bin/bash
# Builds a differently tuned libmcpt into montecarlopathtracing_amd/csrc/variants/libmcpt_<name>.so (select it with MCPT_LIB):
#   bash tools/build_variant.sh <name> "<extra compiler flags, e.g. -DMCPT_TRACE_DIAG>"
set -e
name=$1; shift
flags="$*"
cd "$(dirname "$0")/../montecarlopathtracing_amd/csrc"
mkdir -p variants/obj_$name
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
base="-O3 -std=c++17 -fPIC -ffp-contract=off -pthread -Wall -Wno-unused-function -Wno-unused-result $flags"
for f in kernels wavefront build_kernels; do
  $HIPCC $base --offload-arch=gfx950 -c -o variants/obj_$name/$f.o $f.hip &
done
for f in capi scene_loader bvh_build accel_build png_writer output_formats jpeg_decoder multi_device build_id; do
  [ -f $f.cpp ] && g++ $base -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c -o variants/obj_$name/$f.o $f.cpp &
done
wait
$HIPCC -shared -o variants/libmcpt_$name.so variants/obj_$name/*.o --offload-arch=gfx950
echo built variants/libmcpt_$name.so
