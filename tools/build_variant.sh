#!/bin/bash
# Builds a differently tuned libmcpt into montecarlopathtracing_amd/csrc/variants/libmcpt_<name>.so (select it with MCPT_LIB):
#   bash tools/build_variant.sh <name> "<extra compiler flags, e.g. -DMCPT_TRACE_DIAG>"
# HIP_ONLY="..." in the environment: flags for the device compiler alone (e.g. -mllvm options); they are part of the variant's build id.
set -e
name=$1; shift
flags="$*"
hip_only="${HIP_ONLY:-}"
cd "$(dirname "$0")/../montecarlopathtracing_amd/csrc"
mkdir -p variants/obj_$name
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
base="-O3 -std=c++17 -fPIC -ffp-contract=off -pthread -Wall -Wno-unused-function -Wno-unused-result $flags"
logic_flags="${LOGIC_FLAGS--mllvm -disable-machine-licm}"      # as the Makefile's LOGICFLAGS (LOGIC_FLAGS= in the environment: none)
for f in kernels wavefront build_kernels; do
  $HIPCC $base $hip_only --offload-arch=gfx950 -c -o variants/obj_$name/$f.o $f.hip &
done
$HIPCC $base $hip_only $logic_flags --offload-arch=gfx950 -c -o variants/obj_$name/wavefront_logic.o wavefront_logic.hip &
for f in capi knobs scene_loader bvh_build accel_build png_writer output_formats jpeg_decoder multi_device proc_comm; do
  [ -f $f.cpp ] && g++ $base -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c -o variants/obj_$name/$f.o $f.cpp &
done
# mcpt_build_id() of a variant: the Makefile's recipe (sources + flags + arch) with this variant's extra flags, so that a variant never
# carries the product's id and a profile taken with it is never quoted for the product
srcs=$(ls *.cpp *.hip *.hpp | grep -v '^build_id.cpp$' | LC_ALL=C sort)
id=$( (cat $srcs ../../include/mcpt.h Makefile; echo "$flags $hip_only $logic_flags gfx950") | sha256sum | cut -c1-16)
echo "extern \"C\" const char* mcpt_build_id(void) { return \"$id\"; }" > variants/obj_$name/build_id.cpp
g++ $base -c -o variants/obj_$name/build_id.o variants/obj_$name/build_id.cpp &
wait
$HIPCC -shared -o variants/libmcpt_$name.so variants/obj_$name/*.o --offload-arch=gfx950
echo built variants/libmcpt_$name.so
