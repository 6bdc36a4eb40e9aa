#!/bin/bash
# build an A/B variant of libptcore.so: tools/build_variant.sh NAME [-DPT_...=...]...   -> pathtracer_amd/variants/libptcore_NAME.so
# Only pt_kernels.hip is rebuilt unless FULL=1 (needed for knobs pt_api.cpp sees: PT_TRACE_BLOCK, PT_LDS_STACK, PT_WAVES_PER_CU, PT_TRI_STRIDE).
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
name=$1; shift
out=../variants; mkdir -p $out/obj_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-function"
if [ -n "$FULL" ]; then
  hipcc $F "$@" -x hip -c pt_api.cpp -o $out/obj_$name/pt_api.o &
  hipcc $F "$@" -x hip -c pt_bvh.cpp -o $out/obj_$name/pt_bvh.o &
  hipcc $F "$@" -c pt_build.hip -o $out/obj_$name/pt_build.o &
else
  cp pt_api.o pt_bvh.o pt_build.o $out/obj_$name/
fi
hipcc $F "$@" -c pt_kernels.hip -o $out/obj_$name/pt_kernels.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libptcore_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/libptcore_$name.so
