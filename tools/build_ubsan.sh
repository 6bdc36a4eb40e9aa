#!/bin/bash
# libptcore.so with UndefinedBehaviorSanitizer on the HOST code only (device code cannot be instrumented on this pool):
#   tools/build_ubsan.sh  ->  pathtracer_amd/variants/libptcore_ubsan.so
#   PTCORE_LIB=$PWD/pathtracer_amd/variants/libptcore_ubsan.so python -m pytest tests -m gpu     (any error aborts the run)
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
out=../variants; mkdir -p $out/obj_ubsan
F="--offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-function -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined"
hipcc $F -x hip -c pt_api.cpp -o $out/obj_ubsan/pt_api.o &
hipcc $F -x hip -c pt_bvh.cpp -o $out/obj_ubsan/pt_bvh.o &
hipcc $F -c pt_kernels.hip -o $out/obj_ubsan/pt_kernels.o &
hipcc $F -c pt_build.hip -o $out/obj_ubsan/pt_build.o &
wait
D=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux)
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libptcore_ubsan.so $out/obj_ubsan/*.o -Wl,--whole-archive \
  $D/libclang_rt.ubsan_standalone-x86_64.a $D/libclang_rt.ubsan_standalone_cxx-x86_64.a -Wl,--no-whole-archive -lpthread -ldl
rm -rf $out/obj_ubsan
echo built $out/libptcore_ubsan.so
