#!/bin/bash
# usage: bash tools/build_variant.sh <name> [extra hipcc flags...]   ->  build_ab/libgsrast_<name>.so
# A build of the library with extra -D switches for same-box A/B runs (tools/ab_libs.sh, GSRAST_LIB); never loaded by the product.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=${SRC_DIR:-$root/taichi_3d_gaussian_splatting_amd/csrc}
obj=$root/build_ab/obj_$name
mkdir -p $obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
pids=()
for f in gs_api k_project k_binning k_blend_fwd k_backward k_export k_loss; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $src/$f.hip -o $obj/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_ab/libgsrast_$name.so $obj/*.o
echo built $root/build_ab/libgsrast_$name.so
