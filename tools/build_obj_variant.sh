#!/bin/bash
# Build one variant of a single library object into fairygen_amd/csrc/build/ab/libfgv_<name>.so:
#   tools/build_obj_variant.sh <name> <source.hip> "<extra hipcc flags>"      (select it with FAIRYGEN_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/../fairygen_amd/csrc"
name=$1; src=$2; extra=$3
mkdir -p build/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function -Ibuild $extra -x hip -c $src -o build/ab/${src}_$name.o
objs=""
for o in capi.cpp dit_elementwise.hip attention.hip dit_gemm.hip vae_conv.hip vae_ops.hip text_encoder.hip fp8_linear.hip; do
  if [ "$o" = "$src" ]; then objs="$objs build/ab/${src}_$name.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libfgv_$name.so $objs
echo built build/ab/libfgv_$name.so
