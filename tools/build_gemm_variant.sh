#!/bin/bash
# Build one variant of the DiT GEMM object into fairygen_amd/csrc/build/ab/libfgg_<name>.so (A/B measurements with tools/gemm_ab.py):
#   tools/build_gemm_variant.sh <name> "<gen_gemm_p.py args for the bf16 body>" ["<extra hipcc flags>"] ["<args for the e4m3 body>"]
#   (e.g. stamp "--stamp" "-DFG_GEMM_STAMP")
set -e
cd "$(dirname "$0")/../fairygen_amd/csrc"
name=$1; genargs=$2; extra=$3; qargs=$4
mkdir -p build/ab/ginc_$name
python3 gen_gemm_p.py --nb 4 --tail 1 $genargs > build/ab/ginc_$name/gemm_p41_asm.inc
python3 gen_gemm_p.py --nb 4 --tail 1 --dtype fp8 --budget 40 $qargs > build/ab/ginc_$name/gemm_q41_asm.inc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function -Ibuild/ab/ginc_$name $extra -x hip -c dit_gemm.hip -o build/ab/dit_gemm_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libfgg_$name.so build/capi.cpp.o build/dit_elementwise.hip.o build/attention.hip.o build/ab/dit_gemm_$name.o build/vae_conv.hip.o build/vae_ops.hip.o build/text_encoder.hip.o build/fp8_linear.hip.o
echo built build/ab/libfgg_$name.so
