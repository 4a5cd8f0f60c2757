#!/bin/bash
# Build one variant of the attention object into fairygen_amd/csrc/build/ab/libfg_<name>.so (A/B measurements with tools/attn_ab.py):
#   tools/build_attn_variant.sh <name> "<gen_attn_w4.py args>" ["<extra hipcc flags>"]
set -e
cd "$(dirname "$0")/../fairygen_amd/csrc"
name=$1; genargs=$2; extra=$3
mkdir -p build/ab/inc_$name
python3 gen_attn_w4.py --prescale 0 --name FG_ATTN_W4 $genargs > build/ab/inc_$name/attn_w4_asm.inc
python3 gen_attn_w4.py --prescale 1 --name FG_ATTN_W4P $genargs > build/ab/inc_$name/attn_w4p_asm.inc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wall -Wno-unused-function -Ibuild/ab/inc_$name $extra -x hip -c attention.hip -o build/ab/attention_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libfg_$name.so build/capi.cpp.o build/dit_elementwise.hip.o build/ab/attention_$name.o build/dit_gemm.hip.o build/vae_conv.hip.o build/vae_ops.hip.o build/text_encoder.hip.o build/fp8_linear.hip.o
echo built build/ab/libfg_$name.so
