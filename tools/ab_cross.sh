for i in 1 2; do
python tools/microbench.py attn --nq 27280 --nkv 512 --iters 30 2>/dev/null
FAIRYGEN_HIP_LIB=fairygen_amd/csrc/build/ab/libfg_v4.so python tools/microbench.py attn --nq 27280 --nkv 512 --iters 30 2>/dev/null
done
FAIRYGEN_HIP_LIB=fairygen_amd/csrc/build/ab/libfg_v4.so python tools/microbench.py attn --iters 10 2>/dev/null
