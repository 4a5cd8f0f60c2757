set -e
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "attn or attention" 2>&1 | tail -2
for i in 1 2; do
echo default; timeout -k 10 120 python tools/microbench.py attn --iters 20 2>/dev/null
for v in $AB_VARIANTS; do echo $v; FAIRYGEN_HIP_LIB=fairygen_amd/csrc/build/ab/libfg_$v.so timeout -k 10 120 python tools/microbench.py attn --iters 20 2>/dev/null; done
done
