set -e
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "attn or attention" 2>&1 | tail -2
echo default; timeout -k 10 120 python tools/microbench.py attn --iters 10
for v in k4v3 k4v2 e24 e24k4v3 e28 e32; do echo $v; FAIRYGEN_HIP_LIB=fairygen_amd/csrc/build/ab/libfg_$v.so timeout -k 10 120 python tools/microbench.py attn --iters 10; done
