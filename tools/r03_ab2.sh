#!/bin/bash
# same-box A/B of two builds of the library inside the denoise step: FAIRYGEN_HIP_LIB=<alt> against the in-tree build, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
ALT=$1
cd $R
for rep in 1 2 3; do
  for which in base alt; do
    if [ $which = alt ]; then export FAIRYGEN_HIP_LIB=$R/$ALT; else unset FAIRYGEN_HIP_LIB; fi
    timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --skip-vae > gpurun_out/r03_ab2_${which}_$rep.json 2> gpurun_out/r03_ab2_${which}_$rep.err || exit 1
    python -c "
import json
d=json.loads(open('gpurun_out/r03_ab2_${which}_$rep.json').read().strip().splitlines()[-1]); print('$which', $rep, d['config']['denoise_ms_per_step'], [k['achieved'] for k in d['roofline']['kernels'] if 'gemm' in k['kernel']], d['roofline']['achieved'])"
  done
done
