#!/bin/bash
# Round-3 measurement pass on the GPU box: default bench line (50 steps), rocprofv3 kernel stats of the same command at 2 steps, the
# fp8 Linear mode line, PMC passes (HBM bytes and MFMA busy) on the attention and GEMM kernels.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/r03c_bench_default_line.json 2> $O/r03c_bench_default.err || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 1 --no-cpu-baseline --linear-dtype fp8 > $O/r03c_bench_fp8_s10_line.json 2> $O/r03c_bench_fp8.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03c -o r03c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r03c_bench_steps2_line.json 2> $O/r03c_prof.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_r03_$c -o pmc -- python3 $R/tools/microbench.py attn --iters 3 > $O/r03_pmc_$c.log 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_r03_mfma -o pmc -- python3 $R/tools/microbench.py attn --iters 3 > $O/r03_pmc_mfma.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_r03_gemm -o pmc -- python3 $R/tools/gemm_ab.py --rounds 1 > $O/r03_pmc_gemm.log 2>&1 || exit 1
find $O/prof_r03c $O/pmc_r03_mfma $O/pmc_r03_gemm -name "*.csv" | head -20
