#!/bin/bash
# Round-2 (second session) measurement pass on the GPU box: elementwise microbench, default bench line, rocprofv3 kernel stats, PMC passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 200 python tools/microbench.py elementwise --iters 20 > $O/r02c_elementwise.log 2>&1
timeout -k 10 400 python bench.py > $O/r02c_bench_default_line.json 2> $O/r02c_bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02c -o r02c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r02c_bench_steps2_line.json 2> $O/r02c_prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_r02c_$c -o pmc -- python3 $R/tools/microbench.py attn --iters 3 > $O/r02c_pmc_$c.log 2>&1
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_r02c_mfma -o pmc -- python3 $R/tools/microbench.py attn --iters 3 > $O/r02c_pmc_mfma.log 2>&1
ls -R $O/prof_r02c $O/pmc_r02c_FETCH_SIZE | head -30
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_r02c_gemm -o pmc -- python3 $R/tools/gemm_ab.py --only p --rounds 1 > $O/r02c_pmc_gemm.log 2>&1
ls $O/pmc_r02c_gemm | head
