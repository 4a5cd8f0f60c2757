#!/bin/bash
# Round-3 same-box A/B of the denoise step: DiT GEMM backends (bf16) and the fp8 Linear mode's matmul.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2; do
  for be in all fused; do
    FAIRYGEN_GEMM=$be timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > $O/r03_ab_${be}_$rep.json 2> $O/r03_ab_${be}_$rep.err || exit 1
  done
done
for be in own lib; do
  FAIRYGEN_FP8_GEMM=$be timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --linear-dtype fp8 > $O/r03_ab_fp8_${be}.json 2> $O/r03_ab_fp8_${be}.err || exit 1
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03_ab_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    c = d["config"]
    ks = [(k["kernel"][:28], k["achieved"], k.get("total_s")) for k in d["roofline"]["kernels"] if "gemm" in k["kernel"].lower() or "GEMM" in k["kernel"]]
    print(os.path.basename(f), "denoise ms/step", c["denoise_ms_per_step"], "vae", c["vae_decode_s"], "attn", d["roofline"]["achieved"], ks)
PY
