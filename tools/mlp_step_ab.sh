#!/bin/bash
# tools/mlp_step_ab.sh -- reverse-sweep variants of the resident MLP-root step on one box: staging block x chunk count
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04a
for bs in 16 8 4; do for kw in 24 32 43 64; do
  echo "== BS=$bs KW=$kw"; WDF_MLP_STEP_BS=$bs KW=$kw REPLAN=1 timeout 120 python tools/mlp_step_bench.py 2x16_pre 100 2>&1 | grep -E "kernels:|eager after|HIP graph, 100 replays"
done; done
