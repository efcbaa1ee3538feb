#!/bin/bash
# GPU box: the round's profile set -- rocprofv3 kernel stats of the bench command, the PMC traffic passes,
# and a plain bench line.  usage: bash tools/profile_round.sh <tag> <KF,W,KB>
set -u
TAG="$1"; PLAN="$2"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/prof_${TAG}" -o p -- \
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-batch-major --no-cold --plan "$PLAN" > "gpurun_out/prof_${TAG}_bench.json" 2> "gpurun_out/prof_${TAG}.err"
bash tools/pmc_traffic.sh "$TAG" --plan "$PLAN"
bash tools/pmc_sq.sh "$TAG" --plan "$PLAN"
python bench.py --plan "$PLAN" > "gpurun_out/bench_${TAG}.json" 2> "gpurun_out/bench_${TAG}.err"
