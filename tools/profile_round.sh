#!/bin/bash
# GPU box: the round's profile set of the HEADLINE, all on one build and one pinned plan:
#   1. a plain bench line (its autotuned chunk count and settled warm-up name the plan the passes are pinned to, unless given)
#   2. rocprofv3 --kernel-trace --stats of the bench command            -> <tag>_kernel_stats.csv, <tag>_kernel_steady.json
#      (steady state only: the untimed warm-up launches dropped; min / median / mean / max)
#   3. PMC traffic passes (FETCH_SIZE, WRITE_SIZE: separate --pmc passes)  -> <tag>_pmc_traffic.json
#   4. SQ activity passes                                               -> <tag>_sq_counters.json
# Every file is stamped with the sha of the library the run loaded (bench.py refuses a pass of another build).
# usage: bash tools/profile_round.sh <tag> [KF,W,KB] [steps]
set -u
TAG="$1"; PLAN="${2:-}"; STEPS="${3:-200}"; WARM=20
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python bench.py ${PLAN:+--plan "$PLAN"} > "gpurun_out/bench_${TAG}.json" 2> "gpurun_out/bench_${TAG}.err"
if [ -z "$PLAN" ]; then
  PLAN=$(python - "gpurun_out/bench_${TAG}.json" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
tp = d["config"]["time_parallel"]
print(f'{tp["fwd_chunks"]},{tp["fwd_warmup_steps"]},{tp["bwd_chunks"] or 16}')
PY
)
fi
echo "plan $PLAN"
rm -rf "gpurun_out/prof_${TAG}"
rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/prof_${TAG}" -o p -- \
    python bench.py --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline --no-parity --no-batch-major --no-cold --no-sustained --no-fwd-1024 --no-strong-proxy --plan "$PLAN" \
    > "gpurun_out/prof_${TAG}_bench.json" 2> "gpurun_out/prof_${TAG}.err"
cp "$(find gpurun_out/prof_${TAG} -name '*kernel_stats.csv' | head -1)" "gpurun_out/${TAG}_kernel_stats.csv"
python tools/kernel_steady.py "$TAG" "$WARM"
bash tools/pmc_traffic.sh "$TAG" --plan "$PLAN"
bash tools/pmc_sq.sh "$TAG" --plan "$PLAN"
# the line again, now that the passes of THIS build exist next to it (bench.py looks in profiles/ and gpurun_out/)
python bench.py --plan "$PLAN" > "gpurun_out/bench_${TAG}_pinned.json" 2> "gpurun_out/bench_${TAG}_pinned.err"
python bench.py --plan "$PLAN" --steps 20 --warmup 3 > "gpurun_out/bench_${TAG}_driver_form.json" 2> "gpurun_out/bench_${TAG}_driver_form.err"
