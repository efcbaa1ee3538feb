#!/bin/bash
# GPU box: collect the HBM-traffic counters of the bench command, one rocprofv3 pass per counter
# set (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in their own --pmc passes, --kernel-trace only),
# then summarise into gpurun_out/<tag>_pmc_traffic.json with tools/pmc_summarize.py.
# Pass --plan KF,W,KB so every pass profiles the same chunking (autotune may differ run to run).
# usage: bash tools/pmc_traffic.sh <tag> [bench args...]
set -u
TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  N=$(echo "$C" | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "gpurun_out/pmc_${TAG}_${N}" -o p -- \
      python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-batch-major --no-cold --no-sustained --no-fwd-1024 --no-strong-proxy "$@" > "gpurun_out/pmc_${TAG}_${N}.log" 2>&1
done
python tools/pmc_summarize.py "$TAG"
