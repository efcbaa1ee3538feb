#!/bin/bash
# GPU box: profile of the resident one-pass steps (RC lowpass: linear; HPF clipper: diode root) in lpf.py's loop shape.
# usage: bash tools/prof_ss_step.sh TAG   ->  gpurun_out/<TAG>_ss_step_kernel_stats.csv, gpurun_out/<TAG>_ss_step.txt
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ONLY_RESIDENT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_ss_step -o p -- \
    python tools/ss_step_bench.py 200 > gpurun_out/${TAG}_ss_step_under_rocprof.txt 2> gpurun_out/prof_${TAG}_ss_step.err
cp "$(find gpurun_out/prof_${TAG}_ss_step -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_ss_step_kernel_stats.csv
ONLY_RESIDENT=1 python tools/ss_step_bench.py 200 2>/dev/null | grep -v amdgpu > gpurun_out/${TAG}_ss_step.txt
cat gpurun_out/${TAG}_ss_step.txt
python - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${TAG}_ss_step_kernel_stats.csv")))[:14]:
    print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
