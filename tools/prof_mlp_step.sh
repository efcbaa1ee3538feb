#!/bin/bash
# GPU box: the MLP-root resident training step's profile set.  usage: bash tools/prof_mlp_step.sh TAG [NET]
#   gpurun_out/<TAG>_<NET>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --root <NET> --steps 200 --warmup 60`
#   gpurun_out/<TAG>_bench_<NET>.json         the same command's line under the profiler; ..._plain.json: without it
TAG=${1:-r04}; NET=${2:-mlp2x16}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${NET} -o p -- \
    python bench.py --root $NET --steps 200 --warmup 60 --graph off --no-parity > gpurun_out/${TAG}_bench_${NET}_under_rocprof.json 2> gpurun_out/prof_${TAG}_${NET}.err
cp "$(find gpurun_out/prof_${TAG}_${NET} -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_${NET}_kernel_stats.csv
python bench.py --root $NET --steps 200 --warmup 60 > gpurun_out/${TAG}_bench_${NET}.json 2>> gpurun_out/prof_${TAG}_${NET}.err
python - <<PY
import csv, json
for r in list(csv.DictReader(open("gpurun_out/${TAG}_${NET}_kernel_stats.csv")))[:12]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
d = json.load(open("gpurun_out/${TAG}_bench_${NET}.json"))
print(d["ms_per_step"], d["step_launch"], d["parity"] and {k: v for k, v in d["parity"].items() if k != "checked"})
PY
