# GPU box: SQ activity counters of the bench command (one-pass step), two rocprofv3 passes; summarised per kernel.
# usage: bash tools/pmc_sq.sh <tag> [bench args...]
TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_sq1 -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-batch-major "$@" > gpurun_out/pmc_${TAG}_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_IFETCH --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_sq2 -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-batch-major "$@" > gpurun_out/pmc_${TAG}_sq2.log 2>&1
python - "$TAG" <<'PY'
import csv, glob, re, statistics, sys, json
tag = sys.argv[1]
per = {}
for f in glob.glob(f"gpurun_out/pmc_{tag}_sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"wdf::(\w+)", row["Kernel_Name"])
        if m:
            per.setdefault(m.group(1), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {k: {n: statistics.median(v) for n, v in c.items()} for k, c in per.items()}
json.dump(out, open(f"gpurun_out/{tag}_sq_counters.json", "w"), indent=1)
for k, c in out.items():
    if "fused_tp" in k or "fwd_tp" in k or "bwd_tp" in k:
        print(k, json.dumps(c))
PY
