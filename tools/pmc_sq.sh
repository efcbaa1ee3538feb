# GPU box: SQ activity counters of the bench command (one-pass step), two rocprofv3 passes; summarised per kernel.
# usage: bash tools/pmc_sq.sh <tag> [bench args...]
TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_sq1 -o p -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-batch-major --no-cold --no-sustained --no-fwd-1024 --no-strong-proxy "$@" > gpurun_out/pmc_${TAG}_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_IFETCH --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_sq2 -o p -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-batch-major --no-cold --no-sustained --no-fwd-1024 --no-strong-proxy "$@" > gpurun_out/pmc_${TAG}_sq2.log 2>&1
python - "$TAG" <<'PY'
import csv, glob, re, statistics, sys, json
tag = sys.argv[1]
per = {}
for f in glob.glob(f"gpurun_out/pmc_{tag}_sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"wdf::(\w+)", row["Kernel_Name"])
        if m:
            per.setdefault(m.group(1), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
kernels = {k: {n: statistics.median(v) for n, v in c.items()} for k, c in per.items()}
cfg = library = None                          # the configuration and the build the profiled run used (bench.py matches on both)
for line in open(f"gpurun_out/pmc_{tag}_sq1.log"):
    if line.startswith("{") and '"metric"' in line:
        d = json.loads(line)
        library = d.get("library")
        tp = d["config"]["time_parallel"]
        ws = tp.get("warm_start")
        cfg = {"B": d["config"]["global_batch"] // d["n_gpus"], "T": d["config"]["seq_len"],
               "x_layout": "time-major" if d["config"]["x_layout"].startswith("time-major") else "batch-major",
               "loss": "mse+esr" if "MSE+ESR" in d["config"]["workload"] else "mse", "fused_chunks": tp["fwd_chunks"],
               "fwd_warmup_steps": tp["fwd_warmup_steps"] if not ws else ws.get("warm_unit_steps", 32) * max(0, ws["last_warm_tiles"])}
out = {"_doc": "rocprofv3 --pmc SQ_* (two passes, --kernel-trace) of `python bench.py --steps 200 --warmup 20 ...` on MI355X "
               "(tools/pmc_sq.sh); median per launch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed "
               "over waves (MI355X_MICROARCH.md).", "config": cfg, "library": library, "kernels": kernels}
json.dump(out, open(f"gpurun_out/{tag}_sq_counters.json", "w"), indent=1)
for k, c in kernels.items():
    if "fused_tp" in k or "fwd_tp" in k or "bwd_tp" in k:
        print(k, json.dumps(c))
PY
