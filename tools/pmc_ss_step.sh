#!/bin/bash
# GPU box: HBM traffic of the resident one-pass loops' kernels (tools/ss_step_bench.py), one rocprofv3 pass per counter
# (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in their own --pmc passes, --kernel-trace only) -> gpurun_out/<TAG>_ss_step_pmc.json
# (median per launch; bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE reports half the bytes on this box, profiles/r01_pmc_calibration_*).
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  ONLY_RESIDENT=1 REPS=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_ss_${C} -o p -- \
      python tools/ss_step_bench.py 40 > gpurun_out/pmc_${TAG}_ss_${C}.log 2>&1
done
python - <<PY
import csv, glob, json, re, statistics
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_${TAG}_ss_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"wdf::(\w+)", row["Kernel_Name"])
            if m and row["Counter_Name"] == c:
                per.setdefault(m.group(1), {}).setdefault(c, []).append(float(row["Counter_Value"]))
out = {"_doc": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/ss_step_bench.py 40 (resident loops, 8192 x 4096); "
               "median per launch, KiB; traffic_bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024", "samples": 8192 * 4096, "kernels": {}}
for k, c in per.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f, w = statistics.median(c["FETCH_SIZE"]), statistics.median(c["WRITE_SIZE"])
        out["kernels"][k] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w, "traffic_bytes": (2 * f + w) * 1024, "launches": len(c["FETCH_SIZE"]),
                             "bytes_per_sample": (2 * f + w) * 1024 / (8192 * 4096)}
json.dump(out, open("gpurun_out/${TAG}_ss_step_pmc.json", "w"), indent=1)
for k, v in out["kernels"].items():
    print(k, round(v["traffic_bytes"] / 1e6, 1), "MB", round(v["bytes_per_sample"], 2), "B/sample")
PY
