"""Summarise the PMC passes of tools/pmc_traffic.sh into gpurun_out/<tag>_pmc_traffic.json:
per kernel, the median counter value per launch and HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB
(FETCH_SIZE reports exactly half the bytes on this box for every access pattern we use:
profiles/r01_pmc_calibration_*.csv, tools/pmc_calibrate.py)."""
import csv, glob, json, os, re, statistics, sys

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
per = {}
for d in glob.glob(os.path.join(root, f"pmc_{tag}_*")):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"wdf::(\w+)", row["Kernel_Name"])
            if not m:
                continue
            per.setdefault(m.group(1), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
# the bench line of any pass names the configuration the counters belong to
cfg = library = None
for log in glob.glob(os.path.join(root, f"pmc_{tag}_*.log")):
    for line in open(log):
        if line.startswith("{") and '"metric"' in line:
            d = json.loads(line)
            library = d.get("library")
            tp = d["config"]["time_parallel"]
            fused = str(d.get("step_kernels", "")).startswith("one pass")
            cfg = {"B": d["config"]["global_batch"] // d["n_gpus"], "T": d["config"]["seq_len"],
                   "x_layout": "time-major" if d["config"]["x_layout"].startswith("time-major") else "batch-major",
                   "loss": "mse+esr" if "MSE+ESR" in d["config"]["workload"] else "mse",
                   **({"fused_chunks": tp["fwd_chunks"]} if fused else {"fwd_chunks": tp["fwd_chunks"], "bwd_chunks": tp["bwd_chunks"]}),
                   # warm-started forward: the warm-up the device controller settled at, not the cold one
                   "fwd_warmup_steps": tp["fwd_warmup_steps"] if not tp.get("warm_start") else tp["warm_start"].get("warm_unit_steps", 32) * max(0, tp["warm_start"]["last_warm_tiles"])}
out = {"_doc": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ_* (separate passes, --kernel-trace) of `python bench.py "
               "--steps 200 --warmup 20 --no-cpu-baseline ...` on MI355X (tools/pmc_traffic.sh); median per launch. Units KiB. "
               "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE reports exactly half the bytes on this box "
               "(calibration: profiles/r01_pmc_calibration_*.csv), WRITE_SIZE is exact.",
       "config": cfg, "library": library, "kernels": {}}
for k, c in per.items():
    med = {n: statistics.median(v) for n, v in c.items()}
    e = {"FETCH_SIZE_KiB_raw": med.get("FETCH_SIZE"), "fetch_correction": 2.0, "WRITE_SIZE_KiB": med.get("WRITE_SIZE")}
    if e["FETCH_SIZE_KiB_raw"] is not None and e["WRITE_SIZE_KiB"] is not None:
        e["traffic_bytes"] = (2.0 * e["FETCH_SIZE_KiB_raw"] + e["WRITE_SIZE_KiB"]) * 1024.0
    e["SQ"] = {n: v for n, v in med.items() if n.startswith("SQ_")}
    out["kernels"][k] = e
path = os.path.join(root, f"{tag}_pmc_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps({k: v.get("traffic_bytes") for k, v in out["kernels"].items()}), cfg)
