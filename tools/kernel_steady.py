"""Steady-state kernel durations of a rocprofv3 --kernel-trace run of bench.py (tools/profile_round.sh):
gpurun_out/<tag>_kernel_steady.json -- per wdf:: kernel the launches sorted by start time, the first `drop` (the run's
untimed warm-up steps: the cold call and the warm-start controller's descent) left out, n / min / median / mean / max of
the rest in microseconds -- stamped with the library the run loaded (the bench line's `library` block) and carrying the
line's own ms_per_step and HIP-event kernel times, so the three clocks can be compared from one file.
usage: python tools/kernel_steady.py <tag> <drop> [bench line file]"""
import csv, glob, json, os, re, statistics, sys

tag, drop = sys.argv[1], int(sys.argv[2])
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
line_file = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, f"prof_{tag}_bench.json")
rows = {}
for f in glob.glob(os.path.join(root, f"prof_{tag}", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"wdf::(\w+)", r["Kernel_Name"])
        if m:
            rows.setdefault(m.group(1), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
line = None
for l in open(line_file):
    if l.startswith("{") and '"metric"' in l:
        line = json.loads(l)
out = {"_doc": "rocprofv3 --kernel-trace of `python bench.py --steps K --warmup W --plan ...` (tools/profile_round.sh); per kernel: launches "
               f"in start order, the first {drop} dropped (untimed warm-up steps), durations in microseconds.  `all_launches` keeps "
               "the full-run average rocprofv3 --stats reports (it includes the cold calls).",
       "library": None if line is None else line.get("library"),
       "bench": None if line is None else {"ms_per_step": line["ms_per_step"], "steps": line["steps"], "warmup": line["warmup"],
                                           "value": line["value"], "kernel_ms_in_region": line.get("kernel_ms_in_region"),
                                           "kernel_ms": line.get("kernel_ms"),
                                           "time_parallel": line["config"].get("time_parallel")},
       "kernels": {}}
for k, v in rows.items():
    v.sort()
    d_all = [(e - s) / 1e3 for s, e in v]
    d = d_all[drop:] if len(d_all) > drop + 2 else d_all
    st = [s for s, _ in v][drop:] if len(v) > drop + 2 else [s for s, _ in v]
    gaps = [(st[i + 1] - st[i]) / 1e3 for i in range(len(st) - 1)]
    out["kernels"][k] = {"n": len(d), "min_us": min(d), "median_us": statistics.median(d), "mean_us": statistics.fmean(d), "max_us": max(d),
                         "spread_pct": 100.0 * (max(d) - min(d)) / statistics.median(d),
                         "start_to_start_median_us": statistics.median(gaps) if gaps else None,
                         "all_launches": {"n": len(d_all), "mean_us": statistics.fmean(d_all), "max_us": max(d_all)}}
path = os.path.join(root, f"{tag}_kernel_steady.json")
json.dump(out, open(path, "w"), indent=1)
for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["mean_us"] * kv[1]["n"])[:6]:
    print(f"{k:40s} n={e['n']:4d} min {e['min_us']:8.2f} median {e['median_us']:8.2f} mean {e['mean_us']:8.2f} max {e['max_us']:8.2f} us "
          f"(all launches: mean {e['all_launches']['mean_us']:.2f})")
print(path)
