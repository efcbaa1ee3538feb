"""Per-step timeline of the headline step from a rocprofv3 --kernel-trace run (round 6: chunk kernel + finish launch):
for the last `n` steps the median of {chunk kernel, gap to the finish launch, finish launch, gap to the next step's chunk kernel}.
usage: python tools/step_timeline.py <trace dir> [n]"""
import csv, glob, os, re, statistics, sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"wdf::(\w+)", r["Kernel_Name"])
        if m:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)))
rows.sort()
main = [i for i, r in enumerate(rows) if r[2] == "clipper_fused_tp_kernel"][-n - 1:-1]
out = {"chunk_kernel": [], "gap_to_next_launch": [], "second_launch": [], "gap_to_next_step": [], "start_to_start": []}
names = set()
for i in main:
    s, e, _ = rows[i]
    if i + 2 >= len(rows):
        continue
    s2, e2, n2 = rows[i + 1]
    s3, _, n3 = rows[i + 2]
    names.add(n2)
    out["chunk_kernel"].append((e - s) / 1e3)
    out["gap_to_next_launch"].append((s2 - e) / 1e3)
    out["second_launch"].append((e2 - s2) / 1e3)
    out["gap_to_next_step"].append((s3 - e2) / 1e3)
    out["start_to_start"].append((s3 - s) / 1e3)
print("second launch:", names, "steps:", len(out["chunk_kernel"]))
for k, v in out.items():
    print(f"{k:22s} median {statistics.median(v):8.2f} us   mean {statistics.fmean(v):8.2f}   min {min(v):8.2f}   max {max(v):8.2f}")
