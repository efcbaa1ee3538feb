"""Per-kernel durations and the idle gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV."""
import csv, sys, statistics, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dur = collections.defaultdict(list); gap_after = collections.defaultdict(list)
for i, r in enumerate(rows):
    m = re.search(r"wdf::(\w+)", r["Kernel_Name"]); name = m.group(1) if m else r["Kernel_Name"][:40]
    if i < skip: continue
    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if i + 1 < len(rows):
        gap_after[name].append((int(rows[i + 1]["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3)
for n in dur:
    d, g = dur[n], gap_after[n]
    print(f"{n:40s} n={len(d):4d} dur med {statistics.median(d):8.1f} us  min {min(d):8.1f}  gap-after med {statistics.median(g) if g else 0:7.1f} us")
