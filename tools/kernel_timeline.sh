cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/prof_timeline
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_timeline -o p -- python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-parity --no-cold --no-batch-major --no-strong-proxy > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_timeline/**/p_kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "clipper_fused_tp_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-303:]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
st=[int(r["Start_Timestamp"]) for r in rows]
per=[(st[i+1]-st[i])/1e3 for i in range(len(st)-1)]
print("kernel us, by step (every step for the first 30, then means of 10):")
print(" ".join(f"{x:.1f}" for x in d[:30]))
print(" ".join(f"{sum(d[i:i+10])/10:.1f}" for i in range(30,len(d)-9,10)))
print("start-to-start us:")
print(" ".join(f"{x:.1f}" for x in per[:30]))
print(" ".join(f"{sum(per[i:i+10])/10:.1f}" for i in range(30,len(per)-9,10)))
PY
