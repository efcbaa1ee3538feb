cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in ${AB_LIBS:-libwdf_hip.so}; do
  rm -rf gpurun_out/prof_ab_$L
  WDF_HIP_LIB=$PWD/differentiable-wdfs_amd/lib/wdf_hip/$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$L -o p -- python bench.py --steps 200 --warmup 20 --no-optimizer --plan 32,192,32 --no-cpu-baseline --no-parity --no-batch-major --no-cold > /dev/null 2>&1
  python - $L <<'PY'
import csv,glob,sys
L=sys.argv[1]
f=glob.glob(f"gpurun_out/prof_ab_{L}/**/p_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
def dur(name):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in rows if name in r["Kernel_Name"]]
    d=d[-150:]
    d.sort(); return len(d), d[len(d)//2]/1e3, d[len(d)//10]/1e3
print(L, "fused main (n, median us, p10)", dur("clipper_fused_tp_kernel"), "repair", dur("clipper_fused_repair_kernel"))
# gaps: start of main minus end of previous repair, start of repair minus end of main (last 150 steps)
ks=[r for r in rows if "clipper_fused" in r["Kernel_Name"]][-300:]
g1=[];g2=[]
for a,b in zip(ks,ks[1:]):
    gap=(int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3
    (g1 if "repair" in b["Kernel_Name"] else g2).append(gap)
g1.sort();g2.sort()
print("   gap main->repair median %.2f us; repair->next main median %.2f us" % (g1[len(g1)//2], g2[len(g2)//2]))
PY
done
