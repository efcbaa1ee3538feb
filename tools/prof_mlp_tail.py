"""Steady-state durations (last 12 calls) of the MLP-root kernels in a tools/prof_mlp.sh trace.  usage: prof_mlp_tail.py TAG"""
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/prof_{sys.argv[1]}/p_kernel_trace.csv")))
for name in ("fwd_tp_kernel", "wgrad_tp_kernel", "adjoint_scan", "clipper_mlp_row_fwd_kernel", "kappa_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    if d:
        print(f"{name:28s} n={len(d):4d} last 12 (us): {[round(x) for x in d[-12:]]}")
