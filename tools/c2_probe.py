"""GPU: the stateless forward at BASELINE configs[1]'s shape (1024 x 4096) over chunk counts; WDF_FWD_PAIR=0/1 picks the kernel."""
import os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, engine, workload
B, T, FS = 1024, 4096, workload.FS
th = workload.clipper_theta()
x = torch.as_tensor(workload.sweep_batch(B, T, seed=3), device="cuda").t().contiguous()
theta = torch.tensor(th, dtype=torch.float32, device="cuda")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 160
for k in [int(v) for v in os.environ.get("C2_KS", "16,32,42,64,84,128").split(",")]:
    used = wb.lib().wdf_clipper_tp_chunks(T, k)
    f = lambda: wb.clipper_fwd_tp(x, theta, FS, k, W, 1e-6, want_stash=False, time_major=True)
    y, _, _, st = f(); torch.cuda.synchronize()
    s = wb.tp_status(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"pair={os.environ.get('WDF_FWD_PAIR','1')} chunks asked {k:4d} used {used:4d}: {e0.elapsed_time(e1)/50*1e3:7.1f} us  n_bad {s['n_bad']} max_miss {s['max_miss']:.1e}")
