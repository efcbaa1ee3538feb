"""GPU box: the MLP-root time-parallel forward call at B = 1340, T = 2048 over chunk counts (warm-up PROBE_W steps, default 0;
verification tolerance off: timing only).  WDF_MLP_FWD_ROW=1/0 picks the kernel."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "differentiable-wdfs_amd", "lib"))
import torch
from wdf_hip import binding as wb, workload

FS, B, T = 48000.0, 1340, 2048
wh, hidden, n_tanh = workload.reference_mlp_weights("2x16")
w = torch.tensor(wh, device="cuda")
th2 = torch.tensor([45.0e3, 4.7e-9], device="cuda")
x = torch.tensor(workload.sweep_batch(B, T, seed=3) * 0.5, device="cuda", dtype=torch.float32)
r = torch.tensor(workload.dataset_resistance_batch(B, T), device="cuda", dtype=torch.float32)
WARM = int(os.environ.get("PROBE_W", "0"))
for kap in (False, True):
    for K in (1, 2, 3, 6, 12, 24):
        fn = lambda: wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, FS, K, WARM, r=r, tol=1e9, want_kappa=kap)
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        L = -(-T // K)
        print(f"kappa={kap!s:5s} K={K:2d}  {dt*1e3:7.3f} ms   {dt*1e6/L:6.3f} us per step of a chunk", flush=True)
