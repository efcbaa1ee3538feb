"""GPU box: MLP-root forward, row kernel (4 sequences per wave, DPP) vs MFMA kernel (16 per wave) over batch sizes.
usage: python tools/mlp_fwd_probe.py [net]   (WDF_MLP_FWD_ROW=1 selects the row kernel inside the tp entry point)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "differentiable-wdfs_amd", "lib"))
import numpy as np, torch
from wdf_hip import binding as wb, workload

net = sys.argv[1] if len(sys.argv) > 1 else "2x16"
FS = 48000.0
wh, hidden, n_tanh = workload.reference_mlp_weights(net)
w = torch.tensor(wh, device="cuda")
th2 = torch.tensor([45.0e3, 4.7e-9], device="cuda")
T = 2048
for B in (1340, 4096, 16384, 65536):
    x = torch.tensor(workload.sweep_batch(min(B, 2048), T, seed=3) * 0.5, device="cuda", dtype=torch.float32).repeat(-(-B // 2048), 1)[:B].contiguous()
    r = torch.tensor(workload.dataset_resistance_batch(min(B, 2048), T), device="cuda", dtype=torch.float32).repeat(-(-B // 2048), 1)[:B].contiguous()
    for name, fn in (("sequential row kernel", lambda: wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, FS, r=r)),
                     ("tp entry, K=1", lambda: wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, FS, 1, 0, r=r)),
                     ("tp entry, K=1, kappa", lambda: wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, FS, 1, 0, r=r, want_kappa=True))):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f"{net} B={B:6d} {name:24s} {dt*1e3:8.3f} ms  {B*T/dt/1e9:6.2f} G samples/s", flush=True)
    y0 = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, FS, r=r)[0]
    y1 = wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, FS, 1, 0, r=r)[0]
    print("   max |y_tp - y_seq| =", float((y0 - y1).abs().max()))
