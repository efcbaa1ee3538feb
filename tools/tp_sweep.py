"""GPU: sweep chunk count / warm-up of the time-parallel kernels at the headline size."""
import sys, time
import numpy as np, torch
import os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, workload

B, T, fs = 8192, 4096, workload.FS
x = torch.as_tensor(workload.sweep_batch(B, T), device="cuda")
th = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda")
tgt, _, _ = wb.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device="cuda"), fs, want_stash=False)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = wb.Event(), wb.Event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    return e0.elapsed_ms(e1) / n


y, zs, _ = wb.clipper_fwd(x, th, fs)
gy = (2.0 * (y - tgt) / y.numel()).contiguous()
print("seq fwd ms", timeit(lambda: wb.clipper_fwd(x, th, fs)), " seq bwd ms", timeit(lambda: wb.clipper_bwd(x, th, fs, zs, gy)))
zT = wb.clipper_fwd(x, th, fs, want_zT=True)[2]
xt = x.t().contiguous()
gscale = 2.0 / y.numel()
for tm, xin in ((False, x), (True, xt)):
    for K in (8, 16, 32, 64):
        ws = wb.bwd_tp_workspace(B, K, "cuda")
        g = torch.empty(4, device="cuda"); sse = torch.empty(1, device="cuda")
        ms = timeit(lambda: wb.clipper_bwd_mse_tp(xin, th, fs, zs, zT, tgt, gscale, K, ws=ws, gtheta=g, sse=sse, time_major=tm))
        print(f"bwd_mse_tp time_major={tm} K={K}: {ms:.3f} ms")
for tm, xin in ((False, x), (True, xt)):
    for K, W in ((8, 192), (16, 192), (16, 160), (16, 128), (24, 192), (32, 192), (32, 160), (32, 128), (64, 192)):
        ws = torch.empty((wb.lib().wdf_clipper_fwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device="cuda")
        st = torch.empty(4, dtype=torch.int32, device="cuda")
        ms = timeit(lambda: wb.clipper_fwd_tp(xin, th, fs, K, W, ws=ws, status=st, time_major=tm))
        print(f"fwd_tp time_major={tm} K={K} W={W}: {ms:.3f} ms  n_bad {wb.tp_status(st)['n_bad']} miss {wb.tp_status(st)['max_miss']:.1e}")
