"""GPU: sweep chunk count / warm-up of the time-parallel kernels at the headline size."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "differentiable-wdfs_amd/lib")
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
for pack in (False, True):
    for K in (8, 16, 32, 64, 128):
        ws = torch.empty((wb.lib().wdf_clipper_bwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device="cuda")
        g = torch.empty(4, device="cuda")
        ms = timeit(lambda: wb.clipper_bwd_tp(x, th, fs, zs, gy, K, ws=ws, gtheta=g, pack=pack))
        print(f"bwd_tp pack={pack} K={K}: {ms:.3f} ms")
for pack in (False, True):
    for K, W in ((8, 192), (16, 192), (16, 64), (32, 192), (32, 64), (64, 192), (64, 64), (128, 64)):
        ws = torch.empty((wb.lib().wdf_clipper_fwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device="cuda")
        st = torch.empty(4, dtype=torch.int32, device="cuda")
        ms = timeit(lambda: wb.clipper_fwd_tp(x, th, fs, K, W, ws=ws, status=st, pack=pack))
        print(f"fwd_tp pack={pack} K={K} W={W}: {ms:.3f} ms  n_bad {wb.tp_status(st)['n_bad']} miss {wb.tp_status(st)['max_miss']:.1e}")
