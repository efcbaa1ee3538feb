"""GPU: forward time-parallel kernel time against warm-up length W and chunk count K with the
verification tolerance opened wide (no repair runs): what a warm-started forward could reach."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, workload
B, T, fs = 8192, 4096, workload.FS
x = torch.as_tensor(workload.sweep_batch(B, T), device="cuda")
xt = x.t().contiguous()
th = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda")
def timeit(fn, n=15):
    fn(); torch.cuda.synchronize()
    e0, e1 = wb.Event(), wb.Event()
    ts = []
    for _ in range(n):
        wb.Event.bracket_next(e0, e1); fn(); ts.append(e0.elapsed_ms(e1))
    return sorted(ts)[len(ts)//2]
for K in (8, 16, 32):
    ws = torch.empty((wb.lib().wdf_clipper_fwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device="cuda")
    st = torch.empty(4, dtype=torch.int32, device="cuda")
    for W in (0, 32, 64, 96, 160):
        for stash in (True,):
            ms = timeit(lambda: wb.clipper_fwd_tp(xt, th, fs, K, W, tol=1e30, ws=ws, status=st, time_major=True, want_stash=stash))
            print(f"K={K} W={W} stash={stash}: kernel {ms*1e3:.1f} us  miss {wb.tp_status(st)['max_miss']:.2e}", flush=True)
