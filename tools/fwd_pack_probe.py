"""GPU: forward time-parallel kernel, one vs two sequences per lane (WDF_TP_PACK2), kernel-only time."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, workload
B, T, fs = 8192, 4096, workload.FS
x = torch.as_tensor(workload.sweep_batch(B, T), device="cuda")
xt = x.t().contiguous()
th = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = wb.Event(), wb.Event()
    ts = []
    for _ in range(n):
        wb.Event.bracket_next(e0, e1); fn(); ts.append(e0.elapsed_ms(e1))
    return sorted(ts)[len(ts)//2]
for tm, xin in ((True, xt), (False, x)):
    for K, W in ((8, 160), (16, 160), (32, 160), (64, 160)):
        ws = torch.empty((wb.lib().wdf_clipper_fwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device="cuda")
        st = torch.empty(4, dtype=torch.int32, device="cuda")
        for pack in (False, True):
            ms = timeit(lambda: wb.clipper_fwd_tp(xin, th, fs, K, W, ws=ws, status=st, time_major=tm, pack=pack))
            print(f"time_major={tm} K={K} W={W} pack={pack}: kernel {ms*1e3:.1f} us  n_bad {wb.tp_status(st)['n_bad']}")
