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
for K, W in ((16, 160), (8, 160), (32, 160)):
    ws = torch.empty((wb.lib().wdf_clipper_fwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device="cuda")
    st = torch.empty(4, dtype=torch.int32, device="cuda")
    for stash in (True, False):
        ms = timeit(lambda: wb.clipper_fwd_tp(xt, th, fs, K, W, ws=ws, status=st, time_major=True, want_stash=stash))
        print(f"K={K} W={W} stash={stash}: kernel {ms*1e3:.1f} us")
