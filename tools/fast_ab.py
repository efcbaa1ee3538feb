"""GPU: A/B on one box -- the kernels' default step vs WDF_GENERAL_ROOT (per-step ballot, lam as
sign(a)), kernel-only times at the headline size, plus the largest output difference."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, workload
B, T, fs = 8192, 4096, workload.FS
x = torch.as_tensor(workload.sweep_batch(B, T), device="cuda")
xt = x.t().contiguous()
th = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda")
tgt, _, _ = wb.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device="cuda"), fs, want_stash=False)
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = wb.Event(), wb.Event()
    ts = []
    for _ in range(n):
        wb.Event.bracket_next(e0, e1); fn(); ts.append(e0.elapsed_ms(e1))
    return sorted(ts)[len(ts)//2]
K, W, KB = 16, 160, 32
ws = torch.empty((wb.lib().wdf_clipper_fwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device="cuda")
st = torch.empty(4, dtype=torch.int32, device="cuda")
wsb = wb.bwd_tp_workspace(B, KB, "cuda")
g = torch.empty(4, device="cuda"); sse = torch.empty(1, device="cuda")
res = {}
for rep in range(2):
    for general in (False, True):
        wb.GENERAL_ROOT = general
        y, zs, zT, _ = wb.clipper_fwd_tp(xt, th, fs, K, W, ws=ws, status=st, time_major=True, want_zT=True)
        tf = timeit(lambda: wb.clipper_fwd_tp(xt, th, fs, K, W, ws=ws, status=st, time_major=True))
        tb = timeit(lambda: wb.clipper_bwd_mse_tp(xt, th, fs, zs, zT, tgt, 2.0 / y.numel(), KB, ws=wsb, gtheta=g, sse=sse, time_major=True))
        ts = timeit(lambda: wb.clipper_fwd(x, th, fs), n=5)
        res[general] = (y.clone(), g.clone())
        print(f"general={general}: fwd_tp {tf*1e3:.1f} us, bwd_mse_tp {tb*1e3:.1f} us, sequential fwd {ts*1e3:.0f} us")
print("max |y_fast - y_general| =", float((res[False][0] - res[True][0]).abs().max()),
      " grad rel diff =", float(((res[False][1] - res[True][1]).abs() / res[True][1].abs()).max()))
