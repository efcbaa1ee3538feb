"""GPU: the warm-started time-parallel forward inside a training loop (bench.py's step): per step the
warm-up tiles the device controller chose, the verify kernel's miss / re-runs, the forward kernel
time, and the distance of y from the sequential kernel at the same theta."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, engine, workload
B, T, fs = 8192, 4096, workload.FS
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lr_rel = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
dev = torch.device("cuda")
x = torch.as_tensor(workload.sweep_batch(B, T), device=dev)
xt = x.t().contiguous()
th_host = workload.clipper_theta()
tstar = torch.tensor(workload.target_theta(), dtype=torch.float32, device=dev)
target, _, _ = wb.clipper_fwd(x, tstar, fs, want_stash=False)
for K, KB in ((16, 32), (32, 32), (8, 32)):
    theta = torch.tensor(th_host, dtype=torch.float32, device=dev)
    tp = engine.TpPlan(K, 160, 1e-6, KB)
    st = engine.MseStep(B, T, fs, tp, dev, time_major=True, warm=True)
    adam = wb.Adam(4, lr=[lr_rel * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=dev)
    e0, e1 = wb.Event(), wb.Event()
    print(f"--- K={K} lr_rel={lr_rel}")
    for s in range(steps):
        wb.Event.bracket_next(e0, e1)
        st.forward(theta, xt)
        ms = e0.elapsed_ms(e1)
        stat = wb.tp_status(st.status)
        info = st.warm.info()
        err = ""
        if s in (0, 1, 2, 5, steps - 1):
            yref, _, _ = wb.clipper_fwd(xt, theta, fs, want_stash=False, time_major=True)
            err = f" max|y - y_seq| = {float((st.y - yref).abs().max()):.2e}"
        st.backward(theta, xt, target, adam=adam)
        print(f"step {s:2d}: fwd {ms*1e3:6.1f} us  tiles used {info['last_warm_tiles']:2d} -> next {info['next_warm_tiles']}  "
              f"miss {stat['max_miss']:.2e} bad {stat['n_bad']} reruns {stat['repaired_tiles']}{err}", flush=True)
