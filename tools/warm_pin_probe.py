"""GPU: how many warm-up tiles the warm-started one-pass step really needs inside the bench's training loop: the
warm-up is pinned to j tiles (min = max), 40 Adam steps are run, and the largest boundary miss / failed
boundaries / kernel time after the first 4 steps are reported."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, engine, workload
B, T, fs = 8192, 4096, workload.FS
lr_rel = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
dev = torch.device("cuda")
x = torch.as_tensor(workload.sweep_batch(B, T), device=dev)
xt = x.t().contiguous()
th_host = workload.clipper_theta()
target, _, _ = wb.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device=dev), fs, want_stash=False)
for K in (16, 8):
    for j in (1, 2, 3, 4):
        theta = torch.tensor(th_host, dtype=torch.float32, device=dev)
        tp = engine.TpPlan(K, 160, 1e-6, 32)
        st = engine.MseStep(B, T, fs, tp, dev, time_major=True, warm=False)
        st.warm = wb.TpWarmState(B, T, K, j, dev, min_warm_tiles=j)
        adam = wb.Adam(4, lr=[lr_rel * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=dev)
        e0, e1 = wb.Event(), wb.Event()
        miss, bad, ts, hist = 0.0, 0, [], []
        for s in range(40):
            wb.Event.bracket_next(e0, e1)
            st.step_fused(theta, xt, target, adam=adam)
            ms = e0.elapsed_ms(e1)
            stat = wb.tp_status(st.status)
            hist.append(stat["max_miss"])
            if s >= 4:
                miss, bad = max(miss, stat["max_miss"]), bad + stat["n_bad"]
                ts.append(ms)
        ts.sort()
        print(f"K={K:2d} pinned warm-up {j} tiles: max miss {miss:.2e}  failed boundaries {bad}  kernel median {ts[len(ts)//2]*1e3:6.1f} us  "
              f"first misses {[f'{m:.1e}' for m in hist[:8]]}", flush=True)
