import sys, os, torch
sys.path.insert(0, "/root/repo/differentiable-wdfs_amd/lib")
from wdf_hip import binding as wb, engine, workload
B, T, fs = 8192, 4096, workload.FS
dev = torch.device("cuda")
x = torch.as_tensor(workload.sweep_batch(B, T), device=dev)
xt = x.t().contiguous()
th_host = workload.clipper_theta()
target, _, _ = wb.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device=dev), fs, want_stash=False)
for K in (32,):
    for j in (0, 1, 2, 3):
        theta = torch.tensor(th_host, dtype=torch.float32, device=dev)
        tp = engine.TpPlan(K, 192, 1e-6, 32)
        st = engine.MseStep(B, T, fs, tp, dev, time_major=True, warm=False)
        st.warm = wb.TpWarmState(B, T, K, max(j, 1), dev, min_warm_tiles=j)
        if j == 0:
            # pin to zero: max 1 but floor 0 -> the controller may move between 0 and 1; report what it does
            pass
        adam = wb.Adam(4, lr=[1e-3 * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=dev)
        e0, e1 = wb.Event(), wb.Event()
        miss, bad, ts, used = 0.0, 0, [], []
        for s in range(120):
            wb.Event.bracket_next(e0, e1)
            st.step_fused(theta, xt, target, adam=adam)
            ms = e0.elapsed_ms(e1)
            stat = wb.tp_status(st.status)
            info = st.warm.info()
            used.append(info["last_warm_tiles"])
            if s >= 20:
                miss, bad = max(miss, stat["max_miss"]), bad + stat["n_bad"]
                ts.append(ms)
        ts.sort()
        print(f"K={K} floor {j} max {max(j,1)}: max miss {miss:.2e} failed {bad} kernel median {ts[len(ts)//2]*1e3:6.1f} us min {ts[0]*1e3:6.1f}; tiles used (last 40) {used[-40:]}", flush=True)
