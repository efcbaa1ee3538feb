#!/usr/bin/env python3
"""Times the one-pass training step (wdf_clipper_step_mse_tp) against the two-kernel step at the headline
shape, over chunk counts, warm and cold: python tools/fused_probe.py [B T]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "differentiable-wdfs_amd", "lib"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wdf_hip import binding, engine, workload  # noqa: E402

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 4096)
fs = workload.FS
dev = torch.device("cuda", 0)
x = torch.as_tensor(workload.sweep_batch(B, T), device=dev)
xt = x.t().contiguous()
th_host = workload.clipper_theta()
tgt, _, _ = binding.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device=dev), fs, want_stash=False)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = binding.Event(), binding.Event()
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record()
        ts.append(e0.elapsed_ms(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


KS = [int(v) for v in os.environ.get("FUSED_K", "4,8,16,32,64").split(",")]
MODES = {"tm2": (True, False), "tm1": (True, True), "bm2": (False, False), "bm1": (False, True)}
for tm, one in [MODES[m] for m in os.environ.get("FUSED_MODES", "tm2,tm1,bm2").split(",")]:
    xk = xt if tm else x
    binding.ONE_SEQUENCE_PER_LANE = one
    print(f"--- x {'time' if tm else 'batch'}-major, {'one sequence' if one else 'two sequences'} per lane", flush=True)
    for K in KS:
        for warm in (False, True):
            if K >= 43 and not warm:
                continue
            plan = engine.TpPlan(K, int(os.environ.get("FUSED_W", "160")), 1e-6, 32)
            st = engine.MseStep(B, T, fs, plan, dev, time_major=tm, warm=warm, max_warm_tiles=int(os.environ.get("FUSED_MWT", "8")))
            theta = torch.tensor(th_host, dtype=torch.float32, device=dev)
            adam = binding.Adam(4, lr=[1e-3 * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=dev)
            med, best = timed(lambda: st.step_fused(theta, xk, tgt, adam=adam))
            s = binding.tp_status(st.status)
            w = st.warm.info() if st.warm is not None else None
            print(f"fused  tm={int(tm)} K={K:3d} warm={int(warm)}: median {med*1e3:7.1f} us  best {best*1e3:7.1f} us  "
                  f"{B*T/med/1e6:7.1f} G samples/s  n_bad={s['n_bad']} miss={s['max_miss']:.1e} "
                  f"warm_tiles={None if w is None else w['last_warm_tiles']}", flush=True)
    plan = engine.TpPlan(16, 160, 1e-6, 32)
    st = engine.MseStep(B, T, fs, plan, dev, time_major=tm, warm=True)
    theta = torch.tensor(th_host, dtype=torch.float32, device=dev)
    adam = binding.Adam(4, lr=[1e-3 * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=dev)

    def two():
        st.forward(theta, xk)
        st.backward(theta, xk, tgt, adam=adam)
    med, best = timed(two)
    print(f"2-kern tm={int(tm)} K=16/32 warm=1: median {med*1e3:7.1f} us  best {best*1e3:7.1f} us  {B*T/med/1e6:7.1f} G samples/s", flush=True)
