"""GPU: BASELINE configs[3] shape on one GPU -- the dataset-style training step with the pot resistance
streamed per sample (clipper_pot.py:116): 1340 sequences of 2048 tiled to 8192 per GPU, R per sequence on
the reference's file-name grid {10k, 25.2k, 75k, 99.1k} -- fused MSE step, samples/s."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, engine, workload
B, T, fs = 8192, 2048, workload.FS
reps = -(-B // 1340)
x = torch.as_tensor(np.tile(workload.sweep_batch(1340, T, seed=4), (reps, 1))[:B], device="cuda")
r = torch.as_tensor(np.tile(workload.pot_resistance_batch(1340, T), (reps, 1))[:B], device="cuda")
th = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda")
tgt, _, _ = wb.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device="cuda"), fs, r=r, want_stash=False)
for tm in (False, True):
    xin, rin = (x.t().contiguous(), r.t().contiguous()) if tm else (x, r)
    plan = engine.plan_time_parallel(B, T, float(r.max()), float(th[3]), fs, time_major=tm)
    tuned = engine.autotune_time_parallel(th, xin, tgt, fs, plan, time_major=tm, r=rin)
    for name, pl in (("planned", plan), ("sequential forward", plan._replace(k_fwd=1)), ("autotuned", tuned)):
        kf, kb, plan_used = pl.k_fwd, pl.k_bwd, pl
        st = engine.MseStep(B, T, fs, pl, "cuda", time_major=tm)
        for _ in range(3):
            st.step(th, xin, tgt, r=rin)
        e0, e1 = wb.Event(), wb.Event()
        e0.record()
        for _ in range(20):
            st.step(th, xin, tgt, r=rin)
        e1.record()
        ms = e0.elapsed_ms(e1) / 20
        s = wb.tp_status(st.status) if kf > 1 else None
        print(f"time_major={tm} {name}: fwd chunks {kf} (W={plan_used.warmup}) bwd chunks {kb}: {ms:.3f} ms/step = {B * T / ms / 1e6:.1f} G samples/s  {s}")
    # the one-pass step (wdf_clipper_step_mse_tp) on the same batch, cold and warm-started
    for k in sorted({max(1, tuned.k_fwd // 2), tuned.k_fwd, tuned.k_fwd * 2}):
        for warm in (False, True):
            st = engine.MseStep(B, T, fs, tuned._replace(k_fwd=k), "cuda", time_major=tm, warm=warm, max_warm_tiles=16)
            for _ in range(4):
                st.step_fused(th, xin, tgt, r=rin)
            e0, e1 = wb.Event(), wb.Event()
            e0.record()
            for _ in range(20):
                st.step_fused(th, xin, tgt, r=rin)
            e1.record()
            ms = e0.elapsed_ms(e1) / 20
            print(f"time_major={tm} one-pass step: chunks {k} (W={tuned.warmup}) warm={warm}: {ms:.3f} ms/step = {B * T / ms / 1e6:.1f} G samples/s  "
                  f"{wb.tp_status(st.status)} {st.warm.info() if st.warm is not None else ''}")

# the training loop itself at this shape: MSE + ESR past 50 samples in one pass, Adam on {Is, nVt, C} in the launch (the pot
# resistance is data), parameters moving every step -- what the warm start has to cope with at 99.1 kOhm
th_host = workload.clipper_theta()
for tm in (True,):
    xin, rin = (x.t().contiguous(), r.t().contiguous()) if tm else (x, r)
    for k in (8, 16, 32):
        theta = torch.tensor(th_host, dtype=torch.float32, device="cuda")
        adam = wb.Adam(4, lr=[1e-3 * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device="cuda")
        plan = engine.TpPlan(k, 448, 1e-6, 16)
        st = engine.MseStep(B, T, fs, plan, "cuda", time_major=tm, loss="mse+esr", skip=50, warm=True, max_warm_tiles=16)
        for _ in range(12):
            st.step_fused(theta, xin, tgt, r=rin, adam=adam)
        e0, e1 = wb.Event(), wb.Event()
        e0.record()
        for _ in range(20):
            st.step_fused(theta, xin, tgt, r=rin, adam=adam)
        e1.record()
        ms = e0.elapsed_ms(e1) / 20
        print(f"training loop (MSE + ESR, Adam in the launch), one-pass step, chunks {k}: {ms:.3f} ms/step = {B * T / ms / 1e6:.1f} G samples/s  "
              f"{wb.tp_status(st.status)} {st.warm.info()} loss {[float(v) for v in st.loss]}")
