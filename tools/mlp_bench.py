"""GPU: timing of the MLP-root clipper kernels at the reference's training-set shape
(1340 sequences x 2048 samples, clipper_pot.py:58 / SURVEY 8d C4)."""
import os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, workload

B, T, fs = 1340, 2048, workload.FS
x = torch.as_tensor(workload.sweep_batch(B, T, seed=4), device="cuda")
r = torch.as_tensor(workload.pot_resistance_batch(B, T), device="cuda")
th2 = torch.tensor([45.0e3, 4.7e-9], dtype=torch.float32, device="cuda")


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = wb.Event(), wb.Event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    return e0.elapsed_ms(e1) / n


for hidden, n_tanh in ((4, 3), (8, 3), (16, 3), (8, 5)):
    nw = wb.lib().wdf_mlp_weight_count(hidden, n_tanh)
    w = (torch.randn(nw, device="cuda") * 0.3).contiguous()
    y, zs, _ = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, fs, r=r)
    gy = torch.randn_like(y) / y.numel()
    tf_ = timeit(lambda: wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, fs, r=r))
    tb_ = timeit(lambda: wb.clipper_mlp_bwd(x, th2, w, hidden, n_tanh, fs, zs, gy, r=r))
    print(f"{n_tanh - 1}x{hidden}: fwd {tf_:.3f} ms  bwd(kernel) {tb_:.3f} ms   -> {B * T / (tf_ + tb_) / 1e3:.1f} M samples/s (kernels only)")

# whole training step (forward + autograd backward incl. the dense weight-gradient pass):
# sequential kernels vs the data-level segmented time-parallel path, static R and per-sample R
from wdf_hip import mlp_root
for R_kind in ("static 45k", "per-sample pot"):
    rr = r if R_kind != "static 45k" else None
    for hidden, n_tanh in ((8, 3), (16, 3)):
        nw = wb.lib().wdf_mlp_weight_count(hidden, n_tanh)
        w = (torch.randn(nw, device="cuda") * 0.3).requires_grad_(True)
        th = th2.clone().requires_grad_(True)
        gy = torch.randn(T, B, device="cuda") / (T * B)

        def step(tp):
            y, _ = mlp_root.clipper_mlp(th, w, x, rr, None, fs, hidden, n_tanh, 4.7e-9, R_static=45.0e3, time_parallel=tp)
            torch.autograd.grad((y * gy).sum(), [th, w])

        t_seq, t_tp = timeit(lambda: step(None), 3), timeit(lambda: step("auto"), 3)
        plan = mlp_root.segment_plan(B, T, mlp_root.engine.resistance_max(rr) if rr is not None else 45.0e3, 4.7e-9, fs)
        print(f"step {n_tanh - 1}x{hidden} R {R_kind}: sequential {t_seq:.2f} ms, segmented {t_tp:.2f} ms "
              f"(plan K,L,W={plan}, miss={mlp_root.LAST_SEGMENT_MISS['miss']})")

# forward kernel A/B: 16-lane row per sequence (default) vs one lane per sequence
print("forward kernel, per-sample R, 1340 x 2048:")
for hidden, n_tanh in ((4, 3), (8, 3), (16, 3), (8, 5)):
    nw = wb.lib().wdf_mlp_weight_count(hidden, n_tanh)
    w = (torch.randn(nw, device="cuda") * 0.3).contiguous()
    res = {}
    for lane in (False, True):
        wb.MLP_LANE_PER_SEQUENCE = lane
        y, _, _ = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, fs, r=r)
        res[lane] = (timeit(lambda: wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, fs, r=r)), y)
    wb.MLP_LANE_PER_SEQUENCE = False
    print(f"  {n_tanh - 1}x{hidden}: row {res[False][0]:.3f} ms, lane {res[True][0]:.3f} ms, max |dy| {float((res[False][1] - res[True][1]).abs().max()):.2e}")

print("reverse-sweep kernel (+ theta reduce), per-sample R, 1340 x 2048:")
for hidden, n_tanh in ((4, 3), (8, 3), (16, 3), (8, 5)):
    nw = wb.lib().wdf_mlp_weight_count(hidden, n_tanh)
    w = (torch.randn(nw, device="cuda") * 0.3).contiguous()
    y, zs, _ = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, fs, r=r)
    gy = torch.randn_like(y) / y.numel()
    res = {}
    for lane in (False, True):
        wb.MLP_LANE_PER_SEQUENCE = lane
        out = wb.clipper_mlp_bwd(x, th2, w, hidden, n_tanh, fs, zs, gy, r=r)
        res[lane] = (timeit(lambda: wb.clipper_mlp_bwd(x, th2, w, hidden, n_tanh, fs, zs, gy, r=r)), out)
    wb.MLP_LANE_PER_SEQUENCE = False
    a, b = res[False][1], res[True][1]
    print(f"  {n_tanh - 1}x{hidden}: row {res[False][0]:.3f} ms, lane {res[True][0]:.3f} ms, "
          f"gtheta rel diff {float(((a[0] - b[0]).abs() / b[0].abs().clamp_min(1e-30)).max()):.1e}, "
          f"max |d gb| {float((a[1] - b[1]).abs().max()):.1e} of {float(b[1].abs().max()):.1e}")

# in-kernel time-parallel kernels (csrc/wdf_mlp_tp.h): forward (per-wave warm-up, verified) and the exact
# all-steps-parallel reverse sweep, over chunk counts
# (pot values in contiguous blocks as the reference's loader leaves them, and the reference's trained weights:
# random small weights make the circuit forget its state in ~10 steps and flatter every warm-up)
print("time-parallel kernels, dataset-layout pot channel, reference weights, 1340 x 2048 (whole call: kernels + helpers):")
r = torch.as_tensor(workload.dataset_resistance_batch(B, T), device="cuda")
wrow, wmax = mlp_root.warmup_per_wave(r, 4.7e-9, fs)
print(f"  warm-up per wave: min {int(wrow.min())} mean {float(wrow.float().mean()):.0f} max {wmax}")
for name in ("2x8", "2x16", "4x8"):
    wh, hidden, n_tanh = workload.reference_mlp_weights(name)
    w = torch.as_tensor(wh, device="cuda")
    y, zs, _ = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, fs, r=r)
    gy = torch.randn_like(y) / y.numel()
    t_seq_f = timeit(lambda: wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, fs, r=r))
    t_seq_b = timeit(lambda: wb.clipper_mlp_bwd_w(x, th2, w, hidden, n_tanh, fs, zs, gy, r=r))
    print(f"  {n_tanh - 1}x{hidden}: sequential fwd {t_seq_f:.3f} ms, bwd_w {t_seq_b:.3f} ms")
    for K in (2, 4, 6, 8, 12, 16):
        st = torch.zeros(4, dtype=torch.int32, device="cuda")
        tf_ = timeit(lambda: wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, fs, K, wmax, r=r, warmup_per_wave=wrow, status=st))
        tu_ = timeit(lambda: wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, fs, K, wmax, r=r, status=st))
        s = wb.mlp_tp_status(st)
        tb_ = timeit(lambda: wb.clipper_mlp_bwd_w_tp(x, th2, w, hidden, n_tanh, fs, zs, gy, K, r=r))
        tb2 = timeit(lambda: wb.clipper_mlp_bwd_w_tp(x, th2, w, hidden, n_tanh, fs, zs, gy, 2 * K, r=r))
        print(f"    K={K:2d}: fwd_tp {tf_:.3f} ms (uniform W: {tu_:.3f}; status {s})  bwd_w_tp K {tb_:.3f} / 2K {tb2:.3f} ms")
