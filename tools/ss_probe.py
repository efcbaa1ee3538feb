"""HPF diode clipper (HPFDiodeClipper.h:28-32: Parallel(R, Series(Vs, C)) + diode pair) through the generic state-space
kernels at the headline shape: sequential kernels vs the time-parallel ones (csrc/wdf_statespace.h), forward + reverse
sweep through the element API (tape.gradient), samples/s; and the RC lowpass of lpf.py (linear tree)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
import numpy as np, torch
import tf_wdf as wdf
from tf_wdf import tf
from wdf_hip import binding, lowering, workload

B, T, FS = int(os.environ.get("SS_B", "8192")), int(os.environ.get("SS_T", "4096")), 48000
x = torch.as_tensor(workload.sweep_batch(B, T), device="cuda")
gy = torch.randn((T, B), device="cuda") / (B * T)


def hpf(tp):
    R = wdf.Resistor(33.0e3, True); Vs = wdf.ResistiveVoltageSource(1.0e3, trainable=True); C = wdf.Capacitor(22.0e-9, FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs, C))
    dp = wdf.DiodePair(top, 4.352e-9, Vt=25.85e-3, nDiodes=1.906, trainable=True)
    return wdf.Circuit(top, dp, R, time_parallel=tp), [R.R, Vs.R, C.C, dp.Is, dp.nVt]


def lpf(tp):
    Vs = wdf.IdealVoltageSource(); R1 = wdf.Resistor(1000, True); C1 = wdf.Capacitor(1.0e-6, FS, True)
    I1 = wdf.Inverter(wdf.Series(R1, C1))
    return wdf.Circuit(I1, Vs, C1, time_parallel=tp), [C1.C, R1.R]


def timed(build, tp, reps=10):
    circ, params = build(tp)
    e = [binding.Event() for _ in range(3)]
    tf_, tb_ = [], []
    for it in range(reps + 2):
        e[0].record()
        y = circ(x)
        e[1].record()
        g = tf.GradientTape().gradient(tf.reduce_sum(y * gy), params)
        e[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf_.append(e[0].elapsed_ms(e[1])); tb_.append(e[1].elapsed_ms(e[2]))
    return float(np.median(tf_)), float(np.median(tb_)), y, [float(v) for v in g]


for name, build in (("HPF diode clipper (ns=1, diode root)", hpf), ("RC lowpass (ns=1, linear)", lpf)):
    f0, b0, y0, g0 = timed(build, None)
    f1, b1, y1, g1 = timed(build, "auto")
    circ, _ = build("auto")
    coef64, _ = circ.matrices()
    kind = binding.ROOT_DIODE_PAIR if "diode" in name else binding.ROOT_NONE
    plan = lowering.plan_ss_time_parallel(coef64, circ.ns, circ.ni, kind, B, T)
    st = None if lowering.LAST_SS_TP_STATUS["status"] is None or kind == binding.ROOT_NONE else binding.ss_tp_status(lowering.LAST_SS_TP_STATUS["status"])
    dy = float((y1 - y0).abs().max())
    dg = max(abs(a - b) / abs(b) for a, b in zip(g1, g0))
    print(f"{name}, {B} x {T}: sequential fwd {f0:.3f} + bwd {b0:.3f} ms = {B*T/(f0+b0)/1e6:.1f} G samples/s | time-parallel {plan}: "
          f"fwd {f1:.3f} + bwd {b1:.3f} ms = {B*T/(f1+b1)/1e6:.1f} G samples/s  (max |dy| {dy:.1e}, max rel dgrad {dg:.1e}, verdict {st})", flush=True)

# ---- the kernels themselves (C ABI, matrices precomputed: no host work between the events) --------------------------------
def kernel_times(build, kind, reps=10):
    circ, _ = build(None)
    coef64, r_port = circ.matrices()
    coef = coef64.detach().float().cuda()
    ns, ni = circ.ns, circ.ni
    rootp = None
    if kind == binding.ROOT_DIODE_PAIR:
        dp = circ.root
        rootp = torch.tensor([float(dp.Is), float(dp.nVt), float(r_port)], dtype=torch.float32, device="cuda")
    plan = lowering.plan_ss_time_parallel(coef64, ns, ni, kind, B, T)

    def t(fn):
        e0, e1 = binding.Event(), binding.Event()
        ts = []
        for it in range(reps + 2):
            e0.record(); out = fn(); e1.record()
            ts.append(e0.elapsed_ms(e1))
        return float(np.median(ts[2:])), out
    fs, (y, zs, _) = t(lambda: binding.ss_fwd(x, coef, ns, ni, kind, rootp))
    bs, _ = t(lambda: binding.ss_bwd(x, coef, ns, ni, zs, gy, kind, rootp))
    res = {"fwd_seq_ms": fs, "bwd_seq_ms": bs}
    if kind == binding.ROOT_DIODE_PAIR and plan is not None and plan.k_fwd >= 2:
        res["fwd_tp_ms"], _ = t(lambda: binding.ss_fwd_tp(x, coef, ns, ni, rootp, plan.k_fwd, plan.warmup, plan.tol))
        for k in (plan.k_fwd * 2, plan.k_fwd * 4):
            if T // k >= 64:
                res[f"fwd_tp_k{k}_ms"], out = t(lambda: binding.ss_fwd_tp(x, coef, ns, ni, rootp, k, plan.warmup, plan.tol))
    for k in sorted({plan.k_bwd, max(2, plan.k_bwd // 2), min(T // 64, plan.k_bwd * 2)}):
        res[f"bwd_tp_k{k}_ms"], _ = t(lambda: binding.ss_bwd_tp(x, coef, ns, ni, zs, gy, k, kind, rootp))
    return plan, res


for name, build, kind in (("HPF diode clipper", hpf, binding.ROOT_DIODE_PAIR), ("RC lowpass", lpf, binding.ROOT_NONE)):
    plan, res = kernel_times(build, kind)
    best_f = min(v for k, v in res.items() if k.startswith("fwd"))
    best_b = min(v for k, v in res.items() if k.startswith("bwd"))
    print(f"{name} kernels only, {B} x {T}, {plan}: " + ", ".join(f"{k} {v:.3f}" for k, v in res.items()) +
          f" -> best fwd + bwd {best_f + best_b:.3f} ms = {B*T/(best_f+best_b)/1e6:.1f} G samples/s "
          f"(sequential pair {B*T/(res['fwd_seq_ms']+res['bwd_seq_ms'])/1e6:.1f})", flush=True)


# ---- a training loop on the HPF clipper (one Adam per component): the recurrence kernels alone, chunks started warm ------
def training_loop(steps=80, lr_rel=1.0e-3):
    circ, params = hpf("auto")
    ref, _ = hpf(None)
    tgt = (ref(x) * 0.8).as_subclass(torch.Tensor).detach()
    opts = [tf.keras.optimizers.Adam(learning_rate=lr_rel * float(p)) for p in params]
    ev = [binding.Event() for _ in range(4)]
    tf_, tb_, used = [], [], []
    for it in range(steps):
        binding.Event.bracket_next(ev[0], ev[1])            # the forward's recurrence kernel
        with tf.GradientTape() as tape:
            y = circ(x)
            loss = tf.reduce_mean(tf.square(y - tgt))
        binding.Event.bracket_next(ev[2], ev[3])            # the reverse sweep's
        grads = tape.gradient(loss, params)
        for o, g, p in zip(opts, grads, params):
            o.apply_gradients([(g, p)])
        torch.cuda.synchronize()
        tf_.append(ev[0].elapsed_ms(ev[1])); tb_.append(ev[2].elapsed_ms(ev[3]))
        st = binding.ss_tp_status(lowering.LAST_SS_TP_STATUS["status"])
        used.append((lowering.LAST_SS_TP_STATUS["chunks_used"], lowering.LAST_SS_TP_STATUS["warmup_used"], st["gated_waves"], st["max_miss"]))
    f, b = float(np.median(tf_[steps // 2:])), float(np.median(tb_[steps // 2:]))
    print(f"HPF diode clipper, training loop ({steps} Adam steps of {lr_rel:g} relative, chunks started from the previous calls' states): "
          f"cold call fwd kernel {tf_[0]:.3f} ms; later calls fwd {f:.3f} + bwd {b:.3f} ms = {B*T/(f+b)/1e6:.1f} G samples/s (recurrence kernels)")
    print("  per call chunks/warm-up/gated waves/max miss: " + "  ".join(f"{k}/{w}/{g}/{m:.0e}" for k, w, g, m in used[::4]))


if os.environ.get("SS_LOOP", "1") != "0":
    training_loop()
