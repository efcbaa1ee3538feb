#!/usr/bin/env python3
"""tools/ss_step_bench.py -- lpf.py's training loop shape (GradientTape -> circ.mse -> tape.gradient -> two Adam optimizers,
lpf.py:77-99) on the RC lowpass at 8192 x 4096 through the element API: plain path (host probe, forward + reverse-sweep
kernels, torch autograd) against the resident one-pass step (Circuit.to_device)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "differentiable-wdfs_amd", "lib"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import tf_wdf as wdf  # noqa: E402
from tf_wdf import tf  # noqa: E402
from wdf_hip import binding  # noqa: E402

FS = 48000
B, T = int(os.environ.get("B", 8192)), int(os.environ.get("T", 4096))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
x = torch.randn((B, T), device="cuda")
tgt = torch.randn((T, B), device="cuda") * 0.3
for resident in (() if os.environ.get("ONLY") == "hpf" else (True,) if os.environ.get("ONLY_RESIDENT") else (False, True)):
    Vs = wdf.IdealVoltageSource()
    R1, C1 = wdf.Resistor(1000, True), wdf.Capacitor(1.0e-6, FS, True)
    circ = wdf.Circuit(wdf.Inverter(wdf.Series(R1, C1)), Vs, C1)
    if resident:
        circ.to_device()
    oR, oC = tf.keras.optimizers.Adam(learning_rate=25.0), tf.keras.optimizers.Adam(learning_rate=1.0e-8)

    def step():
        with tf.GradientTape() as tape:
            loss = circ.mse(x, tgt)
        gC, gR = tape.gradient(loss, [C1.C, R1.R])
        oC.apply_gradients([(gC, C1.C)])
        oR.apply_gradients([(gR, R1.R)])
        return loss

    for _ in range(10):
        step()
    for rep in range(int(os.environ.get("REPS", "2"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep + 1 < int(os.environ.get("REPS", "2")):
            print(f"  (pass {rep}: {dt / steps * 1e3:.4f} ms per step, host side {t_host / steps * 1e3:.4f} ms)")
    e0, e1 = binding.Event(), binding.Event()
    binding.Event.bracket_next(e0, e1)
    step()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_ms(e1) if resident else float("nan")
    print(f"{'resident one-pass step' if resident else 'plain path            '}: {dt / steps * 1e3:.4f} ms per step = {B * T / (dt / steps) / 1e9:.1f} G samples/s "
          f"(host side {t_host / steps * 1e3:.4f} ms; step kernel {k_ms:.4f} ms = {16.0 * B * T / (k_ms * 1e-3) / 1e12 if resident else float('nan'):.2f} TB/s of its 16 B/sample); "
          f"loss {float(loss):.5e}, R {float(R1.R):.2f}, C {float(C1.C):.4e}")


# ---- the HPF diode clipper (HPFDiodeClipper.h:28-32) in the same loop shape: host probe against the device probe -------
x2 = torch.randn((B, T), device="cuda") * 1.2


def hpf(r=33.0e3, rs=1.0e3, c=22.0e-9, i_s=4.352e-9, nd=1.906):
    R = wdf.Resistor(r, True); Vs2 = wdf.ResistiveVoltageSource(rs, trainable=True); C = wdf.Capacitor(c, FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs2, C))
    dp = wdf.DiodePair(top, i_s, Vt=25.85e-3, nDiodes=nd, trainable=True)
    return wdf.Circuit(top, dp, R), [R.R, Vs2.R, C.C, dp.Is, dp.nVt]


# the target: the same circuit with other component values (a teacher the loop converges to -- against random noise the
# optimizers drive C into its lower clamp within a few hundred steps and the circuit degenerates)
with torch.no_grad():
    tgt2 = hpf(39.0e3, 1.5e3, 15.0e-9, 2.52e-9, 1.752)[0](x2).as_subclass(torch.Tensor).detach().clone()
for resident in ((True,) if os.environ.get("ONLY_RESIDENT") else (False, True)):
    circ, params = hpf()
    if resident:
        circ.to_device()
    opts = [tf.keras.optimizers.Adam(learning_rate=1.0e-3 * float(p)) for p in params]

    def step2():
        with tf.GradientTape() as tape:
            loss = circ.mse(x2, tgt2)
        grads = tape.gradient(loss, params)
        for o, g, p in zip(opts, grads, params):
            o.apply_gradients([(g, p)])
        return loss

    for _ in range(10):
        step2()
    for rep in range(int(os.environ.get("REPS", "2"))):          # (the first loop of a process also pays the host's warm-up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step2()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep + 1 < int(os.environ.get("REPS", "2")):
            print(f"  (pass {rep}: {dt / steps * 1e3:.4f} ms per step, host side {t_host / steps * 1e3:.4f} ms)")
        if resident and os.environ.get("VERBOSE"):
            c = circ._tree.read_ctl(next(iter(circ._tree.cache.values())))
            print(f"    params {[float(p) for p in params]}; cold warm-up estimate {circ._tree.cold_warmup()}; w_used {c['w_used']}, "
                  f"max_miss {c['max_miss']:.2e}, repaired so far {c['total_gated']}; loss {float(loss):.4e}")
    if resident:
        print(f"  control block: {circ._tree.read_ctl(next(iter(circ._tree.cache.values())))}")
    print(f"HPF clipper, {'resident (device probe)' if resident else 'plain path (host probe) '}: {dt / steps * 1e3:.4f} ms per step = "
          f"{B * T / (dt / steps) / 1e9:.1f} G samples/s (host side {t_host / steps * 1e3:.4f} ms); loss {float(loss):.5e}")
