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
for resident in ((True,) if os.environ.get("ONLY_RESIDENT") else (False, True)):
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
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
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
for resident in ((True,) if os.environ.get("ONLY_RESIDENT") else (False, True)):
    R = wdf.Resistor(33.0e3, True); Vs2 = wdf.ResistiveVoltageSource(1.0e3, trainable=True); C = wdf.Capacitor(22.0e-9, FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs2, C))
    dp = wdf.DiodePair(top, 4.352e-9, Vt=25.85e-3, nDiodes=1.906, trainable=True)
    circ = wdf.Circuit(top, dp, R)
    params = [R.R, Vs2.R, C.C, dp.Is, dp.nVt]
    if resident:
        circ.to_device()
    opts = [tf.keras.optimizers.Adam(learning_rate=1.0e-3 * float(p)) for p in params]

    def step2():
        with tf.GradientTape() as tape:
            loss = circ.mse(x2, tgt)
        grads = tape.gradient(loss, params)
        for o, g, p in zip(opts, grads, params):
            o.apply_gradients([(g, p)])
        return loss

    for _ in range(10):
        step2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step2()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"HPF clipper, {'resident (device probe)' if resident else 'plain path (host probe) '}: {dt / steps * 1e3:.4f} ms per step = "
          f"{B * T / (dt / steps) / 1e9:.1f} G samples/s (host side {t_host / steps * 1e3:.4f} ms); loss {float(loss):.5e}")
