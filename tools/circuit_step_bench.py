"""GPU: the element API (tf_wdf.Circuit fast tier) on the headline shape, per training step of the reference's loop shape
(lpf.py:86-99: GradientTape -> loss -> tape.gradient -> Adam.apply_gradients):
  plain     circ(x) + torch MSE, host Variables
  fused     circ.mse(x, target), host Variables (gradient copied back to the host every step)
  resident  circ.to_device(); circ.mse(x, target): Variables in one device block, one-launch Adam, no host round trip
PROFILE=1 adds a cProfile of the resident loop's host side."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
import tf_wdf as wdf
from wdf_hip import workload, binding as wb
tf = wdf.tf
B, T, FS = 8192, 4096, 48000
th = workload.clipper_theta()
x = torch.as_tensor(workload.sweep_batch(B, T), device="cuda")


def build():
    Vs = wdf.ResistiveVoltageSource(float(th[2]), trainable=True); Cap = wdf.Capacitor(float(th[3]), FS, trainable=True)
    P1 = wdf.Parallel(Vs, Cap); dp = wdf.DiodePair(P1, float(th[0]), Vt=float(th[1]), trainable=True)
    return wdf.Circuit(P1, dp, Cap), [dp.Is, dp.nVt, Vs.R, Cap.C]


circ, vs = build()
tgt = wb.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device="cuda"), FS, want_stash=False)[0]


def loop(circ, vs, loss_fn, n):
    opts = [tf.keras.optimizers.Adam(learning_rate=1.0e-3 * float(t)) for t in th]
    def step():
        with tf.GradientTape() as tape:
            loss = loss_fn(circ)
        grads = tape.gradient(loss, vs)
        for o, g, v in zip(opts, grads, vs):
            o.apply_gradients([(g, v)])
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): loss = step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    return dt, float(loss), step


def loop_one_opt(circ, vs, n):
    opt = tf.keras.optimizers.Adam(learning_rate=1.0e-13)
    def step():
        with tf.GradientTape() as tape:
            loss = circ.mse(x, tgt)
        grads = tape.gradient(loss, vs)
        opt.apply_gradients(zip(grads, vs))
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): loss = step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    return dt, float(loss), step


dt, l, _ = loop(circ, vs, lambda c: tf.reduce_mean(tf.square(c(x) - tgt)), 10)
print(f"plain    (circ(x) + torch MSE, host Variables):        {dt*1e3:.3f} ms/step = {B*T/dt/1e9:6.1f} G samples/s  loss {l:.3e}")
circ, vs = build()
dt, l, _ = loop(circ, vs, lambda c: c.mse(x, tgt), 50)
print(f"fused    (circ.mse, host Variables):                   {dt*1e3:.3f} ms/step = {B*T/dt/1e9:6.1f} G samples/s  loss {l:.3e}")
circ, vs = build(); circ.to_device()
dt, l, step = loop(circ, vs, lambda c: c.mse(x, tgt), 300)
print(f"resident (circ.to_device(); circ.mse; Adam per variable): {dt*1e3:.3f} ms/step = {B*T/dt/1e9:6.1f} G samples/s  loss {l:.3e}")
circ, vs = build(); circ.to_device()
dt, l, step1 = loop_one_opt(circ, vs, 300)
print(f"resident (one Adam for the four Variables):             {dt*1e3:.3f} ms/step = {B*T/dt/1e9:6.1f} G samples/s  loss {l:.3e}")
# host time alone: the same loop with the GPU left to drain afterwards
torch.cuda.synchronize(); t0 = time.time()
for _ in range(300): step1()
th_host = (time.time() - t0) / 300
torch.cuda.synchronize()
print(f"  host side of that loop: {th_host*1e3:.3f} ms/step (the kernels of a step take ~0.11 ms)")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): step1()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
