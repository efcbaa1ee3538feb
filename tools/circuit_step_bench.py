"""GPU: the element API (tf_wdf.Circuit fast tier) on the headline shape: forward + torch MSE + tape.gradient."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
import tf_wdf as wdf
from wdf_hip import workload, binding as wb
tf = wdf.tf
B, T, FS = 8192, 4096, 48000
th = workload.clipper_theta()
x = torch.as_tensor(workload.sweep_batch(B, T), device="cuda")
Vs = wdf.ResistiveVoltageSource(float(th[2]), trainable=True); Cap = wdf.Capacitor(float(th[3]), FS, trainable=True)
P1 = wdf.Parallel(Vs, Cap); dp = wdf.DiodePair(P1, float(th[0]), Vt=float(th[1]), trainable=True)
circ = wdf.Circuit(P1, dp, Cap)
tgt = circ(x).detach() * 0.9
def step():
    with tf.GradientTape() as tape:
        y = circ(x); loss = tf.reduce_mean(tf.square(y - tgt))
    return tape.gradient(loss, [dp.Is, dp.nVt, Vs.R, Cap.C])
step(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): g = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f"Circuit fast tier, autograd MSE step: {dt*1e3:.2f} ms = {B*T/dt/1e9:.1f} G samples/s")

def step_fused():
    with tf.GradientTape() as tape:
        loss = circ.mse(x, tgt)
    return tape.gradient(loss, [dp.Is, dp.nVt, Vs.R, Cap.C])
g2 = step_fused(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): g2 = step_fused()
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f"Circuit.mse (loss inside the sweep): {dt*1e3:.2f} ms = {B*T/dt/1e9:.1f} G samples/s;  gradients vs the plain path:",
      [abs(float(a) - float(b)) / abs(float(b)) for a, b in zip(g2, g)])
