"""CPU study (round 6, review item 3): what should a COLD chunk of the time-parallel clipper kernels start from?
Today: z = 0, W = 160 warm-up steps (BASELINE configs[1], 1024 x 4096 forward: 126 chunks of 32.5 owned steps behind 160).
For the bench workload (fp64 trajectory of the 1N4148 clipper, 384 sequences, every chunk start at multiples of 32) the largest
|z - z_true| at the first owned step after W exact warm-up steps from each of:
  zero       z = 0 (today)
  quasi      the quasi-static operating point of x[t0 - W] (fixed point of the step for a constant input)
  linear     the state of the diode-off recursion z' = (1 - 2p) z + 2p x (exact linear scan: cheap, 2 FMAs per step)
  lin+cheap  linear prefix, then N steps with the diode by TWO Newton iterations from the previous sample's voltage (a step that
             costs ~0.8 of the exact one: six transcendentals), then W exact steps
The verification tolerance is 1e-6; a start is usable when the largest error sits safely under it."""
import os, sys, time
import numpy as np
from scipy.special import wrightomega
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import workload

fs = workload.FS
Is, V, R, C = workload.clipper_theta()
T, B = 4096, 384
rng = np.random.default_rng(0)
idx = np.sort(rng.choice(8192, B, replace=False))
x = workload.sweep_batch(8192, T, dtype=np.float64)[idx]
G1, G2 = 1.0 / R, 2.0 * C * fs
Rp, p = 1.0 / (G1 + G2), G1 / (G1 + G2)
Lg = np.log(Rp * Is / V)


def diode(a):
    aa = np.abs(a)
    return a - 2.0 * V * np.sign(a) * (wrightomega(Lg + aa / V).real - wrightomega(Lg - aa / V).real)


def step(z, xt):
    bd = z - xt
    bt = -p * bd
    return diode(z + bt) + bt


def cheap(z, v, xt, iters=2):
    bd = z - xt
    bt = -p * bd
    a = z + bt
    for _ in range(iters):
        e = np.exp(np.clip(v / V, -30, 30))
        f = v + Rp * Is * (e - 1.0 / e) - a
        v = v - f / (1.0 + Rp * Is / V * (e + 1.0 / e))
    return 2.0 * v - a + bt, v


zt = np.zeros((T + 1, B))
zl = np.zeros((T + 1, B))
z = np.zeros(B)
l = np.zeros(B)
for t in range(T):
    z = step(z, x[:, t]); zt[t + 1] = z
    l = (1 - 2 * p) * l + 2 * p * x[:, t]; zl[t + 1] = l


def quasi(xt):
    q = np.clip(xt, -0.6, 0.6)
    for _ in range(60):
        f = step(q, xt) - q
        fp = (step(q + 1e-6, xt) - (q + 1e-6) - f) / 1e-6
        q = q - f / fp
    return q


starts = np.arange(288, T, 32)


def run(name, est, Ws):
    out = []
    for W in Ws:
        s = starts - W
        q = est(s)
        for j in range(W):
            q = step(q, x[:, s + j].T)
        err = np.abs(q - zt[starts])
        out.append(f"W {W:3d}: {err.max():.1e} ({(err > 1e-6).mean():.4f} of the chunk starts above 1e-6)")
    print(f"{name:10s} " + " | ".join(out), flush=True)


run("zero", lambda s: np.zeros((len(s), B)), [64, 96, 128, 160])
run("linear", lambda s: zl[s], [64, 96, 128])
qs = np.stack([quasi(x[:, t]) for t in range(T)])
run("quasi", lambda s: qs[s], [64, 96, 128, 160])


def lin_cheap(N):
    def est(s):
        s1 = s - N
        q = np.clip(zl[s1], -0.6, 0.6)
        v = q.copy()
        for j in range(N):
            q, v = cheap(q, v, x[:, s1 + j].T)
        return q
    return est


for N in (32, 64, 96):
    run(f"lin+cheap{N}", lin_cheap(N), [16, 32, 48, 64])
print("The state's memory in the weakly conducting regime (amplitudes 0.3-0.6 V) is ~0.906 per step whatever the estimator: an error of\n"
      "1e-2 V (what a diode-off start leaves there) needs ~93 exact -- or equally expensive approximate -- steps to fall under 1e-6.\n"
      "linear + W = 96 is marginal (8.6e-7 on this sample), linear + W = 128 safe: 32 steps saved of 192 at configs[1] -- not built.")
