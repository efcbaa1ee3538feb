"""CPU (numpy, fp64): feasibility of a parallel-in-time NEWTON solve for the MLP-root clipper forward
(clipper_pot.py:113-124), the alternative to the warm-up-and-verify chunks of csrc/wdf_mlp_step.h.

One Newton iteration on the whole trajectory: (i) evaluate the network and kappa = dz'/dz at ALL B T points of the current
trajectory (independent: throughput-bound), (ii) solve the affine recurrence z[n+1] = kappa_n z[n] + c_n exactly (a scan).
The question this answers: starting from the PREVIOUS epoch's trajectory, how many iterations until the trajectory is
within the forward's 4e-6, for weight changes of the size the reference's optimizer makes (Adam(1e-4, beta_1 0.5): every
weight moves by ~1e-4 per step; the trajectory moves by 1e-2 ... 4e-1 V per call, profiles/r03_mlp_start_probe.txt).

usage: python tools/newton_study.py [net, default 2x16_pre]     (no GPU; reads the package's weight file)
Result (profiles/r05_newton_study.txt): 2 iterations from a 6e-3 V move, 3 from 2e-2, 4-5 from 9e-2, 6-7 from 0.2, 6 from 0.4 --
the diode knee bends on the scale of nVt = 50 mV, so moves of 0.1 V and more start outside Newton's quadratic basin.
Priced: one evaluation pass is the forward's owned work at its throughput-bound rate -- 172 k wave-steps (84 columns x 2048) of
18 MFMA + 149 VALU + 27 transcendental instructions; two waves per SIMD buy 0-14 % over the one-wave 0.68 us per step
(DESIGN section 4), i.e. >= 0.6 us per wave-step per SIMD -> >= 100 us per iteration on 1024 SIMDs, before the scan.  Three to seven
iterations = 0.3 - 0.7 ms against the 0.217 ms the chunked forward takes now: discarded."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import workload
FS, C = 48000.0, 4.7e-9

def unpack(w, hidden, n_layers):
    Ws, o, n_in = [], 0, 2
    for l in range(n_layers + 1):
        n_out = hidden if l < n_layers else 1
        K = w[o:o + n_in * n_out].reshape(n_in, n_out); o += n_in * n_out
        b = w[o:o + n_out]; o += n_out
        Ws.append((K, b)); n_in = n_out
    return Ws

def net(Ws, a, lr):
    """f = -NN(a, lr), df/da, elementwise over arrays a, lr."""
    h = np.stack([a, lr], -1)
    dh = np.zeros_like(h); dh[..., 0] = 1.0
    for i, (K, b) in enumerate(Ws):
        pre, dpre = h @ K + b, dh @ K
        if i < len(Ws) - 1:
            h = np.tanh(pre); dh = (1 - h * h) * dpre
        else:
            h, dh = pre, dpre
    return -h[..., 0], -dh[..., 0]

def seq_forward(Ws, x, p, lr):
    B, T = x.shape
    z = np.zeros((T + 1, B))
    for n in range(T):
        a = (1 - p[:, n]) * z[n] + p[:, n] * x[:, n]
        f, _ = net(Ws, a, lr[:, n])
        z[n + 1] = f - p[:, n] * (z[n] - x[:, n])
    return z

def newton(Ws, x, p, lr, z, zref, iters=8):
    B, T = x.shape
    errs = [np.abs(z - zref).max()]
    for it in range(iters):
        zc = z[:-1].T                                  # [B,T]
        a = (1 - p) * zc + p * x
        f, df = net(Ws, a, lr)
        F = f - p * (zc - x)
        kap = df * (1 - p) - p
        c = F - kap * zc
        znew = np.zeros_like(z)
        for n in range(T):                             # the exact affine scan
            znew[n + 1] = kap[:, n] * znew[n] + c[:, n]
        z = znew
        errs.append(np.abs(z - zref).max())
    return errs

net_name = sys.argv[1] if len(sys.argv) > 1 else "2x16_pre"
wh, hidden, n_layers = workload.reference_mlp_weights(net_name)
w0 = wh.astype(np.float64)
B, T = 64, 2048
idx = np.linspace(0, 1339, B).astype(int)
x = (workload.sweep_batch(1340, T, seed=4) * 0.6)[idx].astype(np.float64)
r = workload.dataset_resistance_batch(1340, T)[idx].astype(np.float64)
Rc = 1.0 / (2 * C * FS)
p = Rc / (r + Rc)                                      # p1R = G1/G with G1 = 1/R, G2 = 1/Rc
lr = np.log(r * Rc / (r + Rc))
rng = np.random.default_rng(0)
z0 = seq_forward(unpack(w0, hidden, n_layers), x, p, lr)
print("trajectory range", z0.min(), z0.max())
for scale in (3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3):
    w1 = w0 + scale * np.sign(rng.standard_normal(w0.size))      # an Adam step moves EVERY weight by ~lr (1e-4); swings accumulate over ~16 calls
    Ws1 = unpack(w1, hidden, n_layers)
    z1 = seq_forward(Ws1, x, p, lr)
    errs = newton(Ws1, x, p, lr, z0.copy(), z1)
    print(f"|dw|={scale:.0e}  trajectory moved {np.abs(z1 - z0).max():.2e}  newton errors:", " ".join(f"{e:.1e}" for e in errs))
