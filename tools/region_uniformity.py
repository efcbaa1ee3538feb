"""CPU: how often a wave of the one-pass step (128 neighbouring sequences: 64 lanes x 2 packed) has the argument of its
general Wright-omega evaluation (u0 = |a| / nVt + log(Rp Is / nVt), csrc/wdf_omega.h diode_pair) in ONE of the three start
regions (x <= -2 | -2 < x <= 1 + pi | above: toms917.cpp:240-296) at a time step -- what a ballot-gated single-region start
would need (round-4 review, item 9).  numpy restatement of the clipper recursion in fp64 over the bench's own batch."""
import os, sys, json
import numpy as np
from scipy.special import wrightomega
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
from wdf_hip import workload

FS = 48000.0
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
Is, nVt, R, C = workload.clipper_theta()
x = workload.sweep_batch(8192, T, b0=0, b1=B).astype(np.float64)
Rc = 1.0 / (2.0 * C * FS)
G1, G2 = 1.0 / R, 1.0 / Rc
p1 = G1 / (G1 + G2)
Rp = 1.0 / (G1 + G2)
L = np.log(Rp * Is / nVt)
z = np.zeros(B)
reg = np.empty((T, B), dtype=np.int8)
for t in range(T):
    a = p1 * x[:, t] + (1.0 - p1) * z                           # the wave up to the root
    s = np.abs(a) / nVt
    u0 = L + s
    reg[t] = (u0 > -2.0).astype(np.int8) + (u0 > 1.0 + np.pi)
    b = a - 2.0 * nVt * np.sign(a) * (wrightomega(u0).real - wrightomega(L - s).real)
    z = b - p1 * (z - x[:, t])                                  # the capacitor's next incident wave
out = {"B": B, "T": T, "share_of_samples": [float(np.mean(reg == k)) for k in range(3)]}
for w in (128, 64, 16, 2):
    g = reg[:, : B // w * w].reshape(T, B // w, w)
    out["uniform_%d" % w] = float(np.mean(g.min(axis=2) == g.max(axis=2)))
print(json.dumps(out))
