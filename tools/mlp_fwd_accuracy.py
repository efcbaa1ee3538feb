"""GPU box: the two MLP-root forward kernels (row / matrix cores) against an fp64 evaluation of the same recursion
(torch, on the device): which one is closer, and how far apart rounding alone puts two fp32 evaluations."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "differentiable-wdfs_amd", "lib"))
import numpy as np, torch
from wdf_hip import binding as wb, workload

FS = 48000.0
B, T = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.tensor(workload.sweep_batch(B, T, seed=5) * 0.5, device="cuda", dtype=torch.float32)
r = torch.tensor(workload.dataset_resistance_batch(B, T), device="cuda", dtype=torch.float32)
th2 = torch.tensor([45.0e3, 4.7e-9], device="cuda")


def fp64_forward(x, r, wh, H, n_tanh):
    w = torch.tensor(wh, device="cuda", dtype=torch.float64)
    k0, b0 = w[:2 * H].view(2, H), w[2 * H:3 * H]
    mids, o = [], 3 * H
    for _ in range(n_tanh - 1):
        mids.append((w[o:o + H * H].view(H, H), w[o + H * H:o + H * H + H])); o += H * H + H
    wo, bo = w[o:o + H], w[o + H]
    G2 = 2.0 * 4.7e-9 * FS
    z = torch.zeros(B, device="cuda", dtype=torch.float64)
    xd, rd = x.double(), r.double()
    ys = []
    for t in range(T):
        G1 = 1.0 / rd[:, t]; Rp = 1.0 / (G1 + G2); p = G1 * Rp; lr = torch.log(Rp)
        bt = -p * (z - xd[:, t]); a = z + bt
        h = torch.tanh(torch.stack([a, lr], 1) @ k0 + b0)
        for K, bb in mids: h = torch.tanh(h @ K + bb)
        zn = bt - (h @ wo + bo)
        ys.append(0.5 * (zn + z)); z = zn
    return torch.stack(ys)


for net in ("2x16", "2x8", "4x8"):
    wh, H, n_tanh = workload.reference_mlp_weights(net)
    w = torch.tensor(wh, device="cuda")
    ref = fp64_forward(x, r, wh, H, n_tanh)
    out = {}
    for mode in ("1", "0"):
        os.environ["WDF_MLP_FWD_ROW"] = mode
        out[mode] = wb.clipper_mlp_fwd(x, th2, w, H, n_tanh, FS, r=r)[0].double()
    print(f"{net} T={T}: |row - fp64| = {float((out['1'] - ref).abs().max()):.2e}   |matrix cores - fp64| = {float((out['0'] - ref).abs().max()):.2e}"
          f"   |row - matrix cores| = {float((out['1'] - out['0']).abs().max()):.2e}   max |y| = {float(ref.abs().max()):.2f}")
