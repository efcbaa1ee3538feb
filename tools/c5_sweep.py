"""BASELINE config 5: asymmetric (two-different-diode) clipper, fp64 Newton tolerance sweep vs the
fp32 Wright-omega closed form -- error against the oracle's exact solve and samples/s on one
MI355X.  Prints one JSON line per row; copy the output under profiles/."""
import json
import sys
import numpy as np
import torch
import os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib")); sys.path.insert(0, os.path.join(_R, "oracle"))
from wdf_hip import binding as wb, workload
import oracle as O

FS = workload.FS
THETA6 = np.array([4.352e-9, 25.85e-3 * 1.906, 2.0e-6, 25.85e-3 * 1.4, 45.0e3, 4.7e-9])
B, T = 8192, 4096
x = workload.sweep_batch(B, T)
xd = torch.as_tensor(x, device="cuda")
th = torch.tensor(THETA6, dtype=torch.float32, device="cuda")
t32 = THETA6.astype(np.float32).astype(np.float64)
pick = np.random.default_rng(0).choice(B, 32, replace=False)
ref = O.clipper_asym_fwd(t32, FS, x[pick].astype(np.float64))
pk = torch.as_tensor(pick, device="cuda")


def run(mode, tol, max_iter):
    wb.clipper_asym_fwd(xd, th, FS, mode, tol=tol, max_iter=max_iter)
    torch.cuda.synchronize()
    e0, e1 = wb.Event(), wb.Event()
    e0.record()
    n = 3
    for _ in range(n):
        y, _, it = wb.clipper_asym_fwd(xd, th, FS, mode, tol=tol, max_iter=max_iter, want_iters=True)
    e1.record()
    ms = e0.elapsed_ms(e1) / n
    err = float(np.max(np.abs(y[:, pk].cpu().numpy() - ref)))
    iters = float(it.sum()) / (it.numel() * T)
    return ms, err, iters


rows = []
ms, err, _ = run(wb.ASYM_OMEGA_F32, 1e-12, 1)
rows.append({"root": "fp32 Wright-omega closed form (1 FSC step)", "ms": ms, "samples_per_s": B * T / ms * 1e3, "max_abs_err_vs_exact": err})
for tol in (1e-4, 1e-6, 1e-8, 1e-10, 1e-12, 1e-14):
    ms, err, iters = run(wb.ASYM_NEWTON_F64, tol, 50)
    rows.append({"root": f"fp64 Newton tol={tol:g}", "ms": ms, "samples_per_s": B * T / ms * 1e3, "max_abs_err_vs_exact": err,
                 "mean_newton_iters_per_wave_step": iters})


def run_tp(mode, tol, max_iter, K, W=192):
    """the same forward cut into K time chunks (wdf_clipper_asym_fwd_tp: verified on the device)"""
    wb.clipper_asym_fwd_tp(xd, th, FS, mode, K, W, tol=tol, max_iter=max_iter)
    torch.cuda.synchronize()
    e0, e1 = wb.Event(), wb.Event()
    e0.record()
    n = 5
    for _ in range(n):
        y, _, _, st = wb.clipper_asym_fwd_tp(xd, th, FS, mode, K, W, tol=tol, max_iter=max_iter)
    e1.record()
    ms = e0.elapsed_ms(e1) / n
    return ms, float(np.max(np.abs(y[:, pk].cpu().numpy() - ref))), wb.mlp_tp_status(st)


for K in (4, 8, 16, 32):
    ms, err, st = run_tp(wb.ASYM_OMEGA_F32, 1e-12, 1, K)
    rows.append({"root": f"fp32 Wright-omega closed form, {K} time chunks", "ms": ms, "samples_per_s": B * T / ms * 1e3,
                 "max_abs_err_vs_exact": err, "verify": st})
    for tol in (1e-6, 1e-12):
        ms, err, st = run_tp(wb.ASYM_NEWTON_F64, tol, 50, K)
        rows.append({"root": f"fp64 Newton tol={tol:g}, {K} time chunks", "ms": ms, "samples_per_s": B * T / ms * 1e3,
                     "max_abs_err_vs_exact": err, "verify": st})
for r in rows:
    print(json.dumps(r))
