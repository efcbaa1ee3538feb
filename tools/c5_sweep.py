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


# ---- forward + reverse sweep (gradients to all six parameters of L = mean((y - y*)^2)) --------------------------------------
# error columns: the gradient on 32 picked sequences against fp64 central differences of the oracle's exact forward (Newton
# mode: the model being differentiated; OMEGA mode differentiates its own closed form, so its distance to the exact model's
# gradient is MODEL error) and, Newton mode, the time-parallel sweep against the sequential one that re-solves every root.
th_star = torch.tensor(THETA6 * np.array([1.2, 0.95, 0.8, 1.05, 0.9, 1.1]), dtype=torch.float32, device="cuda")
tgt, _, _ = wb.clipper_asym_fwd(xd, th_star, FS, wb.ASYM_NEWTON_F64, tol=1e-12)
tgt_pick = tgt[:, pk].cpu().numpy().astype(np.float64)
x_pick = x[pick].astype(np.float64)


def fd_grad():
    g = np.zeros(6)
    for i in range(6):
        h = 1e-6 * t32[i]
        tp_, tm_ = t32.copy(), t32.copy()
        tp_[i] += h
        tm_[i] -= h
        lp = np.mean((O.clipper_asym_fwd(tp_, FS, x_pick) - tgt_pick) ** 2)
        lm = np.mean((O.clipper_asym_fwd(tm_, FS, x_pick) - tgt_pick) ** 2)
        g[i] = (lp - lm) / (2 * h)
    return g


g_fd = fd_grad()
xp_d = xd[pk].contiguous()


def picked_grad(mode, kb):
    y, zT, _, zs = wb.clipper_asym_fwd(xp_d, th, FS, mode, tol=1e-12, want_zT=True, want_stash=True)
    gy = (2.0 * (y - tgt[:, pk]) / y.numel()).contiguous()
    if kb == 0:
        return wb.clipper_asym_bwd(xp_d, th, FS, zs, gy).cpu().numpy().astype(np.float64)
    return wb.clipper_asym_bwd_tp(xp_d, th, FS, mode, zs, zT, gy, kb).cpu().numpy().astype(np.float64)


def run_fwd_bwd(mode, kf, kb, tol=1e-12, W=192):
    def once():
        if kf > 1:
            y, zT, zs, _ = wb.clipper_asym_fwd_tp(xd, th, FS, mode, kf, W, tol=tol, want_stash=True, want_zT=True)
        else:
            y, zT, _, zs = wb.clipper_asym_fwd(xd, th, FS, mode, tol=tol, want_stash=True, want_zT=True)
        gy = (y - tgt) * (2.0 / y.numel())
        if kb == 0:
            return wb.clipper_asym_bwd(xd, th, FS, zs, gy, tol=tol)
        return wb.clipper_asym_bwd_tp(xd, th, FS, mode, zs, zT, gy, kb)
    once()
    torch.cuda.synchronize()
    e0, e1 = wb.Event(), wb.Event()
    n = 3
    e0.record()
    for _ in range(n):
        g = once()
    e1.record()
    return e0.elapsed_ms(e1) / n, g.cpu().numpy().astype(np.float64)


g_seq_pick = picked_grad(wb.ASYM_NEWTON_F64, 0)
for name, mode in (("fp64 Newton tol=1e-12", wb.ASYM_NEWTON_F64), ("fp32 Wright-omega closed form", wb.ASYM_OMEGA_F32)):
    cases = ([(1, 0), (16, 0)] if mode == wb.ASYM_NEWTON_F64 else []) + [(16, 8), (16, 16), (16, 32), (32, 16), (32, 32)]
    for kf, kb in cases:
        ms, g = run_fwd_bwd(mode, kf, kb)
        gp = picked_grad(mode, kb)
        row = {"root": name, "step": "forward + reverse sweep", "fwd_chunks": kf,
               "reverse": "sequential, Newton re-solve per step" if kb == 0 else f"time-parallel, {kb} chunks, no re-solve",
               "ms": ms, "samples_per_s": B * T / ms * 1e3,
               "max_rel_grad_err_vs_fd_of_exact_model_32seq": float(np.max(np.abs(gp - g_fd) / np.abs(g_fd)))}
        if mode == wb.ASYM_NEWTON_F64:
            row["max_rel_grad_err_vs_sequential_sweep_32seq"] = float(np.max(np.abs(gp - g_seq_pick) / np.abs(g_seq_pick)))
        rows.append(row)
for r in rows:
    print(json.dumps(r))
