import os, sys, subprocess, json
sys.path.insert(0, "differentiable-wdfs_amd/lib")
import numpy as np, torch
from wdf_hip import binding as wb, workload
FS = workload.FS
mode = os.environ.get("WDF_FUSED_FINISH", "later")
out = {}
for (B, T, K, W, esr, with_r) in [(128, 8192, 256, 192, False, False), (70, 4096, 128, 192, False, False), (2, 2048, 64, 192, True, False),
                                  (300, 16384, 512, 160, False, True), (1, 1024, 32, 192, False, False), (129, 3000, 47, 192, True, True)]:
    x = torch.as_tensor(workload.sweep_batch(max(B, 2), T, seed=B + K)[:B], device="cuda")
    th = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda")
    ths = torch.tensor(workload.target_theta(), dtype=torch.float32, device="cuda")
    r = torch.as_tensor(workload.pot_resistance_batch(B, T), device="cuda") if with_r else None
    tgt, _, _ = wb.clipper_fwd(x, ths, FS, r=r, want_stash=False)
    K = wb.lib().wdf_clipper_tp_chunks(T, K)
    if esr:
        y, _, s10, g, l3, st = wb.clipper_step_esr_tp(x, th, FS, tgt, float(B * (T - 50)), 2.2e-16, 50, K, W, r=r)
        res = (float(y.double().sum()), [float(v) for v in g], [float(v) for v in l3])
    else:
        y, _, g, sse, st = wb.clipper_step_mse_tp(x, th, FS, tgt, 2.0 / (B * T), K, W, r=r)
        res = (float(y.double().sum()), [float(v) for v in g], float(sse))
    out[f"{B}x{T} K{K} esr{esr} r{with_r}"] = (res, wb.tp_status(st))
print(json.dumps(out))
