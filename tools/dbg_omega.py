import sys, numpy as np, torch
sys.path.insert(0, "differentiable-wdfs_amd/lib"); sys.path.insert(0, "oracle")
from wdf_hip import binding as wb
import oracle as O
rng = np.random.default_rng(3)
x32 = np.concatenate([rng.uniform(-110, 110, 400000), rng.uniform(-5, 6, 400000), np.linspace(-6,8,200001)]).astype(np.float32)
w, it = wb.omega(torch.as_tensor(x32, device="cuda"), want_iters=True)
w = w.cpu().numpy().astype(np.float64); it = it.cpu().numpy()
ref = O.wright_omega(x32.astype(np.float64))
err = np.abs(w - ref) / np.maximum(np.abs(ref), 1e-35)
i = np.argsort(err)[-15:]
for j in i: print(x32[j], w[j], ref[j], err[j], it[j])
for lo, hi in [(-110,-87),(-87,-4),(-4,-2),(-2,4.1415),(4.1416,110)]:
    m = (x32>lo)&(x32<=hi); print(lo,hi,err[m].max(), np.unique(it[m]))
