#!/usr/bin/env python3
"""tools/mlp_step_placement.py -- where the dispatcher places the 1008 single-wave workgroups of the resident MLP-root step's
forward (one per SIMD wanted), for several reverse-sweep grids launched before it."""
import os
import sys
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "differentiable-wdfs_amd", "lib"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wdf_hip import binding, mlp_root, workload  # noqa: E402

B, T, fs, skip = 1340, 2048, workload.FS, 50
dev = torch.device("cuda", 0)
x = torch.as_tensor(workload.sweep_batch(B, T, seed=4) * 0.6, device=dev)
r = torch.as_tensor(workload.dataset_resistance_batch(B, T), device=dev)
wh, hidden, n_layers = workload.reference_mlp_weights("2x16_pre")
th4 = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device=dev)
target, _, _ = binding.clipper_fwd(x, th4, fs, r=r, want_stash=False)
for kw in (24, 32, 43, 64):
    for n_items in (1008, 1024, 924):
        w = torch.tensor(wh, device=dev)
        adam = binding.Adam(w.numel(), lr=1.0e-4, beta_1=0.5, device=dev)
        st = mlp_root.MlpTrainStep(x, r, target, w, hidden, n_layers, fs, workload.C_CLIPPER, skip=skip, adam=adam, wgrad_chunks=kw, n_items=n_items)
        for _ in range(6):
            st.step()
        ev = [binding.Event() for _ in range(2)]
        ts = []
        for _ in range(6):
            st.backward_only()
            binding.Event.bracket_next(ev[0], ev[1])
            st.forward_only()
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_ms(ev[1]))
        pl = st.placement()
        per_simd = Counter(map(tuple, pl))
        per_cu = Counter(map(tuple, pl[:, :2]))
        print(f"KW {kw:2d} items {st.n_items}: forward {np.median(ts):.4f} ms; SIMDs used {len(per_simd)}, waves per SIMD {sorted(Counter(per_simd.values()).items())}; "
              f"CUs used {len(per_cu)}, waves per CU {sorted(Counter(per_cu.values()).items())}")
