"""GPU: what the device tape interpreter (csrc/wdf_ss_dyn_rows.h) costs per operation and per row entry: synthetic tapes --
a chain of n_ops operations on the channel and one parameter -- at the reference's training-set shape (1340 x 2048)."""
import os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb

B, T = 1340, 2048
r = torch.rand((T, B), device="cuda") * 1.0e3 + 100.0
params = torch.tensor([2.0e3, 3.0e-8], dtype=torch.float64, device="cuda")


def tape(n_ops, kind):
    ops = [[1, 0, 0], [1, 1, 0]]                                  # PARAM 0 (the channel), PARAM 1
    while len(ops) < n_ops:
        i = len(ops)
        ops.append({"add": [2, i - 1, i - 2], "mul": [4, i - 1, 0], "div": [5, i - 2, i - 1]}[kind] if kind != "mix" else
                   [[2, i - 1, i - 2], [4, i - 1, 0], [3, i - 1, 1], [5, i - 2, i - 1]][i % 4])
    return ops


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for kind in ("add", "mix", "div"):
    for n_ops in (4, 16, 31, 62, 124):
        for n_out in (1, 9):
            ops = tape(n_ops, kind)
            rt = wb.RowsTape(ops, [], [n_ops - 1 - (k % 2) for k in range(n_out)])
            grows = torch.randn((T, n_out, B), device="cuda")
            f = timed(lambda: wb.ss_dyn_rows(rt, params, 0, r))
            b = timed(lambda: wb.ss_dyn_rows_bwd(rt, params, 0, r, grows))
            print(f"{kind:4s} n_ops {n_ops:4d} n_out {n_out:2d}: rows {f:8.1f} us   rows_bwd {b:8.1f} us")
