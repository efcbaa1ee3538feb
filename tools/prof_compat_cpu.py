"""CPU: where the compat tier spends its host time -- tests/loops.BridgedLadder recorded for 1280 steps with the kernel launch\nreplaced by a stub (as tests/test_trace_cpu.py does): the call under cProfile.  The per-step symbolic replay of the script's\nown loop is >= 97 % of it; the lowering at stack() is noise."""
import sys, time, cProfile, pstats
import os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib")); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
import tf_wdf as wdf
from wdf_hip import lowering, trace
from loops import BridgedLadder
class FakeFn:
    @staticmethod
    def apply(coef, rootp, x, z0, ns, ni, kind, n_up, n_down, want_zT):
        T, B = x.shape[1], x.shape[0]
        return (coef.sum() * 0.0 + torch.zeros(T, B)).float(), torch.zeros(ns, B)
trace._device = lambda: torch.device("cpu")
lowering._StateSpaceFn.apply = FakeFn.apply
lad = BridgedLadder(wdf, 48000)
x = np.random.default_rng(0).standard_normal((2, 1280)).astype(np.float32)
lad.run(x)
def best(fn, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
print("ladder 1280 steps: %.1f ms" % best(lambda: (lad.reset(), lad.run(x))))
pr = cProfile.Profile(); pr.enable(); lad.reset(); lad.run(x); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
