"""GPU: host time of the compat tier -- a clipper_pot-shaped recorded forward (pot clipper with the reference's
2x16 root, 2048 steps, hand-written loop of tests/loops.py) and a linear ladder loop: wall time of the forward
call (the script's Python loop running symbolically + the lowering + the kernels), best of 5."""
import os, sys, time
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib")); sys.path.insert(0, os.path.join(_R, "tests"))
import tf_wdf as wdf
from loops import PotClipper, BridgedLadder
from test_gpu_mlp_root import model_json
from wdf_hip import workload
g = np.load(os.path.join(_R, "tests", "golden", "g3_mlp_clipper.npz"))
B, T = 64, 2048
data = np.stack([workload.sweep_batch(B, T, seed=2) * 0.6, workload.dataset_resistance_batch(B, T)], axis=-1)


def best(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


m = PotClipper(wdf, 48000, float(g["C"]), mlp_json=model_json(g, "2x16"))
m.run(data)
print(f"pot clipper, MLP root, {T} recorded steps: {best(lambda: m.run(data)):.1f} ms per forward (round 1: 640 ms)")
lad = BridgedLadder(wdf, 48000)
x = np.random.default_rng(0).standard_normal((2, 1280)).astype(np.float32)
lad.run(x)
print(f"bridged ladder, 1280 recorded steps: {best(lambda: (lad.reset(), lad.run(x))):.1f} ms per forward")
