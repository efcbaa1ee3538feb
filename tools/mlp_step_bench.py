#!/usr/bin/env python3
"""tools/mlp_step_bench.py [net] [steps] -- the resident MLP-root training step (csrc/wdf_mlp_step.h) at the reference's
training-set shape (1340 x 2048, pot value per sample): eager and HIP-graph step time, forward / reverse-sweep kernel
times, the controller's per-column warm-ups before and after the re-plan."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "differentiable-wdfs_amd", "lib"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wdf_hip import binding, mlp_root, workload  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "2x16_pre"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B, T, fs, skip = int(os.environ.get("B", 1340)), 2048, workload.FS, 50
dev = torch.device("cuda", 0)
x = torch.as_tensor(workload.sweep_batch(B, T, seed=4) * 0.6, device=dev)
r = torch.as_tensor(workload.dataset_resistance_batch(B, T), device=dev)
wh, hidden, n_layers = workload.reference_mlp_weights(net)
w = torch.tensor(wh, device=dev)
th4 = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device=dev)
target, _, _ = binding.clipper_fwd(x, th4, fs, r=r, want_stash=False)
adam = binding.Adam(w.numel(), lr=1.0e-4, beta_1=0.5, device=dev)
kw = {}
if os.environ.get("N_ITEMS"):
    kw["n_items"] = int(os.environ["N_ITEMS"])
if os.environ.get("KW"):
    kw["wgrad_chunks"] = int(os.environ["KW"])
st = mlp_root.MlpTrainStep(x, r, target, w, hidden, n_layers, fs, workload.C_CLIPPER, skip=skip, adam=adam, **kw)
print(f"net {net}: {B} x {T}, items {st.n_items}, wgrad chunks {st.wgrad_chunks}, cold warm-up {st.cold}")
if os.environ.get("FREEZE") == "1":                      # the controller leaves the first-guess warm-ups alone
    st.freeze(True)


def chunks_per_class():
    return [int((st.items[:, 0] == c).sum()) for c in (0, st.ncol // 3, 2 * st.ncol // 3, st.ncol - 1)]


def timed(n, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def report(tag):
    info, wc = st.read()
    q = [int(v) for v in (wc[0], wc[st.ncol // 3], wc[2 * st.ncol // 3], wc[-1])]
    print(f"  [{tag}] verdict {info}  W(units) by pot class {q}  chunks by pot class {chunks_per_class()}  loss {float(st.loss3[2]):.5e}")


st.step()
report("cold call")
print(f"eager, first 30 calls: {timed(30, st.step):.4f} ms/step")
report("30 calls")
print(f"eager, next 30: {timed(30, st.step):.4f} ms/step")
report("60 calls")
if os.environ.get("REPLAN", "1") == "1":
    print("re-plan:", st.replan())
print(f"eager after re-plan, {steps} calls: {timed(steps, st.step):.4f} ms/step")
report("eager")
# kernel times: phases launched separately with the bracket around the recurrence kernel
ev = [binding.Event() for _ in range(4)]
tf, tb = [], []
for _ in range(10):
    binding.Event.bracket_next(ev[0], ev[1])
    st.forward_only()
    binding.Event.bracket_next(ev[2], ev[3])
    st.backward_only()
    torch.cuda.synchronize()
    tf.append(ev[0].elapsed_ms(ev[1]))
    tb.append(ev[2].elapsed_ms(ev[3]))
print(f"kernels: forward chunks {np.median(tf):.4f} ms, reverse sweep {np.median(tb):.4f} ms")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        st.step()
torch.cuda.current_stream().wait_stream(side)
print(f"HIP graph, {steps} replays: {timed(steps, g.replay):.4f} ms/step")
report("graph")
print(f"HIP graph, {steps} more: {timed(steps, g.replay):.4f} ms/step")
report("graph 2")
