#!/usr/bin/env python3
"""tools/mlp_step_trace.py [net] [calls] -- per-call trace of the resident MLP-root step's controller: ms per step (events
around the whole step), boundaries flagged, columns gone sequential, max miss, per-class warm-ups.  Env: SLACK, SHRINK_AT,
GROW_AT, COOL_MISS, COOL_SHRINK override the controller's knobs; REPLAN_AT (default 40)."""
import os
import struct
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "differentiable-wdfs_amd", "lib"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wdf_hip import binding, mlp_root, workload  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "2x16_pre"
net = {"2x16": "2x16_pre"}.get(net, net) if os.environ.get("PRE", "1") == "1" else net
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B, T, fs, skip = 1340, 2048, workload.FS, 50
dev = torch.device("cuda", 0)
x = torch.as_tensor(workload.sweep_batch(B, T, seed=4) * 0.6, device=dev)
r = torch.as_tensor(workload.dataset_resistance_batch(B, T), device=dev)
wh, hidden, n_layers = workload.reference_mlp_weights(net)
w = torch.tensor(wh, device=dev)
th4 = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device=dev)
target, _, _ = binding.clipper_fwd(x, th4, fs, r=r, want_stash=False)
adam = binding.Adam(w.numel(), lr=1.0e-4, beta_1=0.5, device=dev)
st = mlp_root.MlpTrainStep(x, r, target, w, hidden, n_layers, fs, workload.C_CLIPPER, skip=skip, adam=adam)
f2i = lambda v: struct.unpack("i", struct.pack("f", float(v)))[0]  # noqa: E731
for name, field, conv in (("SLACK", 6, int), ("COOL_MISS", 7, int), ("COOL_SHRINK", 8, int), ("GROW_AT", 10, f2i), ("SHRINK_AT", 11, f2i), ("REPAIR_AT", 13, f2i)):
    if os.environ.get(name):
        binding._check(st.lib.wdf_clipper_mlp_step_set(binding._ptr(st.state), field, conv(os.environ[name]), binding._stream()), name)
replan_at = int(os.environ.get("REPLAN_AT", 40))
e0, e1 = binding.Event(), binding.Event()
ms, bad, seq = [], [], []
q = [0, st.ncol // 3, 2 * st.ncol // 3, st.ncol - 1]
for i in range(calls):
    e0.record()
    st.step()
    e1.record()
    torch.cuda.synchronize()
    info, wc = st.read()
    ms.append(e0.elapsed_ms(e1)); bad.append(info["n_bad"]); seq.append(info["sequential_columns"])
    if i < 12 or i % 10 == 0 or info["n_bad"]:
        print(f"{i:4d} {ms[-1]:.3f} ms  bad {info['n_bad']:3d} cols {info['flagged_columns']:2d} seq {info['sequential_columns']}  "
              f"miss {info['max_miss']:.1e}  W {[int(wc[c]) for c in q]} (mean {wc.mean():.1f})")
    if os.environ.get("DUMP_COLS") and i in (int(v) for v in os.environ["DUMP_COLS"].split(",")):
        cm = st.column_misses()
        for c in range(st.ncol):
            print(f"      col {c:2d} K {int((st.items[:, 0] == c).sum()):2d} W {int(wc[c]):2d}  m0 {cm[c, 0]:.1e}  -1: {cm[c, 1]:.1e}  -2: {cm[c, 2]:.1e}  -3: {cm[c, 3]:.1e}")
    if i == replan_at and os.environ.get("AUTOTUNE"):
        print("autotune:", st.autotune(verbose=True))
    elif i == replan_at:
        print("re-plan:", st.replan(), [int((st.items[:, 0] == c).sum()) for c in q])
        wp = st.warmup_peaks()
        print("   plan K per column:", np.bincount(st.items[:, 0]).tolist())
        print("   W at plan        :", wc.tolist())
h = calls // 2
clean = [m for m, b in zip(ms[h:], bad[h:]) if b == 0]
print(f"second half: mean {np.mean(ms[h:]):.4f} ms, median {np.median(ms[h:]):.4f}; calls with a flagged boundary {sum(b > 0 for b in bad[h:])}/{calls - h}, "
      f"sequential {sum(seq[h:])}; clean calls {np.mean(clean):.4f} ms")

ev = [binding.Event() for _ in range(4)]
tf, tb = [], []
for _ in range(12):
    binding.Event.bracket_next(ev[0], ev[1]); st.forward_only()
    binding.Event.bracket_next(ev[2], ev[3]); st.backward_only()
    torch.cuda.synchronize()
    tf.append(ev[0].elapsed_ms(ev[1])); tb.append(ev[2].elapsed_ms(ev[3]))
pl = st.placement()
from collections import Counter
print(f"kernels: forward {np.median(tf):.4f} ms, reverse sweep {np.median(tb):.4f} ms; forward waves per SIMD {sorted(Counter(Counter(map(tuple, pl)).values()).items())}")
