"""GPU: what limits the MLP-root forward's warm-up.  Runs the bench's training loop (bench.py run_mlp_root: Adam 1e-4,
beta_1 0.5, MSE + ESR) with the SEQUENTIAL forward, keeping the exact state trajectories of the last calls; then, at a
late call, asks how well the next call's chunk-start states can be predicted from the history (order 0 / secant /
parabola) and what the chunked forward's verification says for each predictor and warm-up length.
usage: python tools/mlp_start_probe.py [root=2x16] [calls=120]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding, workload, mlp_root
root = sys.argv[1] if len(sys.argv) > 1 else "2x16"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda")
fs, T, B = workload.FS, 2048, 1340
x = torch.as_tensor(workload.sweep_batch(B, T, seed=4) * 0.6, device=dev)
r = torch.as_tensor(workload.dataset_resistance_batch(B, T), device=dev)
wh, hidden, n_tanh = workload.reference_mlp_weights(root + "_pre" if root == "2x16" else root)
w = torch.tensor(wh, device=dev)
theta2 = torch.tensor([45.0e3, workload.C_CLIPPER], dtype=torch.float32, device=dev)
target, _, _ = binding.clipper_fwd(x, torch.tensor(workload.clipper_theta(), dtype=torch.float32, device=dev), fs, r=r, want_stash=False)
skip, eps = 50, float(np.finfo(float).eps)
n_global = float(B * (T - skip))
plan = mlp_root.plan_mlp_time_parallel(B, T, r, None, workload.C_CLIPPER, fs, hidden=hidden, n_tanh=n_tanh)
print("plan", plan)
adam = binding.Adam(w.numel(), lr=1.0e-4, beta_1=0.5, device=dev)
sums = torch.zeros(2, dtype=torch.float64, device=dev)
gcoef, loss3 = torch.zeros(2, dtype=torch.float32, device=dev), torch.zeros(3, dtype=torch.float32, device=dev)
hist = []
K = plan.k_fwd
grid = (96, 128, 160, 192, 224, 256, 288)
idxs = {W: torch.tensor(binding.mlp_tp_starts(T, K, W), dtype=torch.int64, device=dev) for W in grid}
need, jumps = [], []
for n in range(calls):
    if hist and os.environ.get("EVERY_CALL"):
        # the smallest warm-up of the grid at which the chunked forward, started from the previous call's states, verifies clean
        ok = None
        for W in grid:
            out = binding.clipper_mlp_fwd_tp(x, theta2, w, hidden, n_tanh, fs, K, W, r=r, tol=plan.tol, want_stash=True, want_kappa=True,
                                             zinit=hist[0].index_select(0, idxs[W]).contiguous())
            if binding.mlp_tp_status(out[3])["gated_waves"] == 0:
                ok = W
                break
        need.append(ok)
    y, zs, _ = binding.clipper_mlp_fwd(x, theta2, w, hidden, n_tanh, fs, r=r)
    if hist:
        jumps.append(float((zs - hist[0]).abs().max()))
    hist = ([zs] + hist)[:3]
    binding.loss_sums(y, target, skip, sums=sums)
    binding.esr_coef(sums, n_global, eps, gcoef=gcoef, loss=loss3)
    gy = binding.loss_esr_grad(y, target, gcoef, skip)
    _, gw = binding.clipper_mlp_bwd_w(x, theta2, w, hidden, n_tanh, fs, zs, gy, r=r)
    adam.apply(w, gw)
if need:
    print("per call: least warm-up of", grid, "that verifies clean from the previous call's states (None: none of them) / largest state change:")
    for i in range(0, len(need), 10):
        print("  " + "  ".join(f"{w}/{j:.0e}" for w, j in zip(need[i:i + 10], jumps[i:i + 10])))
# the call to predict: the trajectory under the weights as they are now
_, zs_now, _ = binding.clipper_mlp_fwd(x, theta2, w, hidden, n_tanh, fs, r=r)
z1, z2, z3 = hist
d1, d2 = z1 - z2, z2 - z3
a = float((d1 * d2).sum() / (d2 * d2).sum())                 # AR(1) on the differences, fitted on the last two
a_seq = ((d1 * d2).sum(dim=0) / (d2 * d2).sum(dim=0).clamp_min(1e-30)).clamp(-1.0, 1.0)      # ... per sequence
print(f"AR(1) coefficient of the call-to-call differences: global {a:+.3f}; per sequence: median {float(a_seq.median()):+.3f}, 10%..90% {float(torch.quantile(a_seq, 0.1)):+.3f}..{float(torch.quantile(a_seq, 0.9)):+.3f}")
pred = {"order 0": z1, "secant": 2 * z1 - z2, "ar1": z1 + a * d1, "ar1/seq": z1 + a_seq * d1, "back half": z1 - 0.5 * d1}
print(f"loss {float(loss3[2]):.3e}; |z| max {float(zs_now.abs().max()):.2f}; change of the trajectory per call: max {float((zs_now - z1).abs().max()):.2e}")
q = torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)
for name, p in pred.items():
    e = (p - zs_now).abs()
    per_seq = e.amax(dim=0)
    print(f"  {name:9s} |prediction - truth| over all samples: median {float(e.median()):.1e}  99% {float(torch.quantile(e.flatten()[::7], 0.99)):.1e}  max {float(e.max()):.1e};"
          f"  sequences whose worst sample is > 4e-6: {int((per_seq > 4e-6).sum())} / {B}, > 1e-5: {int((per_seq > 1e-5).sum())}")
# the chunked forward's verdict per predictor and warm-up
for W in (64, 96, 112, 128, 144, 160, 176, 192):
    idx = torch.tensor(binding.mlp_tp_starts(T, K, W), dtype=torch.int64, device=dev)
    line = f"W = {W:3d} (bad/max miss/gated): "
    for name, p in pred.items():
        zinit = p.index_select(0, idx).contiguous()
        out = binding.clipper_mlp_fwd_tp(x, theta2, w, hidden, n_tanh, fs, K, W, r=r, tol=plan.tol, want_stash=True, want_kappa=True, zinit=zinit)
        st = binding.mlp_tp_status(out[3])
        line += f" {name}: {st['n_bad']:4d}/{st['max_miss']:.1e}/{st['gated_waves']:3d} |"
    print(line)
