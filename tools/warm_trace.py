"""GPU: the one-pass step's warm-start controller inside the bench's training loop: per step the warm-up units the device
chose, the boundary miss it measured, failures; one line per step (compact)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, engine, workload
B, T, fs = 8192, 4096, workload.FS
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda")
x = torch.as_tensor(workload.sweep_batch(B, T), device=dev)
xt = x.t().contiguous()
th_host = workload.clipper_theta()
target, _, _ = wb.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device=dev), fs, want_stash=False)
theta = torch.tensor(th_host, dtype=torch.float32, device=dev)
st = engine.MseStep(B, T, fs, engine.TpPlan(K, 192, 1e-6, 32), dev, time_major=True, warm=True)
adam = wb.Adam(4, lr=[1e-3 * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=dev)
out = []
for s in range(steps):
    st.step_fused(theta, xt, target, adam=adam)
    stat, info = wb.tp_status(st.status), st.warm.info()
    out.append(f"{s}:{info['last_warm_tiles']}/{stat['max_miss']:.1e}{'!' if stat['n_bad'] else ''}")
print("step:units/miss (! = a boundary failed):")
for i in range(0, len(out), 10):
    print("  " + "  ".join(out[i:i + 10]))
