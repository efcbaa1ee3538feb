"""Wall-clock stamps along the one-pass step's finish launch (clipper_fused_finish_kernel), per tile's wave 0.
Needs a -DWDF_DBG_TIMES build of the clipper translation unit linked into another library (tools/dbg_fin_build.sh):
WDF_HIP_LIB=$PWD/differentiable-wdfs_amd/lib/wdf_hip/libwdf_dbg.so python tools/dbg_fin_times.py [B] [K]"""
import os, sys, ctypes as C
sys.path.insert(0, "differentiable-wdfs_amd/lib")
import numpy as np, torch
from wdf_hip import binding, engine, workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T, fs = 4096, workload.FS
dev = torch.device("cuda", 0)
x = torch.as_tensor(workload.sweep_batch(8192, T, b0=0, b1=B), device=dev); xt = x.t().contiguous()
th_host = workload.clipper_theta()
tgt, _, _ = binding.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device=dev), fs, want_stash=False)
st = engine.MseStep(B, T, fs, engine.TpPlan(K, 192, 1e-6, 32), dev, time_major=True, warm=True)
theta = torch.tensor(th_host, dtype=torch.float32, device=dev)
adam = binding.Adam(4, lr=[1e-3 * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=dev)
ntile = B // 128
nw = ntile * K
buf = torch.zeros(8 * nw + 8 * ntile, dtype=torch.int64, device=dev)
L = binding.lib(); L.wdf_debug_set_times.argtypes = [C.c_void_p]
for _ in range(10): st.step_fused(theta, xt, tgt, adam=adam)
assert L.wdf_debug_set_times(buf.data_ptr()) == 0
st.step_fused(theta, xt, tgt, adam=adam)
torch.cuda.synchronize()
L.wdf_debug_set_times(None)
allv = buf.cpu().numpy()
a = allv[:8 * nw].reshape(nw, 8).astype(np.float64)
tail = allv[8 * nw:].reshape(ntile, 8).astype(np.float64)
tick = 1e-2
chunk_end = a[:, 1].max()
print(f"B {B} K {K}: chunk kernel's last wave body ends at 0; the finish launch's stamps (us after that), min / median / max over {ntile} tiles")
names = ["kernel entered", "loads + boundary check done", "maps composed (phase A)", "walk + wave sums done (phase B)", "tile totals ready"]
for i, n in enumerate(names):
    v = (tail[:, i] - chunk_end) * tick
    v = v[tail[:, i] > 0]
    if len(v):
        print(f"  {i} {n:45s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}")
