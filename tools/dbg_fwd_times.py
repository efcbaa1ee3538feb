"""GPU, a -DWDF_DBG_TIMES build (WDF_HIP_LIB=...): per-wave stamps of the stateless time-parallel forward at BASELINE configs[1]'s
shape (1024 x 4096): start, end of the body, end of the wave (after the tile's verification), shader clock, placement."""
import os, sys, ctypes as C, collections
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, workload
B, T, FS = 1024, 4096, workload.FS
x = torch.as_tensor(workload.sweep_batch(B, T, seed=3), device="cuda").t().contiguous()
theta = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda")
L = wb.lib(); L.wdf_debug_set_times.argtypes = [C.c_void_p]
for k in [int(v) for v in os.environ.get("C2_KS", "16,32,64,128").split(",")]:
    K = wb.lib().wdf_clipper_tp_chunks(T, k)
    nw = (B // 64) * K
    buf = torch.zeros(8 * nw, dtype=torch.int64, device="cuda")
    f = lambda: wb.clipper_fwd_tp(x, theta, FS, k, 160, 1e-6, want_stash=False, time_major=True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    assert L.wdf_debug_set_times(buf.data_ptr()) == 0
    e0, e1 = wb.Event(), wb.Event()
    wb.Event.bracket_next(e0, e1)
    f(); torch.cuda.synchronize()
    L.wdf_debug_set_times(None)
    a = buf.cpu().numpy().reshape(nw, 8)
    t0, t1, t2 = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64), a[:, 4].astype(np.float64)
    base, tick = t0.min(), 1e-2
    hw = a[:, 2]
    simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 0xf
    cnt = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
    life = (t1 - t0) * tick
    mhz = a[:, 3].astype(np.float64) / ((t1 - t0) * tick)
    kk = np.arange(nw) // (B // 64)
    print(f"chunks {K}: kernel (events) {e0.elapsed_ms(e1)*1e3:.1f} us; waves {nw}; waves per SIMD {sorted(collections.Counter(cnt.values()).items())}")
    print(f"  start: median {np.median(t0-base)*tick:.1f} max {(t0.max()-base)*tick:.1f} us; body end: min {(t1.min()-base)*tick:.1f} median {np.median(t1-base)*tick:.1f} "
          f"p90 {np.percentile(t1-base,90)*tick:.1f} max {(t1.max()-base)*tick:.1f} us; wave end max {(t2.max()-base)*tick:.1f} us")
    print(f"  body lifetime: min {life.min():.1f} median {np.median(life):.1f} max {life.max():.1f} us; shader clock over it: median {np.median(mhz):.0f} MHz (min {mhz.min():.0f})")
    print("  body lifetime median by chunk:", " ".join(f"{q}:{np.median(life[kk == q]):.0f}" for q in range(0, K, max(1, K // 16))))
