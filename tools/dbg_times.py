import os, sys, ctypes as C
sys.path.insert(0, "differentiable-wdfs_amd/lib")
import numpy as np, torch
from wdf_hip import binding, engine, workload
one = len(sys.argv) > 1 and sys.argv[1] == "one"
binding.ONE_SEQUENCE_PER_LANE = one
B, T, fs, K = 8192, int(os.environ.get("DBG_T", "4096")), workload.FS, int(os.environ.get("DBG_K", "16"))
dev = torch.device("cuda", 0)
TM = not os.environ.get("DBG_BM")          # DBG_BM=1: x batch-major ([B, T], as the caller holds it)
x = torch.as_tensor(workload.sweep_batch(B, T), device=dev); xt = x.t().contiguous() if TM else x
th_host = workload.clipper_theta()
tgt, _, _ = binding.clipper_fwd(x, torch.tensor(workload.target_theta(), dtype=torch.float32, device=dev), fs, want_stash=False)
st = engine.MseStep(B, T, fs, engine.TpPlan(K, 160, 1e-6, 32), dev, time_major=TM, warm=True)
theta = torch.tensor(th_host, dtype=torch.float32, device=dev)
adam = None if os.environ.get("DBG_NO_ADAM") else binding.Adam(4, lr=[1e-3 * float(v) for v in th_host], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=dev)
nw = (B // (64 if one else 128)) * K
ntile = B // (64 if one else 128)
buf = torch.zeros(8 * nw + 8 * ntile, dtype=torch.int64, device=dev)
L = binding.lib(); L.wdf_debug_set_times.argtypes = [C.c_void_p]
for _ in range(int(os.environ.get("DBG_CALLS", "6"))): st.step_fused(theta, xt, tgt, adam=adam)
assert L.wdf_debug_set_times(buf.data_ptr()) == 0
e0, e1 = binding.Event(), binding.Event()
binding.Event.bracket_next(e0, e1)
st.step_fused(theta, xt, tgt, adam=adam)
torch.cuda.synchronize()
ms = e0.elapsed_ms(e1)
L.wdf_debug_set_times(None)
allv = buf.cpu().numpy()
a = allv[:8 * nw].reshape(nw, 8)
tail = allv[8 * nw:].reshape(ntile, 8).astype(np.float64)
t0, t1 = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64)
base = t0.min()
tick = 1e-2   # wall_clock64: 100 MHz -> 10 ns = 0.01 us
print(f"kernel (events) {ms*1e3:.1f} us; waves {nw}")
print(f"start: min 0, median {np.median(t0-base)*tick:.1f} us, p90 {np.percentile(t0-base,90)*tick:.1f}, max {(t0.max()-base)*tick:.1f} us")
print(f"end:   min {(t1.min()-base)*tick:.1f}, median {np.median(t1-base)*tick:.1f}, p90 {np.percentile(t1-base,90)*tick:.1f}, max {(t1.max()-base)*tick:.1f} us")
life = (t1 - t0) * tick
print(f"lifetime: min {life.min():.1f} median {np.median(life):.1f} p90 {np.percentile(life,90):.1f} max {life.max():.1f} us")
mt = a[:, 3].astype(np.float64)
print(f"s_memtime ticks per wall-clock us over the wave's life: median {np.median(mt / ((t1 - t0) * tick)):.1f} (= MHz if s_memtime is the shader clock)")
ph = a[:, 4:8].astype(np.float64) / 2100.0
print("phases (us at 2.1 GHz), median / p90: set-up + first loads issued %.1f / %.1f; warm-up loop %.1f / %.1f; publish + drain %.1f / %.1f; owned loop + record %.1f / %.1f" % tuple(
    v for i in range(4) for v in (np.median(ph[:, i]), np.percentile(ph[:, i], 90))))
wt = a[:, 2].astype(np.float64)
print(f"shader cycles parked in the back-edge s_waitcnt per wave: median {np.median(wt):.0f} of {np.median(mt):.0f} ({100*np.median(wt/mt):.1f} %)")
if os.environ.get("DBG_HWID"):      # a -DWDF_DBG_HWID build: slot 2 is HW_ID | XCC_ID << 32 (wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13])
    import collections
    hw = a[:, 2]
    simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 0xf
    cnt = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
    ccu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    print("SIMDs used:", len(cnt), "waves per SIMD:", sorted(collections.Counter(cnt.values()).items()), "; CUs used:", len(ccu),
          "waves per CU:", sorted(collections.Counter(ccu.values()).items()))
    per = np.array([cnt[t] for t in zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist())])
    for n_on in sorted(set(per.tolist())):
        m = per == n_on
        print(f"  waves on a SIMD with {n_on}: {m.sum()}, lifetime median {np.median(life[m]):.1f} us, end median {np.median(t1[m]-base)*tick:.1f} us")
    print("  lifetime median by XCC:", " ".join(f"{xx}:{np.median(life[xcc == xx]):.1f}" for xx in sorted(set(xcc.tolist()))))
    print("  lifetime median by SE:", " ".join(f"{xx}:{np.median(life[se == xx]):.1f}" for xx in sorted(set(se.tolist()))))
    kk_ = np.arange(nw) // (nw // K)
    print("  lifetime median by chunk:", " ".join(f"{q}:{np.median(life[kk_ == q]):.0f}" for q in range(K)))
    tile_ = np.arange(nw) % (nw // K)
    print("  lifetime median by tile (every 4th):", " ".join(f"{q}:{np.median(life[tile_ == q]):.0f}" for q in range(0, nw // K, 4)))
    # does a SIMD's pair finish together?  spread inside a SIMD vs across SIMDs
    key = xcc * 100000 + se * 10000 + sh * 5000 + cu * 10 + simd
    order = np.argsort(key, kind="stable")
    ks, ls = key[order], life[order]
    same = ks[1:] == ks[:-1]
    print(f"  |lifetime difference| of the two waves of a SIMD: median {np.median(np.abs(ls[1:] - ls[:-1])[same]):.1f} us; SIMD mean lifetime: min {min(np.mean(life[key == q]) for q in set(key.tolist())):.1f} max {max(np.mean(life[key == q]) for q in set(key.tolist())):.1f}")
    cukey = key // 10
    cm = np.array([np.mean(life[cukey == q]) for q in sorted(set(cukey.tolist()))])
    print(f"  CU mean lifetime: min {cm.min():.1f} p10 {np.percentile(cm,10):.1f} median {np.median(cm):.1f} p90 {np.percentile(cm,90):.1f} max {cm.max():.1f}")
    sys.exit(0)
k = np.arange(nw) // (nw // K)
print("chunk: end median us:", " ".join(f"{kk}:{np.median(t1[k == kk]-base)*tick:.0f}" for kk in range(K)))
if tail[:, 7].max() > 0:      # stamps along the tile's tail (the build has them): the tile that finished the step = latest stamp 7
    fin = int(np.argmax(tail[:, 7]))
    names = ["last body end", "tile ticket", "verified", "records walked", "partial landed", "step ticket", "reduced + chain rule", "Adam done"]
    tl = (tail - base) * tick
    print("finishing tile %d, us from kernel start: " % fin + "; ".join(f"{n} {tl[fin, i]:.1f}" for i, n in enumerate(names)))
    print("all tiles, median us: " + "; ".join(f"{n} {np.median(tl[:, i]):.1f}" for i, n in enumerate(names[:6])))
    print(f"events {ms*1e3:.1f} us vs kernel-start -> Adam done {tl[fin, 7]:.1f} us: launch + drain = {ms*1e3 - tl[fin, 7]:.1f} us")
sys.exit(0)
t2, t3 = a[:, 2].astype(np.float64), a[:, 3].astype(np.float64)
last = t2 > 0
print(f"tile-last waves: {last.sum()}; verify took median {np.median((t2-t1)[last])*tick:.2f} us max {((t2-t1)[last]).max()*tick:.2f}; "
      f"combine+finish median {np.median((t3-t2)[last & (t3>0)])*tick:.2f} max {((t3-t2)[last & (t3>0)]).max()*tick:.2f} us; "
      f"last body end {(t1.max()-base)*tick:.1f} us, last finish end {(t3.max()-base)*tick:.1f} us")
sys.exit(0)
hw = a[:, 2]
# HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7; xcc = (hw >> 16) & 0xf
key = (((hw >> 4) & 0xfffff))
import collections
cnt = collections.Counter(zip(xcc.tolist(), se.tolist(), cu.tolist(), simd.tolist()))
print("distinct SIMDs used:", len(cnt), "waves per SIMD histogram:", collections.Counter(cnt.values()))
k = np.arange(nw) // (nw // K)
for kk in (0, 1, K // 2, K - 1):
    m = k == kk
    print(f"chunk {kk}: start median {np.median(t0[m]-base)*tick:.1f} end median {np.median(t1[m]-base)*tick:.1f} life median {np.median(life[m]):.1f}")
