"""GPU: BASELINE config 5 -- clipper with two different antiparallel diodes, fp64 Newton (ballot
terminated) vs the fp32 Wright-omega closed form, against the oracle's exact solve.
Diode constants: up = 1N4148 (diode_config.py:14-16); down = "OA1154-like" germanium values
chosen for this build (Is = 2e-6 A, n = 1.4) -- the reference has none (the datasheet under
diode_dataset/OA1154 lists no Shockley parameters)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000.0
THETA6 = np.array([4.352e-9, 25.85e-3 * 1.906, 2.0e-6, 25.85e-3 * 1.4, 45.0e3, 4.7e-9])


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def test_roots_vs_oracle(oracle):
    from wdf_hip import binding as wb
    th = dev(THETA6)
    t32 = THETA6.astype(np.float32).astype(np.float64)
    Rp = 1.0 / (1.0 / t32[4] + 2.0 * t32[5] * FS)
    a = np.concatenate([np.linspace(-6, 6, 2001), np.linspace(-0.05, 0.05, 501)]).astype(np.float32)
    ref = oracle.asym_root(a.astype(np.float64), Rp, t32[0], t32[1], t32[2], t32[3])
    for tol, bound in ((1e-6, 2e-5), (1e-10, 2e-6), (1e-14, 2e-6)):
        b = wb.asym_root(dev(a), th, FS, wb.ASYM_NEWTON_F64, tol=tol, max_iter=50).cpu().numpy()
        # Rp is formed in fp32 inside the kernel: 6e-8 relative on Rp moves b by up to ~1e-6 V
        assert np.max(np.abs(b - ref)) < bound, (tol, np.max(np.abs(b - ref)))
    bw = wb.asym_root(dev(a), th, FS, wb.ASYM_OMEGA_F32).cpu().numpy()
    # closed form: fp32 rounding + the neglected reverse saturation current (2 Rp Is_down = 8.4 mV)
    assert np.max(np.abs(bw - ref)) < 2.2 * Rp * t32[2] + 1e-5


@pytest.mark.parametrize("B,T", [(70, 600), (256, 2048)])
def test_forward_vs_oracle(oracle, B, T):
    from wdf_hip import binding as wb, workload
    x = workload.sweep_batch(B, T, seed=B)
    t32 = THETA6.astype(np.float32).astype(np.float64)
    ref = oracle.clipper_asym_fwd(t32, FS, x.astype(np.float64))
    y, zT, it = wb.clipper_asym_fwd(dev(x), dev(THETA6), FS, wb.ASYM_NEWTON_F64, tol=1e-12, want_zT=True, want_iters=True)
    assert np.max(np.abs(y.cpu().numpy() - ref)) < 3e-6
    mean_iters = float(it.sum()) / (it.numel() * T)
    assert 1.0 <= mean_iters <= 12.0, mean_iters
    # looser tolerance -> fewer iterations (the ballot stops the wave earlier), still accurate
    y2, _, it2 = wb.clipper_asym_fwd(dev(x), dev(THETA6), FS, wb.ASYM_NEWTON_F64, tol=1e-6, want_iters=True)
    assert float(it2.sum()) <= float(it.sum())
    assert np.max(np.abs(y2.cpu().numpy() - ref)) < 3e-6
    yw, _, _ = wb.clipper_asym_fwd(dev(x), dev(THETA6), FS, wb.ASYM_OMEGA_F32)
    # MODEL error of the closed form, not rounding: eqn 39 drops the reverse diode's saturation
    # current (Rp Is_down = 4.2 mV at the root); the state recursion amplifies it by ~1/(1-0.9)
    assert np.max(np.abs(yw.cpu().numpy() - ref)) < 0.12
    # with identical diodes the omega mode IS the symmetric clipper kernel's root
    th_sym = THETA6.copy()
    th_sym[2:4] = th_sym[0:2]
    ys, _, _ = wb.clipper_asym_fwd(dev(x), dev(th_sym), FS, wb.ASYM_OMEGA_F32)
    yc, _, _ = wb.clipper_fwd(dev(x), dev(th_sym[[0, 1, 4, 5]]), FS, want_stash=False)
    assert float((ys - yc).abs().max()) < 2e-6


def test_reverse_sweep_vs_oracle_finite_differences(oracle):
    """dL/d{Is_up, nVt_up, Is_down, nVt_down, R, C} of L = sum(y gy) through the Newton-mode loop (implicit
    differentiation of the root, re-solved from the stash) against fp64 central differences of the oracle's
    forward (relative step 1e-6 per parameter): 2e-4 relative per component (observed <= 3e-5; the parameters
    themselves are fp32 on the device)."""
    from wdf_hip import binding as wb, engine, workload
    B, T = 70, 600
    x = workload.sweep_batch(B, T, seed=3)
    rng = np.random.default_rng(0)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    th = dev(THETA6).requires_grad_(True)
    y = engine.clipper_asym(th, dev(x), FS)
    (y * dev(gy)).sum().backward()
    got = th.grad.cpu().numpy().astype(np.float64)
    t32 = THETA6.astype(np.float32).astype(np.float64)
    x64, g64 = x.astype(np.float64), gy.astype(np.float64)
    ref = np.zeros(6)
    for i in range(6):
        h = 1e-6 * t32[i]
        tp, tm = t32.copy(), t32.copy()
        tp[i] += h
        tm[i] -= h
        ref[i] = (np.sum(oracle.clipper_asym_fwd(tp, FS, x64) * g64) - np.sum(oracle.clipper_asym_fwd(tm, FS, x64) * g64)) / (2 * h)
    err = np.abs(got - ref) / np.abs(ref)
    assert np.max(err) < 2e-4, (got, ref, err)
    # and the forward it differentiates is the plain Newton forward
    y2, _, _ = wb.clipper_asym_fwd(dev(x), dev(THETA6), FS, wb.ASYM_NEWTON_F64, tol=1e-12)
    assert torch.equal(y.detach(), y2)
    # the same gradient with the forward cut into time chunks (the stash then comes from the verified chunks)
    th2 = dev(THETA6).requires_grad_(True)
    y3 = engine.clipper_asym(th2, dev(x), FS, tp=engine.TpPlan(3, 192, 1e-6, 1))
    (y3 * dev(gy)).sum().backward()
    assert wb.mlp_tp_status(engine.LAST_TP_STATUS["status"])["n_bad"] == 0
    assert float((y3.detach() - y2).abs().max()) <= 1e-6
    assert np.max(np.abs(th2.grad.cpu().numpy() - got) / np.abs(got)) < 2e-5


@pytest.mark.parametrize("mode_name", ["newton", "omega"])
@pytest.mark.parametrize("B,T,K,W", [(70, 1000, 4, 192), (256, 2048, 8, 192), (64, 4096, 16, 192), (5, 130, 2, 64)])
def test_time_parallel_forward_equals_sequential(B, T, K, W, mode_name):
    """wdf_clipper_asym_fwd_tp: chunks warmed up from z = 0, verified on the device -> the sequential kernel's y, stash and
    final state within the verified tolerance, clean status; ragged B and T."""
    from wdf_hip import binding as wb, workload
    mode = wb.ASYM_NEWTON_F64 if mode_name == "newton" else wb.ASYM_OMEGA_F32
    x = dev(workload.sweep_batch(B, T, seed=B + T))
    th = dev(THETA6)
    z0 = dev(np.random.default_rng(B).uniform(-0.2, 0.2, B))
    y, zT, _, zs = wb.clipper_asym_fwd(x, th, FS, mode, tol=1e-12, max_iter=50, z0=z0, want_zT=True, want_stash=True)
    y2, zT2, zs2, st = wb.clipper_asym_fwd_tp(x, th, FS, mode, K, W, tol=1e-12, max_iter=50, z0=z0, want_zT=True, want_stash=True)
    s = wb.mlp_tp_status(st)
    assert s["n_bad"] == 0 and s["gated_waves"] == 0 and s["max_miss"] <= 1e-6, s
    assert float((y2 - y).abs().max()) <= 1e-6 and float((zs2 - zs).abs().max()) <= 2e-6
    assert float((zT2 - zT).abs().max()) <= 2e-6


def test_time_parallel_forward_repairs_a_short_warmup():
    """A warm-up of 8 steps cannot work (the RC network remembers ~150): the verification gates every wave and the gated
    sequential launch restores the sequential result."""
    from wdf_hip import binding as wb, workload
    B, T = 130, 2048
    x = dev(workload.sweep_batch(B, T, seed=3))
    th = dev(THETA6)
    y, _, _, zs = wb.clipper_asym_fwd(x, th, FS, wb.ASYM_NEWTON_F64, want_stash=True)
    y2, _, zs2, st = wb.clipper_asym_fwd_tp(x, th, FS, wb.ASYM_NEWTON_F64, 8, 8, want_stash=True)
    s = wb.mlp_tp_status(st)
    assert s["n_bad"] > 0 and s["gated_waves"] == 3, s
    assert torch.equal(y2, y) and torch.equal(zs2, zs)
