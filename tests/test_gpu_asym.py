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


@pytest.mark.parametrize("B,T,K", [(70, 600, 1), (70, 600, 5), (256, 2048, 16), (5, 131, 3), (64, 4096, 64)])
def test_time_parallel_reverse_sweep_equals_sequential(B, T, K):
    """wdf_clipper_asym_bwd_tp, Newton mode: no root re-solve (b from two consecutive stash entries), chunks composed
    exactly -> the sequential sweep's gradient (which re-solves the root by Newton at every step) to 2e-5 relative, for
    ragged B / T and chunk counts; dL/dz0 and an adjoint entering at the end (gzT) included."""
    from wdf_hip import binding as wb, workload
    x = dev(workload.sweep_batch(B, T, seed=B + T))
    th = dev(THETA6)
    rng = np.random.default_rng(K)
    gy = dev(rng.standard_normal((T, B)) / (B * T))
    y, zT, _, zs = wb.clipper_asym_fwd(x, th, FS, wb.ASYM_NEWTON_F64, tol=1e-12, want_zT=True, want_stash=True)
    g_seq = wb.clipper_asym_bwd(x, th, FS, zs, gy).cpu().numpy().astype(np.float64)
    g_tp, gz0 = wb.clipper_asym_bwd_tp(x, th, FS, wb.ASYM_NEWTON_F64, zs, zT, gy, K, want_gz0=True)
    g_tp = g_tp.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(g_tp - g_seq) / np.abs(g_seq)) < 2e-5, (g_tp, g_seq)
    assert bool(torch.isfinite(gz0).all())
    # linearity in the entering adjoint: sweep(gy, gzT) - sweep(gy, 0) = sweep(0, gzT), and dL/dz0 of a pure end-adjoint is
    # the product of the step multipliers (|kappa| < 1: it decays)
    gzT = dev(rng.standard_normal(B))
    g_a, z_a = wb.clipper_asym_bwd_tp(x, th, FS, wb.ASYM_NEWTON_F64, zs, zT, gy, K, gzT=gzT, want_gz0=True)
    g_b, z_b = wb.clipper_asym_bwd_tp(x, th, FS, wb.ASYM_NEWTON_F64, zs, zT, torch.zeros_like(gy), K, gzT=gzT, want_gz0=True)
    assert torch.allclose(g_a - dev(g_tp), g_b, rtol=1e-3, atol=1e-6 * float(g_b.abs().max()))
    assert torch.allclose(z_a - gz0, z_b, rtol=1e-3, atol=1e-9)
    if T >= 600:
        assert float(z_b.abs().max()) < 1e-6 * float(gzT.abs().max())


def _omega_mode_forward_numpy(theta6, fs, x):
    """The OMEGA-mode loop in numpy (scipy's Wright omega accepts complex arguments: complex-step derivatives)."""
    from scipy.special import wrightomega
    Is1, V1, Is2, V2, R, C = theta6
    G1, G2 = 1.0 / R, 2.0 * C * fs
    Rp, p = 1.0 / (G1 + G2), G1 / (G1 + G2)
    l1, l2 = np.log(Rp * Is1 / V1), np.log(Rp * Is2 / V2)
    B, T = x.shape
    z = np.zeros(B, dtype=complex)
    y = np.zeros((T, B), dtype=complex)
    for t in range(T):
        bd = z - x[:, t]
        a = z - p * bd
        pos = a.real >= 0
        lam = np.where(pos, 1.0, -1.0)
        aa = lam * a
        Vf, Vr = np.where(pos, V1, V2), np.where(pos, V2, V1)
        lf, lr = np.where(pos, l1, l2), np.where(pos, l2, l1)
        b = a - 2.0 * lam * (Vf * wrightomega(lf + aa / Vf) - Vr * wrightomega(lr - aa / Vr))
        zn = b - p * bd
        y[t] = 0.5 * (zn + z)
        z = zn
    return y


def test_omega_mode_reverse_sweep_against_complex_step():
    """OMEGA mode differentiates the closed form its forward evaluates: dL/dtheta6 of L = sum(y gy) against the complex-step
    derivative of a numpy restatement of that loop (scipy.special.wrightomega on complex arguments), 2e-4 relative per
    component; the engine's autograd function in OMEGA mode gives the same numbers."""
    from wdf_hip import binding as wb, engine, workload
    B, T = 48, 500
    x = workload.sweep_batch(B, T, seed=11)
    rng = np.random.default_rng(2)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    t32 = THETA6.astype(np.float32).astype(np.float64)
    y_ref = _omega_mode_forward_numpy(t32, FS, x.astype(np.float64)).real
    yw, zT, _, zs = wb.clipper_asym_fwd(dev(x), dev(THETA6), FS, wb.ASYM_OMEGA_F32, want_zT=True, want_stash=True)
    assert np.max(np.abs(yw.cpu().numpy() - y_ref)) < 5e-6
    ref = np.zeros(6)
    for i in range(6):
        tc = t32.astype(complex)
        h = 1e-20 * t32[i]
        tc[i] += 1j * h
        ref[i] = np.sum(_omega_mode_forward_numpy(tc, FS, x.astype(np.float64)).imag * gy) / h
    for K in (1, 4):
        got = wb.clipper_asym_bwd_tp(dev(x), dev(THETA6), FS, wb.ASYM_OMEGA_F32, zs, zT, dev(gy), K).cpu().numpy().astype(np.float64)
        err = np.abs(got - ref) / np.abs(ref)
        assert np.max(err) < 2e-4, (K, got, ref, err)
    th = dev(THETA6).requires_grad_(True)
    y = engine.clipper_asym(th, dev(x), FS, mode=wb.ASYM_OMEGA_F32)
    (y * dev(gy)).sum().backward()
    assert np.max(np.abs(th.grad.cpu().numpy() - ref) / np.abs(ref)) < 2e-4
