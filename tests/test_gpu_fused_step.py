"""GPU: the one-pass training step (wdf_clipper_step_mse_tp, csrc/wdf_clipper_fused.h) gives what the
two-kernel step gives -- the forward's y and the reverse sweep's gradient -- and what the fp64 oracle gives.

The gradient is carried forward in time (tangent of the state) instead of swept backward: same sums, other
order.  Both forms of the kernel run every test: two adjacent sequences per lane with packed arithmetic, and one.  Covered: chunk counts incl. 1, ragged B and T, asymmetric diode counts, the per-sample resistance
channel, both x layouts, loss masks (skip), an initial state, the warm-started loop, the repair path (a
parameter jump and a circuit whose memory outlasts the warm-up), the Adam update folded into the launch,
and the bench shape against the oracle.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000.0
Y_TOL = 2.0e-6        # volts, vs the sequential kernel / the fp64 oracle (measured ~1e-7)
G_RTOL = 1.0e-4       # per gradient component (measured ~3e-6)


@pytest.fixture(scope="module")
def wb():
    from wdf_hip import binding
    binding.require_gpu()
    return binding


@pytest.fixture(autouse=True, params=["two sequences per lane", "one sequence per lane"])
def lanes(request, wb):
    """Every test runs both forms of the kernel: a lane owning two adjacent sequences with packed fp32 arithmetic (the
    default for an even batch) and one sequence per lane (odd batches; forced here with WDF_ONE_SEQUENCE_PER_LANE)."""
    wb.ONE_SEQUENCE_PER_LANE = request.param.startswith("one")
    yield request.param
    wb.ONE_SEQUENCE_PER_LANE = False


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def problem(B, T, seed=0):
    from wdf_hip import workload
    x = workload.sweep_batch(B, T, seed=seed)
    th = workload.clipper_theta()
    return x, th, workload.target_theta()


def two_kernel_step(wb, x, th, tgt, gscale, r=None, n_up=1, n_down=1, skip=0, z0=None):
    """forward + reverse sweep with dL/dy = gscale (y - target) past skip: the reference for the fused step"""
    y, zs, _ = wb.clipper_fwd(x, th, FS, r=r, n_up=n_up, n_down=n_down, z0=z0)
    gy = gscale * (y - tgt)
    gy[:skip] = 0.0
    g, _ = wb.clipper_bwd(x, th, FS, zs, gy.contiguous(), r=r, n_up=n_up, n_down=n_down)
    sse = ((y - tgt)[skip:] ** 2).double().sum()
    return y, g, float(sse)


def close_grad(g, g_ref, rtol=G_RTOL):
    g, g_ref = g.double().cpu().numpy(), g_ref.double().cpu().numpy()
    return np.all(np.abs(g - g_ref) <= rtol * np.abs(g_ref) + 1e-30), (g, g_ref)


@pytest.mark.parametrize("B,T,K,W", [(64, 512, 1, 0), (64, 2048, 4, 256), (70, 1001, 3, 248), (130, 4096, 16, 256),
                                     (5, 96, 1, 0), (1, 2048, 4, 256), (3, 40, 1, 0), (2, 64, 2, 32), (258, 1000, 2, 256)])
@pytest.mark.parametrize("n_up,n_down", [(1, 1), (2, 3)])
def test_fused_matches_two_kernel_step(wb, B, T, K, W, n_up, n_down):
    x, th, ths = problem(B, T, seed=B + T)
    xd, thd = dev(x), dev(th)
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, n_up=n_up, n_down=n_down, want_stash=False)
    gscale = 2.0 / (B * T)
    y_ref, g_ref, sse_ref = two_kernel_step(wb, xd, thd, tgt, gscale, n_up=n_up, n_down=n_down)
    y, _, g, sse, st = wb.clipper_step_mse_tp(xd, thd, FS, tgt, gscale, K, W, n_up=n_up, n_down=n_down)
    s = wb.tp_status(st)
    assert s["n_bad"] == 0 and not s["fallback_ran"], s
    assert float((y - y_ref).abs().max()) <= Y_TOL
    ok, info = close_grad(g, g_ref)
    assert ok, info
    assert abs(float(sse) - sse_ref) <= 2e-5 * sse_ref
    # time-major x: same numbers
    y2, _, g2, sse2, _ = wb.clipper_step_mse_tp(xd.t().contiguous(), thd, FS, tgt, gscale, K, W, n_up=n_up, n_down=n_down,
                                                time_major=True)
    assert float((y2 - y).abs().max()) <= 1e-6
    ok, info = close_grad(g2, g, rtol=2e-5)
    assert ok, info
    # deterministic
    y3, _, g3, sse3, _ = wb.clipper_step_mse_tp(xd, thd, FS, tgt, gscale, K, W, n_up=n_up, n_down=n_down)
    assert torch.equal(y3, y) and torch.equal(g3, g) and torch.equal(sse3, sse)


def test_fused_vs_oracle_f64(wb, oracle):
    B, T, K, W = 96, 2048, 8, 256
    x, th, ths = problem(B, T, seed=21)
    tgt64 = oracle.clipper_fwd(ths.astype(np.float64), FS, x.astype(np.float64))
    th64 = th.astype(np.float32).astype(np.float64)
    loss_ref, g_ref, y_ref = oracle.clipper_mse_step(th64, FS, x.astype(np.float64), tgt64.astype(np.float32).astype(np.float64),
                                                     dtype=np.float64)
    y, _, g, sse, st = wb.clipper_step_mse_tp(dev(x), dev(th), FS, dev(tgt64), 2.0 / (B * T), K, W)
    assert wb.tp_status(st)["n_bad"] == 0
    assert float(np.max(np.abs(y.cpu().numpy() - y_ref))) <= Y_TOL
    got = g.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got - g_ref) <= G_RTOL * np.abs(g_ref)), (got, g_ref)
    assert abs(float(sse) / (B * T) - loss_ref) <= 1e-5 * loss_ref


@pytest.mark.parametrize("is_scale,tier", [(1.0, "lean"), (2.9, "lean"), (3.05, "fast"), (1.0e-3, "lean"), (40.0, "fast")])
def test_root_tiers_against_the_oracle_across_the_tier_boundary(wb, oracle, is_scale, tier):
    """The LEAN root tier (round 6: csrc/wdf_omega.h omega_lean, omega_1 = p (1 - p); taken when log(Rp Is / nVt) <= -7.5,
    i.e. Is up to 2.97 x the 1N4148's behind this circuit) and the FAST tier above it: y, the loss and the four gradient
    components of the one-pass step against the fp64 oracle on both sides of the boundary and deep inside (Is / 1000: omega_0's
    arguments down to -15.5, where the FSC step now runs although the series alone would do), and the sequential forward
    kernel against the same trajectory.  `tier` names what csrc/wdf_clipper.h root_tier picks for the row (asserted from L)."""
    from wdf_hip import workload
    B, T, K, W = 96, 2048, 8, 256
    x, th, ths = problem(B, T, seed=23)
    th = th.copy()
    th[0] *= is_scale
    ths = ths.copy()
    ths[0] *= is_scale
    th32 = th.astype(np.float32).astype(np.float64)
    Rp = 1.0 / (1.0 / th32[2] + 2.0 * th32[3] * FS)
    L = np.log(Rp * th32[0] / th32[1])
    assert (L <= -7.5) == (tier == "lean"), L
    tgt64 = oracle.clipper_fwd(ths.astype(np.float64), FS, x.astype(np.float64))
    loss_ref, g_ref, y_ref = oracle.clipper_mse_step(th32, FS, x.astype(np.float64), tgt64.astype(np.float32).astype(np.float64),
                                                     dtype=np.float64)
    y, _, g, sse, st = wb.clipper_step_mse_tp(dev(x), dev(th), FS, dev(tgt64), 2.0 / (B * T), K, W)
    assert wb.tp_status(st)["n_bad"] == 0
    e_y = float(np.max(np.abs(y.cpu().numpy() - y_ref)))
    got = g.cpu().numpy().astype(np.float64)
    e_g = float(np.max(np.abs(got - g_ref) / np.abs(g_ref)))
    ys, _, _ = wb.clipper_fwd(dev(x), dev(th), FS, want_stash=False)
    e_s = float(np.max(np.abs(ys.cpu().numpy() - y_ref)))
    print(f"Is x {is_scale}: L {L:.3f} ({tier}); |y - oracle| {e_y:.2e} (sequential kernel {e_s:.2e}), gradient {e_g:.2e}")
    assert e_y <= Y_TOL and e_s <= Y_TOL
    assert e_g <= G_RTOL, (got, g_ref)
    assert abs(float(sse) / (B * T) - loss_ref) <= 1e-5 * loss_ref
    # and against the general root (ballot path, five-term series, conditional FSC step): the tiers agree to fp32 rounding
    wb.GENERAL_ROOT = True
    try:
        yg, _, _ = wb.clipper_fwd(dev(x), dev(th), FS, want_stash=False)
    finally:
        wb.GENERAL_ROOT = False
    assert float((yg - ys).abs().max()) <= 3e-7


def test_fused_per_sample_resistance_and_skip(wb):
    B, T, K, W, skip = 71, 1024, 2, 256, 50
    x, th, ths = problem(B, T, seed=3)
    xd, thd = dev(x), dev(th)
    r = dev(45.0e3 * np.exp(0.8 * np.sin(np.arange(T)[None, :] * 0.01 * (1 + np.arange(B)[:, None] % 5))))
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, r=r, want_stash=False)
    gscale = 2.0 / (B * (T - skip))
    y_ref, g_ref, sse_ref = two_kernel_step(wb, xd, thd, tgt, gscale, r=r, skip=skip)
    y, _, g, sse, st = wb.clipper_step_mse_tp(xd, thd, FS, tgt, gscale, K, W, r=r, skip=skip)
    assert wb.tp_status(st)["n_bad"] == 0
    assert float((y - y_ref).abs().max()) <= Y_TOL
    ok, info = close_grad(g[[0, 1, 3]], g_ref[[0, 1, 3]])            # dL/dR is identically 0 with a streamed resistance
    assert ok and float(g[2]) == 0.0, info
    assert abs(float(sse) - sse_ref) <= 2e-5 * sse_ref
    yt, _, gt, _, _ = wb.clipper_step_mse_tp(xd.t().contiguous(), thd, FS, tgt, gscale, K, W, r=r.t().contiguous(), skip=skip,
                                             time_major=True)
    assert float((yt - y).abs().max()) <= 1e-6
    ok, info = close_grad(gt[[0, 1, 3]], g[[0, 1, 3]], rtol=2e-5)
    assert ok, info


@pytest.mark.parametrize("time_major", [True, False])
@pytest.mark.parametrize("n_up,n_down", [(1, 1), (1, 2)])
def test_fused_one_pot_value_per_sequence(wb, oracle, time_major, n_up, n_down):
    """The reference's recordings hold ONE pot value per file (dataimport.py:96; batch_data cuts the sequences out of it): a
    resistance channel that is constant along every sequence.  The binding notices (r_is_per_sequence: one comparison per
    tensor) and the one-pass step evaluates calc_impedance once per chunk (WDF_R_PER_SEQUENCE: DYN_R = 2 in the kernels; the
    symmetric pair then takes the LEAN root) instead of every step.  Against the per-sample evaluation of the same data (the
    flag switched off) and against the fp64 oracle: y, SSE, gradients; MSE and MSE + ESR; a warm-started second call; a
    channel that moves within a sequence keeps the per-sample path."""
    from wdf_hip import workload
    B, T, K, W, skip = 200, 2048, 8, 448, 50
    x, th, ths = problem(B, T, seed=13)
    r_host = workload.dataset_resistance_batch(B, T)                       # four blocks of sequences, one pot value each
    xd, rd, thd = dev(x), dev(r_host), dev(th)
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, r=rd, n_up=n_up, n_down=n_down, want_stash=False)
    xin, rin = (xd.t().contiguous(), rd.t().contiguous()) if time_major else (xd, rd)
    assert wb.r_is_per_sequence(rin, time_major)
    gscale = 2.0 / (B * T)
    out = {}
    for per_seq in (True, False):
        wb.R_PER_SEQUENCE = per_seq
        try:
            y, _, g, sse, st = wb.clipper_step_mse_tp(xin, thd, FS, tgt, gscale, K, W, r=rin, n_up=n_up, n_down=n_down, time_major=time_major)
            assert wb.tp_status(st)["n_bad"] == 0
            ye, _, s10, ge, l3, _ = wb.clipper_step_esr_tp(xin, thd, FS, tgt, float(B * (T - skip)), 2.2e-16, skip, K, W, r=rin, n_up=n_up,
                                                          n_down=n_down, time_major=time_major)
            out[per_seq] = (y.clone(), g.clone(), float(sse), ye.clone(), ge.clone(), l3.clone())
        finally:
            wb.R_PER_SEQUENCE = True
    a, b = out[True], out[False]
    assert float((a[0] - b[0]).abs().max()) <= 3e-7 and float((a[3] - b[3]).abs().max()) <= 3e-7
    ok, info = close_grad(a[1][[0, 1, 3]], b[1][[0, 1, 3]], rtol=2e-5)
    assert ok and float(a[1][2]) == 0.0, info
    ok, info = close_grad(a[4][[0, 1, 3]], b[4][[0, 1, 3]], rtol=2e-5)
    assert ok, info
    assert abs(a[2] - b[2]) <= 1e-5 * b[2] and float((a[5] - b[5]).abs().max()) <= 1e-5 * float(b[5].abs().max())
    # the oracle: y and the MSE gradient
    th64 = th.astype(np.float32).astype(np.float64)
    t64 = tgt.cpu().numpy().astype(np.float64)
    y64 = oracle.clipper_fwd(th64, FS, x.astype(np.float64), r=r_host.astype(np.float64), n_up=n_up, n_down=n_down)
    _, g64 = oracle.clipper_fwd_bwd(th64, FS, x.astype(np.float64), 2.0 * (y64 - t64) / (B * T), r=r_host.astype(np.float64), n_up=n_up,
                                    n_down=n_down)
    assert float(np.max(np.abs(a[0].cpu().numpy() - y64))) <= Y_TOL
    got = a[1].cpu().numpy().astype(np.float64)
    assert all(abs(got[i] - g64[i]) <= G_RTOL * abs(g64[i]) for i in (0, 1, 3)), (got, g64)
    # a pot that moves within a sequence is not taken for a constant one
    wob = rin.clone()
    if time_major:
        wob[T // 2:, 3] *= 1.01
    else:
        wob[3, T // 2:] *= 1.01
    assert not wb.r_is_per_sequence(wob, time_major)


def test_fused_initial_state_general_root_and_accumulate(wb):
    B, T, K, W = 64, 1024, 4, 256
    x, th, ths = problem(B, T, seed=2)
    xd, thd = dev(x), dev(th)
    z0 = dev(np.random.default_rng(0).uniform(-0.3, 0.3, B))
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, want_stash=False)
    gscale = 2.0 / (B * T)
    y_ref, g_ref, _ = two_kernel_step(wb, xd, thd, tgt, gscale, z0=z0)
    y, zT, g, _, st = wb.clipper_step_mse_tp(xd, thd, FS, tgt, gscale, K, W, z0=z0, want_zT=True)
    assert wb.tp_status(st)["n_bad"] == 0
    assert float((y - y_ref).abs().max()) <= Y_TOL
    ok, info = close_grad(g, g_ref)
    assert ok, info
    _, _, zT_ref = wb.clipper_fwd(xd, thd, FS, z0=z0, want_stash=False, want_zT=True)
    assert float((zT - zT_ref).abs().max()) <= Y_TOL
    # the general per-step root evaluation gives the same step
    wb.GENERAL_ROOT = True
    try:
        y_g, _, g_g, _, _ = wb.clipper_step_mse_tp(xd, thd, FS, tgt, gscale, K, W, z0=z0)
    finally:
        wb.GENERAL_ROOT = False
    assert float((y_g - y).abs().max()) <= 1e-6
    ok, info = close_grad(g_g, g, rtol=2e-5)
    assert ok, info
    # accumulate: gtheta += gradient
    acc = g.clone()
    wb.clipper_step_mse_tp(xd, thd, FS, tgt, gscale, K, W, z0=z0, gtheta=acc, accumulate=True)
    assert torch.allclose(acc, 2.0 * g, rtol=1e-6, atol=0)


def test_fused_repairs_when_warmup_is_too_short(wb):
    """C = 1 uF: the circuit remembers ~4000 samples, a 64-step warm-up cannot work.  Every boundary must be
    caught, the chunks re-run from the exact state (outputs AND tangent records), and the step must equal the
    sequential computation."""
    from wdf_hip import workload
    B, T, K = 70, 2048, 8
    x = workload.sweep_batch(B, T, seed=11)
    theta = workload.clipper_theta()
    theta[3] = 1.0e-6
    ths = workload.target_theta()
    ths[3] = 1.1e-6
    xd, thd = dev(x), dev(theta)
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, want_stash=False)
    gscale = 2.0 / (B * T)
    y_ref, g_ref, sse_ref = two_kernel_step(wb, xd, thd, tgt, gscale)
    y, _, g, sse, st = wb.clipper_step_mse_tp(xd, thd, FS, tgt, gscale, K, 64)
    s = wb.tp_status(st)
    assert s["n_bad"] > 0 and s["repaired_tiles"] == 2 * 7, s     # 7 boundaries x (2 waves | 2 interleaved halves of one wave)
    assert float((y - y_ref).abs().max()) <= Y_TOL
    ok, info = close_grad(g, g_ref, rtol=5e-5)
    assert ok, info
    assert abs(float(sse) - sse_ref) <= 2e-5 * sse_ref
    # and the tickets were left clean: the same workspace serves a good call afterwards
    ws = wb.step_mse_workspace(B, K, xd.device)
    for W in (64, 64, 4096):
        y2, _, g2, _, st2 = wb.clipper_step_mse_tp(xd, thd, FS, tgt, gscale, K, W, ws=ws)
        assert float((y2 - y_ref).abs().max()) <= Y_TOL
        ok, info = close_grad(g2, g_ref, rtol=5e-5)
        assert ok, info


def test_fused_warm_started_training_loop_with_adam(wb):
    """The bench loop in small: resident x, Adam folded into the launch, warm-started chunks.  Every step is
    compared with the two-kernel step at the SAME theta; then theta jumps (repair), then stands still."""
    from wdf_hip import workload
    B, T, K, W = 128, 4096, 16, 256
    x, th0, ths = problem(B, T, seed=5)
    xd = dev(x)
    xt = xd.t().contiguous()
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, want_stash=False)
    gscale = 2.0 / (B * T)
    theta = dev(th0)
    state = wb.TpWarmState(B, T, K, 256 // wb.warm_unit(), xd.device)
    ws = wb.step_mse_workspace(B, K, xd.device)
    lr = [1e-3 * float(v) for v in th0]
    lo, hi = [1e-15, 1e-3, 180.0, 1e-13], [1e-3, 1.0, 1.0e6, 1.0]
    opt = wb.Adam(4, lr=lr, lo=lo, hi=hi, device=xd.device)
    opt_ref = wb.Adam(4, lr=lr, lo=lo, hi=hi, device=xd.device)
    theta_ref = theta.clone()
    losses = []
    for it in range(12):
        if it == 8:                                                  # a jump the snapshots cannot follow
            theta *= torch.tensor([1.5, 1.1, 0.6, 1.6], device="cuda")
            theta_ref.copy_(theta)
        th_before = theta.clone()
        y, _, g, sse, st = wb.clipper_step_mse_tp(xt, theta, FS, tgt, gscale, K, W, ws=ws, state=state, opt=opt, time_major=True)
        s = wb.tp_status(st)
        y_ref, g_ref, sse_ref = two_kernel_step(wb, xd, th_before, tgt, gscale)
        assert float((y - y_ref).abs().max()) <= Y_TOL, (it, s)
        ok, info = close_grad(g, g_ref)
        assert ok, (it, info)
        opt_ref.apply(theta_ref, g.clone())
        assert torch.allclose(theta, theta_ref, rtol=1e-6, atol=0), (it, theta, theta_ref)   # the folded update is wdf_adam_step's
        theta_ref.copy_(theta)
        if it == 8:
            assert s["n_bad"] > 0 and s["repaired_tiles"] > 0, s
        elif it >= 2:
            assert s["n_bad"] == 0, (it, s)
        losses.append(float(sse))
    assert losses[7] < losses[0]
    info = state.info()
    assert info["valid"] == 3 and info["n_calls"] == 12
    # theta stands still: plain warm start from the snapshots, zero miss
    for _ in range(3):
        y, _, g, _, st = wb.clipper_step_mse_tp(xt, theta, FS, tgt, gscale, K, W, ws=ws, state=state, time_major=True)
    assert wb.tp_status(st)["n_bad"] == 0


def test_fused_bench_shape_against_oracle(wb, oracle):
    """BASELINE configs[2] at full size through engine.MseStep.step_fused (what bench.py times): 16 sequences of y and
    the whole-batch gradient against the fp64 oracle."""
    from wdf_hip import engine, workload
    B, T = 8192, 4096
    x = workload.sweep_batch(B, T)
    th, ths = workload.clipper_theta(), workload.target_theta()
    xd = dev(x)
    xt = xd.t().contiguous()
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, want_stash=False)
    plan = engine.plan_time_parallel(B, T, th[2], th[3], FS, time_major=True)
    st = engine.MseStep(B, T, FS, plan, xd.device, time_major=True, warm=True)
    theta = dev(th)
    for _ in range(3):
        sse, g = st.step_fused(theta, xt, tgt)
    stat = wb.tp_status(st.status)
    assert stat["n_bad"] == 0, stat
    pick = np.sort(np.random.default_rng(5).choice(B, 16, replace=False))
    y_ref = oracle.clipper_fwd(th.astype(np.float32).astype(np.float64), FS, x[pick].astype(np.float64))
    assert float(np.max(np.abs(st.y[:, pick].cpu().numpy() - y_ref))) <= Y_TOL
    loss_ref, g_ref, _ = oracle.clipper_mse_step(th.astype(np.float32).astype(np.float64), FS, x.astype(np.float64),
                                                 tgt.cpu().numpy().astype(np.float64), dtype=np.float64)
    got = g.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got - g_ref) <= G_RTOL * np.abs(g_ref)), (got, g_ref)
    assert abs(float(sse) / (B * T) - loss_ref) <= 1e-5 * loss_ref


def _esr_reference(wb, xd, thd, tgt, skip, r=None, n_up=1, n_down=1):
    """MSE + ESR past skip (clipper_pot.py:146-156,177 with the scripts' argument order) through the kernel pair + torch
    autograd: the loss values and d loss / d{Is, nVt, R, C}."""
    from wdf_hip import engine
    th = thd.clone().requires_grad_(True)
    y = engine.clipper(th, xd, FS, r=r, n_up=n_up, n_down=n_down)
    o, t = y[skip:].double(), tgt[skip:].double()
    S, E = ((o - t) ** 2).sum(), (o ** 2).sum()
    n = o.numel()
    mse, esr = S / n, torch.sqrt(S / (E + float(np.finfo(float).eps)) / n)
    (mse + esr).backward()
    return y.detach(), th.grad.detach(), float(mse.detach()), float(esr.detach()), float(S.detach()), float(E.detach())


@pytest.mark.parametrize("B,T,K,W,with_r", [(64, 1024, 2, 256, False), (70, 1000, 3, 248, False), (130, 2048, 8, 256, False),
                                            (71, 1024, 2, 256, True), (96, 2048, 4, 512, True)])
def test_fused_esr_step_matches_autograd(wb, oracle, B, T, K, W, with_r):
    """The scripts' training loss in one pass: both tangent-weighted sums carried, coefficients applied by the last tile.
    Held against torch autograd through the kernel pair AND, directly, against the fp64 oracle (y, the three loss values
    and every gradient component: clipper_pot.py:146-156,177 evaluated on the oracle's own y)."""
    skip = 50
    x, th, ths = problem(B, T, seed=B + T + 1)
    xd, thd = dev(x), dev(th)
    r = dev(45.0e3 * np.exp(0.8 * np.sin(np.arange(T)[None, :] * 0.01 * (1 + np.arange(B)[:, None] % 5)))) if with_r else None
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, r=r, want_stash=False)
    y_ref, g_ref, mse, esr, S, E = _esr_reference(wb, xd, thd, tgt, skip, r=r)
    n = B * (T - skip)
    eps = float(np.finfo(float).eps)
    y, _, sums10, g, loss3, st = wb.clipper_step_esr_tp(xd, thd, FS, tgt, n, eps, skip, K, W, r=r)
    assert wb.tp_status(st)["n_bad"] == 0
    assert float((y - y_ref).abs().max()) <= Y_TOL
    idx = [0, 1, 3] if with_r else [0, 1, 2, 3]
    ok, info = close_grad(g[idx], g_ref[idx])
    assert ok, info
    if with_r:
        assert float(g[2]) == 0.0
    l = loss3.cpu().numpy()
    assert abs(l[0] - mse) <= 2e-5 * mse and abs(l[1] - esr) <= 2e-5 * esr and abs(l[2] - (mse + esr)) <= 2e-5 * (mse + esr)
    s10 = sums10.cpu().numpy()
    assert abs(s10[0] - S) <= 2e-5 * S and abs(s10[1] - E) <= 2e-5 * E
    # ... and the oracle itself, nothing of ours in between: fp64 forward, the loss of the scripts on ITS y, dLoss/dy =
    # ga (y - t) + gb y past skip, the oracle's reverse sweep
    th64 = th.astype(np.float32).astype(np.float64)
    r64 = None if r is None else r.cpu().numpy().astype(np.float64)
    t64 = tgt.cpu().numpy().astype(np.float64)
    y64 = oracle.clipper_fwd(th64, FS, x.astype(np.float64), r=r64)
    o64, tt64 = y64[skip:], t64[skip:]
    S64, E64 = float(np.sum((o64 - tt64) ** 2)), float(np.sum(o64 ** 2)) + eps
    mse64, esr64 = S64 / n, float(np.sqrt(S64 / E64 / n))
    gy64 = np.zeros_like(y64)
    gy64[skip:] = (2.0 / n + 1.0 / (esr64 * E64 * n)) * (o64 - tt64) - (esr64 / E64) * o64
    _, g64 = oracle.clipper_fwd_bwd(th64, FS, x.astype(np.float64), gy64, r=r64)
    assert float(np.max(np.abs(y.cpu().numpy() - y64))) <= Y_TOL
    ok, info = close_grad(g[idx], torch.as_tensor(g64[idx]))
    assert ok, info
    assert abs(l[0] - mse64) <= 2e-5 * mse64 and abs(l[1] - esr64) <= 2e-5 * esr64
    # the two-stage form (what several ranks run: sums out, all-reduce, finish) gives the same numbers
    _, _, sums_b, g_none, l_none, _ = wb.clipper_step_esr_tp(xd, thd, FS, tgt, n, eps, skip, K, W, r=r, finish=False)
    assert g_none is None and l_none is None and torch.equal(sums_b, sums10)
    g2, l2 = wb.esr_finish(sums_b, n, eps)
    assert torch.allclose(g2, g, rtol=1e-6, atol=0) and torch.allclose(l2, loss3, rtol=1e-6, atol=0)
    # time-major inputs
    yt, _, _, gt, lt, _ = wb.clipper_step_esr_tp(xd.t().contiguous(), thd, FS, tgt, n, eps, skip, K, W,
                                                 r=None if r is None else r.t().contiguous(), time_major=True)
    assert float((yt - y).abs().max()) <= 1e-6
    ok, info = close_grad(gt[idx], g[idx], rtol=2e-5)
    assert ok, info


def test_fused_esr_engine_step_with_adam_and_warm_start(wb):
    """engine.MseStep(loss="mse+esr").step_fused: the bench's --loss mse+esr step; against the kernel-pair stepper at the
    same parameters, Adam folded into the launch, warm-started chunks."""
    from wdf_hip import engine
    B, T, skip = 256, 4096, 50
    x, th0, ths = problem(B, T, seed=9)
    xd = dev(x)
    xt = xd.t().contiguous()
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, want_stash=False)
    plan = engine.TpPlan(16, 192, 1e-6, 16)
    one = engine.MseStep(B, T, FS, plan, xd.device, time_major=True, loss="mse+esr", skip=skip, warm=True)
    pair = engine.MseStep(B, T, FS, plan, xd.device, time_major=True, loss="mse+esr", skip=skip)
    theta = dev(th0)
    lr = [1e-3 * float(v) for v in th0]
    opt = wb.Adam(4, lr=lr, lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=xd.device)
    opt_ref = wb.Adam(4, lr=lr, lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0], device=xd.device)
    theta_ref = theta.clone()
    first = None
    for it in range(8):
        th_before = theta.clone()
        one.step_fused(theta, xt, tgt, adam=opt)
        pair.forward(th_before, xt)
        _, g_ref = pair.backward(th_before, xt, tgt)
        assert float((one.y - pair.y).abs().max()) <= Y_TOL, it
        ok, info = close_grad(one.gtheta, g_ref)
        assert ok, (it, info)
        assert torch.allclose(one.loss, pair.loss, rtol=2e-5, atol=0), (it, one.loss, pair.loss)
        opt_ref.apply(theta_ref, one.gtheta.clone())
        assert torch.allclose(theta, theta_ref, rtol=1e-6, atol=0)
        theta_ref.copy_(theta)
        if it >= 2:
            assert wb.tp_status(one.status)["n_bad"] == 0
        first = float(one.loss[2]) if first is None else first
    assert float(one.loss[2]) < first


def test_fused_randomized_plans_and_circuits(wb):
    """tools/stress_tp.py's random cases through the one-pass step: random component values over the clip ranges of
    tf_wdf.py:74,104, diode parameters and counts, amplitudes, shapes, layouts, chunkings, warm-ups (hopeless ones
    included), warm starts from nearby or far-away parameters, one or two sequences per lane.  y must equal the
    sequential forward to 2e-6 whatever the plan (repairs included), the tangent-carried gradient must agree with the
    sequential reverse sweep."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_tp
    worst_y = worst_g = worst_s = 0.0
    repaired = 0
    for case in range(60):
        ey, eg, es, rep = stress_tp.run_case_one_pass(11, case)
        worst_y, worst_g, worst_s, repaired = max(worst_y, ey), max(worst_g, eg), max(worst_s, es), repaired + rep
    assert worst_y <= 2e-6 and worst_g <= 1e-3 and worst_s <= 1e-4, (worst_y, worst_g, worst_s)
    assert repaired >= 5


@pytest.mark.parametrize("B,T,K,W", [(130, 4096, 16, 256), (129, 3000, 47, 192), (64, 64, 1, 0), (300, 4000, 25, 160), (2, 2048, 8, 192),
                                     (1024, 1056, 11, 96)])
def test_fused_batch_major_windows_match_time_major(wb, B, T, K, W):
    """x as the caller holds it, [B, T]: whole 128-byte lines through LDS, the two halves of a wave's rows 16 steps apart
    (XWindow, wdf_clipper_fused.h).  Ragged last waves (130 = a wave of two rows, 129 / 300: one sequence per lane), windows
    across the sequence's end (3000, 4000: T % 32 != 0), a single chunk, and a warm-started loop whose warm-up settles at an
    odd number of 16-step units (a chunk then starts 16 steps into a line).  The steps are the time-major call's: y bit for
    bit, the sums to rounding."""
    x, th0, ths = problem(B, T, seed=B + T)
    xd = dev(x)
    xt = xd.t().contiguous()
    tgt, _, _ = wb.clipper_fwd(xd, dev(ths), FS, want_stash=False)
    gscale = 2.0 / (B * T)
    K = wb.lib().wdf_clipper_tp_chunks(T, K)
    for r in (None, dev(np.linspace(0.4, 1.6, B, dtype=np.float32)[:, None].repeat(T, 1))):     # static R; one pot value per sequence
        rt = None if r is None else r.t().contiguous()
        y_b, _, g_b, sse_b, st_b = wb.clipper_step_mse_tp(xd, dev(th0), FS, tgt, gscale, K, W, r=r, time_major=False)
        y_t, _, g_t, sse_t, st_t = wb.clipper_step_mse_tp(xt, dev(th0), FS, tgt, gscale, K, W, r=rt, time_major=True)
        assert torch.equal(y_b, y_t), float((y_b - y_t).abs().max())
        assert wb.tp_status(st_b)["n_bad"] == wb.tp_status(st_t)["n_bad"]
        ok, info = close_grad(g_b, g_t, rtol=1e-6)
        assert ok, info
        assert abs(float(sse_b) - float(sse_t)) <= 1e-6 * float(sse_t)
    if K < 4:
        return
    state_b = wb.TpWarmState(B, T, K, max(1, W // wb.warm_unit()), xd.device)
    state_t = wb.TpWarmState(B, T, K, max(1, W // wb.warm_unit()), xd.device)
    th_b, th_t = dev(th0), dev(th0)
    lr = [2e-4 * float(v) for v in th0]
    lo, hi = [1e-15, 1e-3, 180.0, 1e-13], [1e-3, 1.0, 1.0e6, 1.0]
    opt_b, opt_t = wb.Adam(4, lr=lr, lo=lo, hi=hi, device=xd.device), wb.Adam(4, lr=lr, lo=lo, hi=hi, device=xd.device)
    units = set()
    for it in range(10):
        y_b, _, g_b, _, st_b = wb.clipper_step_mse_tp(xd, th_b, FS, tgt, gscale, K, W, state=state_b, opt=opt_b, time_major=False)
        y_t, _, g_t, _, st_t = wb.clipper_step_mse_tp(xt, th_t, FS, tgt, gscale, K, W, state=state_t, opt=opt_t, time_major=True)
        assert torch.equal(y_b, y_t), (it, float((y_b - y_t).abs().max()))
        assert torch.allclose(th_b, th_t, rtol=1e-6, atol=0), (it, th_b, th_t)
        th_t.copy_(th_b)
        units.add(state_b.info()["last_warm_tiles"])
        assert state_b.info()["last_warm_tiles"] == state_t.info()["last_warm_tiles"]
    assert any(u % 2 == 1 for u in units), units                # a chunk started 16 steps into a line


def test_fused_step_is_bit_reproducible(wb):
    """Which wave finishes a tile / the step varies from launch to launch; what they compute does not (records and partials
    are re-read in index order, fixed-order reductions): 100 launches, bit-identical y, gradient, SSE and status."""
    B, T, K, W = 1024, 2048, 16, 256
    x, th, ths = problem(B, T, seed=77)
    xd, thd = dev(x).t().contiguous(), dev(th)
    tgt, _, _ = wb.clipper_fwd(dev(x), dev(ths), FS, want_stash=False)
    ws = wb.step_mse_workspace(B, K, xd.device)
    y0, _, g0, s0, st0 = wb.clipper_step_mse_tp(xd, thd, FS, tgt, 2.0 / (B * T), K, W, ws=ws, time_major=True)
    y0, g0, s0, st0 = y0.clone(), g0.clone(), s0.clone(), st0.clone()
    for _ in range(100):
        y, _, g, s, st = wb.clipper_step_mse_tp(xd, thd, FS, tgt, 2.0 / (B * T), K, W, ws=ws, time_major=True)
        assert torch.equal(g, g0) and torch.equal(s, s0) and torch.equal(st, st0)
    assert torch.equal(y, y0)
    eps = float(np.finfo(float).eps)
    r0 = wb.clipper_step_esr_tp(xd, thd, FS, tgt, B * (T - 50), eps, 50, K, W, ws=ws, time_major=True)
    g0, l0, s10 = r0[3].clone(), r0[4].clone(), r0[2].clone()
    for _ in range(50):
        r = wb.clipper_step_esr_tp(xd, thd, FS, tgt, B * (T - 50), eps, 50, K, W, ws=ws, time_major=True)
        assert torch.equal(r[3], g0) and torch.equal(r[4], l0) and torch.equal(r[2], s10)


def test_fused_step_with_skewed_chunk_spans():
    """Skewed chunk spans (the older wave of a SIMD pair gets the longer chunk: csrc/wdf_clipper_fused.h, chunk_span) are
    applied by the library only for launches of about two waves per SIMD; WDF_FUSED_SKEW=force applies them wherever the
    geometry allows, so the same tests -- repairs, warm starts, ragged shapes, both losses -- run over them in a
    subprocess (the switch is read once per process)."""
    import os, subprocess, sys
    env = dict(os.environ, WDF_FUSED_SKEW="force")
    here = os.path.abspath(__file__)
    sel = ("test_fused_matches_two_kernel_step or test_fused_repairs_when_warmup_is_too_short or "
           "test_fused_warm_started_training_loop_with_adam or test_fused_esr_step_matches_autograd or "
           "test_fused_randomized_plans_and_circuits")
    out = subprocess.run([sys.executable, "-m", "pytest", here, "-x", "-q", "-k", sel], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]
    assert " passed" in out.stdout
