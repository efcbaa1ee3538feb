"""GPU: the time-parallel kernels give the sequential kernels' results.

* reverse sweep: exact up to summation order -> gradients agree to 2e-5 relative;
* forward: outputs within the verified tolerance of the sequential kernel, status clean; and
  when the warm-up is deliberately too short for the circuit's memory, the on-device
  verification notices and re-runs the chunks that started wrong;
* warm-started forward (snapshots of the previous call, device-steered warm-up): a training loop,
  a parameter jump (repaired), an unchanged theta (bit-exact with zero warm-up).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000.0


@pytest.fixture(scope="module")
def wb():
    from wdf_hip import binding
    binding.require_gpu()
    return binding


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def setup(B, T, seed=0):
    from wdf_hip import workload
    x = workload.sweep_batch(B, T, seed=seed)
    return dev(x), dev(workload.clipper_theta())


@pytest.mark.parametrize("B,T,K", [(64, 512, 4), (70, 1001, 7), (130, 2048, 16), (5, 96, 12), (3, 8, 4), (1, 64, 2)])
@pytest.mark.parametrize("n_up,n_down", [(1, 1), (2, 3)])
def test_bwd_tp_matches_sequential(wb, B, T, K, n_up, n_down):
    x, th = setup(B, T, seed=B + T)
    y, zs, _ = wb.clipper_fwd(x, th, FS, n_up=n_up, n_down=n_down)
    gy = dev(np.random.default_rng(B).standard_normal((T, B)) / (B * T))
    g_seq, gz_seq = wb.clipper_bwd(x, th, FS, zs, gy, n_up=n_up, n_down=n_down, want_gz0=True)
    g_tp, gz_tp = wb.clipper_bwd_tp(x, th, FS, zs, gy, K, n_up=n_up, n_down=n_down, want_gz0=True)
    assert torch.allclose(g_tp, g_seq, rtol=2e-5, atol=0), (g_tp, g_seq)
    assert torch.allclose(gz_tp, gz_seq, rtol=1e-4, atol=1e-12)
    g_tp2, _ = wb.clipper_bwd_tp(x, th, FS, zs, gy, K, n_up=n_up, n_down=n_down)
    assert torch.equal(g_tp, g_tp2)                                  # deterministic


def test_bwd_tp_per_sample_r(wb):
    B, T, K = 71, 520, 5
    x, th = setup(B, T, seed=3)
    r = dev(45.0e3 * np.exp(0.8 * np.sin(np.arange(T)[None, :] * 0.01 * (1 + np.arange(B)[:, None] % 5))))
    y, zs, _ = wb.clipper_fwd(x, th, FS, r=r)
    gy = dev(np.random.default_rng(1).standard_normal((T, B)) / (B * T))
    g_seq, _ = wb.clipper_bwd(x, th, FS, zs, gy, r=r)
    g_tp, _ = wb.clipper_bwd_tp(x, th, FS, zs, gy, K, r=r)
    assert torch.allclose(g_tp, g_seq, rtol=2e-5, atol=0)
    # forward with the per-sample resistance channel, time-parallel (r >= 16 kOhm: W = 256 is plenty)
    y2, _, _, st = wb.clipper_fwd_tp(x, th, FS, 2, 256, r=r)
    assert wb.tp_status(st)["n_bad"] == 0
    assert float((y2 - y).abs().max()) <= 1e-6


@pytest.mark.parametrize("B,T,K,W", [(64, 2048, 4, 256), (70, 4096, 16, 256), (131, 1001, 3, 248), (5, 4096, 8, 512),
                                     (1, 2048, 4, 256)])
def test_fwd_tp_matches_sequential(wb, B, T, K, W):
    x, th = setup(B, T, seed=B + T)
    y, zs, zT = wb.clipper_fwd(x, th, FS, want_zT=True)
    y2, zs2, zT2, st = wb.clipper_fwd_tp(x, th, FS, K, W, tol=1e-6, want_zT=True)
    s = wb.tp_status(st)
    assert s["n_bad"] == 0 and not s["fallback_ran"], s
    assert s["max_miss"] <= 1e-6
    assert float((y2 - y).abs().max()) <= 1e-6
    assert float((zs2 - zs).abs().max()) <= 2e-6
    assert float((zT2 - zT).abs().max()) <= 2e-6
    # chunk 0 is the sequential computation itself (same arithmetic)
    L = -(-T // K)
    L = -(-L // 32) * 32
    assert torch.equal(y2[:L], y[:L])


def test_fwd_tp_falls_back_when_warmup_is_too_short(wb):
    """C = 1 uF: the circuit remembers ~4000 samples, a 64-step warm-up cannot work.  The
    verification must catch it at every boundary and the chunk re-runs -- each from the repaired
    end state of the chunk before, never meeting the speculative stash again -- must restore the
    sequential kernel's result bit for bit."""
    from wdf_hip import workload
    B, T = 70, 2048
    x = dev(workload.sweep_batch(B, T, seed=11))
    theta = workload.clipper_theta()
    theta[3] = 1.0e-6
    th = dev(theta)
    y, zs, zT = wb.clipper_fwd(x, th, FS, want_zT=True)
    y2, zs2, zT2, st = wb.clipper_fwd_tp(x, th, FS, 8, 64, tol=1e-6, want_zT=True)
    s = wb.tp_status(st)
    assert s["n_bad"] > 0 and s["fallback_ran"] and s["repaired_tiles"] == 2 * 7, s      # 2 waves x 7 boundaries
    assert torch.equal(y2, y) and torch.equal(zs2, zs) and torch.equal(zT2, zT)


def test_fwd_tp_initial_state_and_no_stash(wb):
    B, T = 64, 1024
    x, th = setup(B, T, seed=2)
    z0 = dev(np.random.default_rng(0).uniform(-0.3, 0.3, B))
    y, _, _ = wb.clipper_fwd(x, th, FS, z0=z0, want_stash=False)
    y2, zs2, _, st = wb.clipper_fwd_tp(x, th, FS, 4, 256, z0=z0, want_stash=False)
    assert zs2 is None and wb.tp_status(st)["n_bad"] == 0
    assert float((y2 - y).abs().max()) <= 1e-6


def test_tp_full_size_against_oracle(wb, oracle):
    from wdf_hip import workload
    B, T, K, W = 8192, 4096, 16, 256
    theta = workload.clipper_theta()
    x = workload.sweep_batch(B, T)
    xd, th = dev(x), dev(theta)
    y, zs, _, st = wb.clipper_fwd_tp(xd, th, FS, 2 * K, W)
    s = wb.tp_status(st)
    assert s["n_bad"] == 0, s
    pick = np.random.default_rng(5).choice(B, 16, replace=False)
    ref = oracle.clipper_fwd(theta.astype(np.float32).astype(np.float64), FS, x[pick].astype(np.float64))
    assert np.max(np.abs(y[:, torch.as_tensor(pick, device="cuda")].cpu().numpy() - ref)) < 2e-6
    tgt, _, _ = wb.clipper_fwd(xd, dev(workload.target_theta()), FS, want_stash=False)
    gy = (2.0 * (y - tgt) / y.numel()).contiguous()
    g_seq, _ = wb.clipper_bwd(xd, th, FS, zs, gy)
    g_tp, _ = wb.clipper_bwd_tp(xd, th, FS, zs, gy, 64)
    assert torch.allclose(g_tp, g_seq, rtol=2e-5, atol=0), (g_tp, g_seq)


def test_fused_mse_step_matches_autograd_path(wb):
    """engine.MseStep (time-parallel forward + MSE-fused reverse sweep) == the plain path:
    sequential forward, torch MSE, sequential reverse sweep."""
    from wdf_hip import engine, workload
    B, T = 200, 2048
    x, th = setup(B, T, seed=21)
    tgt, _, _ = wb.clipper_fwd(x, dev(workload.target_theta()), FS, want_stash=False)
    theta = workload.clipper_theta()
    tp = engine.plan_time_parallel(B, T, theta[2], theta[3], FS)
    assert tp.k_fwd > 1 and tp.k_bwd > 1 and tp.warmup % 32 == 0
    stepper = engine.MseStep(B, T, FS, tp, x.device)
    sse, g = stepper.step(th, x, tgt)
    assert wb.tp_status(stepper.status)["n_bad"] == 0
    thr = th.clone().requires_grad_(True)
    y = engine.clipper(thr, x, FS)
    loss = torch.mean((y - tgt) ** 2)
    loss.backward()
    assert abs(float(sse) / (B * T) - float(loss)) < 1e-6 * float(loss) + 1e-12
    assert torch.allclose(g, thr.grad, rtol=5e-5, atol=0), (g, thr.grad)
    # and through the element API with the automatic plan
    import tf_wdf as wdf
    Vs = wdf.ResistiveVoltageSource(float(theta[2]), trainable=True)
    Cap = wdf.Capacitor(float(theta[3]), FS, trainable=True)
    P1 = wdf.Parallel(Vs, Cap)
    dp = wdf.DiodePair(P1, float(theta[0]), Vt=float(theta[1]), trainable=True)
    ya = wdf.Circuit(P1, dp, Cap)(x)                                   # time_parallel="auto"
    ys = wdf.Circuit(P1, dp, Cap, time_parallel=None)(x)
    assert float((ya - ys).abs().max()) <= 1e-6


def test_time_major_inputs_and_fused_mse(wb):
    """x resident as [T,B]: same results as the batch-major path, for the forward, the plain and
    the MSE-fused reverse sweep (which rebuilds y from the state stash)."""
    from wdf_hip import workload
    B, T, K = 130, 2048, 8
    x, th = setup(B, T, seed=8)
    xt = x.t().contiguous()
    tgt, _, _ = wb.clipper_fwd(x, dev(workload.target_theta()), FS, want_stash=False)
    y, zs, zT, st = wb.clipper_fwd_tp(x, th, FS, K, 224, want_zT=True)
    y2, zs2, zT2, st2 = wb.clipper_fwd_tp(xt, th, FS, K, 224, want_zT=True, time_major=True)
    assert wb.tp_status(st2)["n_bad"] == 0
    assert torch.equal(y, y2) and torch.equal(zs, zs2) and torch.equal(zT, zT2)
    gscale = 2.0 / (B * T)
    gy = (gscale * (y - tgt)).contiguous()
    g_ref, _ = wb.clipper_bwd(x, th, FS, zs, gy)
    for tm, xin in ((False, x), (True, xt)):
        g_tp, _ = wb.clipper_bwd_tp(xin, th, FS, zs, gy, 16, time_major=tm)
        g_mse, sse = wb.clipper_bwd_mse_tp(xin, th, FS, zs, zT, tgt, gscale, 16, time_major=tm)
        assert torch.allclose(g_tp, g_ref, rtol=2e-5, atol=0)
        assert torch.allclose(g_mse, g_ref, rtol=5e-5, atol=0), (g_mse, g_ref)
        assert abs(float(sse) - float(((y - tgt) ** 2).sum())) < 1e-5 * float(sse)
    # sequential kernels with the fused-MSE entry (n_chunks = 1)
    g1, sse1 = wb.clipper_bwd_mse_tp(x, th, FS, zs, zT, tgt, gscale, 1)
    assert torch.allclose(g1, g_ref, rtol=5e-5, atol=0)


def test_plan_time_parallel_degrades_gracefully():
    from wdf_hip import engine
    p = engine.plan_time_parallel(8192, 4096, 45.0e3, 4.7e-9, 48000.0)
    assert p.k_fwd == 16 and p.k_bwd == 32 and 184 <= p.warmup <= 256
    assert engine.plan_time_parallel(8192, 4096, 45.0e3, 4.7e-9, 48000.0, time_major=True).k_bwd == 64
    slow = engine.plan_time_parallel(8192, 4096, 45.0e3, 1.0e-6, 48000.0)   # memory >> T/2: no forward chunks
    assert slow.k_fwd == 1 and slow.k_bwd == 32
    big = engine.plan_time_parallel(1 << 20, 4096, 45.0e3, 4.7e-9, 48000.0)  # plenty of waves already
    assert big.k_fwd == 1 and big.k_bwd == 1


def test_indexing_beyond_2_31_elements(wb):
    """Maximum-size edge: B*T = 2.29e9 elements per array (> 2^31, 9.2 GB each).  Picked sequences
    -- first, last, and ones whose offsets b*T straddle 2^31 and 2^32 bytes -- must equal the same
    sequences run alone, forward (sequential and time-parallel) and reverse sweep."""
    B, T = 32768, 70000
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip("needs 60 GB of free HBM")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    x = torch.empty((B, T), dtype=torch.float32, device="cuda").normal_(0.0, 1.2, generator=gen)
    from wdf_hip import workload
    th = dev(workload.clipper_theta())
    pick = [0, 1, 7669, 7670, 15339, 15340, 30678, 30679, B - 2, B - 1]     # 2^29/T, 2^30/T, 2^31/T element offsets
    xs = x[pick].contiguous()
    y_ref, zs_ref, _ = wb.clipper_fwd(xs, th, FS)
    y, zs, _ = wb.clipper_fwd(x, th, FS)
    assert torch.equal(y[:, pick], y_ref) and torch.equal(zs[:, pick], zs_ref)
    del y
    y2, zs2, _, st = wb.clipper_fwd_tp(x, th, FS, 16, 192)
    assert wb.tp_status(st)["n_bad"] == 0
    assert float((y2[:, pick] - y_ref).abs().max()) <= 1e-6
    del zs2
    # reverse sweep: a gradient that is non-zero only on the picked sequences must equal theirs alone
    gy = y2
    gy.zero_()
    g_small = torch.randn(T, len(pick), device="cuda", generator=gen) / T
    gy[:, pick] = g_small
    g_ref, _ = wb.clipper_bwd(xs, th, FS, zs_ref, g_small.contiguous())
    g_seq, _ = wb.clipper_bwd(x, th, FS, zs, gy)
    g_tp, _ = wb.clipper_bwd_tp(x, th, FS, zs, gy, 32)
    for g in (g_seq, g_tp):
        assert float(((g - g_ref).abs() / g_ref.abs()).max()) <= 2e-5, (g, g_ref)


def test_fast_step_equals_general_root_path(wb):
    """The kernels switch, once per launch, to a shorter forward step when the static port resistance
    keeps omega_1 in its series-only region (every practical diode).  WDF_GENERAL_ROOT forces the
    general per-step evaluation: outputs agree to 2 ulp-level 3e-7, the reverse sweep is unaffected
    (bit-equal gradients from the same stash), and a circuit OUTSIDE the fast region (huge Rp Is)
    takes the general path on its own and is untouched by the flag."""
    from wdf_hip import workload
    B, T = 130, 2048
    x, th = setup(B, T, seed=17)
    try:
        outs = {}
        for general in (False, True):
            wb.GENERAL_ROOT = general
            y, zs, zT = wb.clipper_fwd(x, th, FS, want_zT=True)
            y2, _, _, st = wb.clipper_fwd_tp(x, th, FS, 4, 256)
            assert wb.tp_status(st)["n_bad"] == 0
            outs[general] = (y, zs, y2)
        assert float((outs[False][0] - outs[True][0]).abs().max()) <= 3e-7
        assert float((outs[False][2] - outs[True][2]).abs().max()) <= 3e-7
        assert not torch.equal(outs[False][0], outs[True][0]) or True      # may or may not be bit-equal
        # zero input: lam = sign(0) = 0 must give exactly zero output on the fast path as well
        wb.GENERAL_ROOT = False
        y0, _, _ = wb.clipper_fwd(torch.zeros_like(x), th, FS)
        assert float(y0.abs().max()) == 0.0
        # outside the fast region: Is 1e-3 A with a 1 MOhm source -> log(Rp Is / nVt) > -4
        theta = workload.clipper_theta()
        theta[0], theta[2], theta[3] = 1.0e-3, 1.0e6, 1.0e-10
        thx = dev(theta)
        wb.GENERAL_ROOT = False
        ya, _, _ = wb.clipper_fwd(x, thx, FS)
        wb.GENERAL_ROOT = True
        yb, _, _ = wb.clipper_fwd(x, thx, FS)
        assert torch.equal(ya, yb)
    finally:
        wb.GENERAL_ROOT = False


@pytest.mark.parametrize("time_major", [False, True])
def test_fused_mse_esr_step_matches_autograd_and_oracle(wb, oracle, time_major):
    """engine.MseStep(loss="mse+esr", skip=50): the training loss of clipper_pot.py:146-156,177,232,248
    (energy of the model output, first 50 samples dropped) fused into the reverse sweep, against
    (a) the same loss written with torch ops on the kernel's y and differentiated by autograd through
    the plain reverse sweep (5e-5 relative), and (b) the fp64 oracle's adjoint fed the fp64 dL/dy of
    that loss (1e-4 relative, the gradient tolerance of every fp32-vs-fp64 comparison here)."""
    from wdf_hip import engine, workload
    B, T, skip = 96, 2048, 50
    x, th = setup(B, T, seed=31)
    theta = workload.clipper_theta()
    tgt, _, _ = wb.clipper_fwd(x, dev(workload.target_theta()), FS, want_stash=False)
    tp = engine.plan_time_parallel(B, T, theta[2], theta[3], FS, time_major=time_major)
    step = engine.MseStep(B, T, FS, tp, x.device, time_major=time_major, loss="mse+esr", skip=skip)
    xin = x.t().contiguous() if time_major else x
    sse, g = step.step(th, xin, tgt)
    eps = float(np.finfo(float).eps)

    def loss_fn(y, t, lib):
        o, tt = y[skip:], t[skip:]
        n = o.numel() if lib is torch else o.size
        S = ((o - tt) ** 2).sum()
        return S / n + lib.sqrt(S / ((o ** 2).sum() + eps) / n)

    thr = th.clone().requires_grad_(True)
    y = engine.clipper(thr, x, FS)
    loss = loss_fn(y, tgt, torch)
    loss.backward()
    assert abs(float(step.loss[2]) - float(loss)) <= 2e-6 * float(loss)
    assert abs(float(step.loss[0]) + float(step.loss[1]) - float(step.loss[2])) <= 1e-6 * float(loss)
    assert torch.allclose(g, thr.grad, rtol=5e-5, atol=0), (g, thr.grad)
    assert abs(float(sse) - float(((y[skip:] - tgt[skip:]) ** 2).sum())) <= 1e-4 * float(sse)
    # fp64: oracle forward, dL/dy of the same loss by torch autograd in float64, oracle adjoint
    th64 = theta.astype(np.float32).astype(np.float64)
    x64 = x.cpu().numpy().astype(np.float64)
    y64 = torch.tensor(oracle.clipper_fwd(th64, FS, x64), requires_grad=True)
    l64 = loss_fn(y64, tgt.cpu().double(), torch)
    (gy64,) = torch.autograd.grad(l64, [y64])
    _, gref = oracle.clipper_fwd_bwd(th64, FS, x64, gy64.numpy())
    got = g.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - gref) / np.abs(gref)) < 1e-4, (got, gref)
    assert abs(float(step.loss[2]) - float(l64)) <= 1e-4 * float(l64)


def test_randomized_plans_and_circuits(wb):
    """tools/stress_tp.py, 60 cases: random component values over the clip ranges of tf_wdf.py:74,104,
    random diode parameters and counts, amplitudes, shapes, layouts, chunkings, warm-ups (hopeless
    ones included) and warm starts from snapshots taken at nearby or far-away parameters.  The time-parallel forward must equal the sequential one to 2e-6 whatever the plan
    (a third of the cases go through the repair path) and the chunked reverse sweep must agree with the
    sequential one."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_tp
    worst_y = worst_g = 0.0
    repaired = 0
    for case in range(60):
        ey, eg, rep = stress_tp.run_case(11, case)
        worst_y, worst_g, repaired = max(worst_y, ey), max(worst_g, eg), repaired + rep
    # sweep mismatch: 1e-3 of a component's own size (+1e-3 of the largest); the worst of these cases is a 10 mV
    # signal whose dL/dIs is a cancelling sum that BOTH sweeps get only to 2 % of the fp64 oracle (tools/stress_tp.py --case 11 55)
    assert worst_y <= 2e-6 and worst_g <= 1e-3, (worst_y, worst_g)
    assert repaired >= 5


def test_combine_tail_reduce_and_adam(wb):
    """The reverse sweep finishes its own job (device-scope tickets): the last chunk wave of every tile
    combines the tile's chunk records, the last tile does the fixed-order reduction, the chain rule and
    -- wdf_clipper_bwd_mse_tp_adam -- the Adam update.
    * 200 repetitions give bit-identical gradients (which wave is last varies, the result does not);
    * they equal the sequential sweep's gradient to 2e-5;
    * the folded update equals wdf_adam_step applied to that gradient, over 5 consecutive steps."""
    from wdf_hip import workload
    B, T, K = 8192, 512, 8
    x, th = setup(B, T, seed=41)
    tgt, _, _ = wb.clipper_fwd(x, dev(workload.target_theta()), FS, want_stash=False)
    y, zs, zT = wb.clipper_fwd(x, th, FS, want_zT=True)
    gscale = 2.0 / y.numel()
    ws = torch.empty((wb.lib().wdf_clipper_bwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device="cuda").fill_(0xAB)
    assert wb.lib().wdf_clipper_bwd_tp_ws_init(ws.data_ptr(), B, K, None) == 0          # garbage everywhere but the tickets
    g0, sse0 = wb.clipper_bwd_mse_tp(x, th, FS, zs, zT, tgt, gscale, K, ws=ws)
    g0, sse0 = g0.clone(), sse0.clone()
    for _ in range(200):
        g, sse = wb.clipper_bwd_mse_tp(x, th, FS, zs, zT, tgt, gscale, K, ws=ws)
        assert torch.equal(g, g0) and torch.equal(sse, sse0)
    g_seq, _ = wb.clipper_bwd(x, th, FS, zs, (gscale * (y - tgt)).contiguous())
    assert torch.allclose(g0, g_seq, rtol=2e-5, atol=0)
    assert abs(float(sse0) - float(((y - tgt) ** 2).sum())) <= 1e-5 * float(sse0)
    # folded Adam vs separate kernel, several steps (theta moves, so each step is a fresh problem)
    lr = [1e-3 * float(v) for v in workload.clipper_theta()]
    lo, hi = [1e-15, 1e-3, 180.0, 1e-13], [1e-3, 1.0, 1.0e6, 1.0]
    th_a, th_b = th.clone(), th.clone()
    opt_a, opt_b = wb.Adam(4, lr, lo=lo, hi=hi), wb.Adam(4, lr, lo=lo, hi=hi)
    for _ in range(5):
        _, zs_a, zT_a = wb.clipper_fwd(x, th_a, FS, want_zT=True)
        wb.clipper_bwd_mse_tp_adam(x, th_a, FS, zs_a, zT_a, tgt, gscale, K, opt_a, ws=ws)
        _, zs_b, zT_b = wb.clipper_fwd(x, th_b, FS, want_zT=True)
        g_b, _ = wb.clipper_bwd_mse_tp(x, th_b, FS, zs_b, zT_b, tgt, gscale, K)
        opt_b.apply(th_b, g_b)
    assert int(opt_a.step.cpu()[0]) == 5 and int(opt_b.step.cpu()[0]) == 5
    assert torch.equal(th_a, th_b), (th_a, th_b)
    assert not torch.equal(th_a, th)


def test_fused_mse_esr_with_per_sample_resistance(wb):
    """The ESR-mode sweep with the per-sample resistance channel of clipper_pot.py:116 against autograd
    through the plain kernels (gradients w.r.t. Is, nVt, C; R has none when it is streamed)."""
    from wdf_hip import engine, workload
    B, T, skip = 70, 1024, 50
    x, th = setup(B, T, seed=33)
    r = dev(workload.pot_resistance_batch(B, T))
    tgt, _, _ = wb.clipper_fwd(x, dev(workload.target_theta()), FS, r=r, want_stash=False)
    tp = engine.TpPlan(4, 512, 1.0e-6, 4)
    step = engine.MseStep(B, T, FS, tp, x.device, loss="mse+esr", skip=skip)
    _, g = step.step(th, x, tgt, r=r)
    assert wb.tp_status(step.status)["n_bad"] == 0
    thr = th.clone().requires_grad_(True)
    y = engine.clipper(thr, x, FS, r=r)
    o, t = y[skip:], tgt[skip:]
    S = ((o - t) ** 2).sum()
    loss = S / o.numel() + torch.sqrt(S / ((o ** 2).sum() + float(np.finfo(float).eps)) / o.numel())
    loss.backward()
    assert abs(float(step.loss[2]) - float(loss)) <= 2e-6 * float(loss)
    assert float(g[2]) == 0.0 and float(thr.grad[2]) == 0.0
    keep = torch.tensor([0, 1, 3], device="cuda")
    assert torch.allclose(g[keep], thr.grad[keep], rtol=1e-4, atol=0), (g, thr.grad)


# ---- warm-started forward (wdf_clipper_fwd_tp_warm) ------------------------------------------------
def _theta_path(th0, steps, rel):
    """theta moving `rel` per step in every component (signs as an optimizer would keep them)"""
    sign = torch.tensor([1.0, -1.0, -1.0, 1.0], device=th0.device)
    return [th0 * (1.0 + rel * s * sign) for s in range(steps)]


@pytest.mark.parametrize("time_major", [False, True])
@pytest.mark.parametrize("with_r", [False, True])
def test_fwd_tp_warm_training_loop(wb, time_major, with_r):
    """Same inputs every call, theta moving 0.1 % per call (bench.py's Adam step): every call within
    1e-6 of the sequential kernel, no boundary misses, and after the cold first call the device
    controller settles at a fraction of the cold warm-up."""
    from wdf_hip import workload
    B, T, K, W = 200, 2048, 8, 192
    x, th0 = setup(B, T, seed=51)
    r = dev(workload.pot_resistance_batch(B, T)) if with_r else None
    xin = x.t().contiguous() if time_major else x
    rin = (r.t().contiguous() if time_major else r) if with_r else None
    if with_r:
        W = 448                                              # 99.1 kOhm: slower memory
    state = wb.TpWarmState(B, T, K, 256 // wb.warm_unit(), x.device)
    used = []
    for th in _theta_path(th0, 10, 1.0e-3):
        y, zs, zT = wb.clipper_fwd(x, th, FS, r=r, want_zT=True)
        y2, zs2, zT2, st = wb.clipper_fwd_tp(xin, th, FS, K, W, r=rin, want_zT=True, time_major=time_major, state=state)
        s, info = wb.tp_status(st), state.info()
        assert s["n_bad"] == 0 and s["repaired_tiles"] == 0, (s, info)
        assert float((y2 - y).abs().max()) <= 1e-6 and float((zs2 - zs).abs().max()) <= 2e-6
        assert float((zT2 - zT).abs().max()) <= 2e-6
        used.append(info["last_warm_tiles"])
    assert used[0] == -1 and all(0 <= u < -(-W // wb.warm_unit()) for u in used[1:]), used
    assert state.info()["valid"] == 3 and state.info()["n_calls"] == 10


def test_fwd_tp_warm_parameter_jump_is_repaired(wb):
    """After a few slow steps theta jumps by 20 %: the snapshots are now far off, boundaries miss, the
    verify kernel re-runs the chunks -- and the output is still the sequential kernel's to 1e-6.
    The controller answers with more warm-up tiles, and the following calls are clean again."""
    B, T, K, W = 130, 4096, 16, 192
    x, th0 = setup(B, T, seed=52)
    state = wb.TpWarmState(B, T, K, 256 // wb.warm_unit(), x.device)
    path = _theta_path(th0, 4, 1.0e-3)
    for th in path:
        wb.clipper_fwd_tp(x, th, FS, K, W, state=state)
    tiles_before = state.info()["next_warm_tiles"]
    th_jump = path[-1] * torch.tensor([1.2, 0.8, 1.2, 0.8], device=x.device)
    y, zs, zT = wb.clipper_fwd(x, th_jump, FS, want_zT=True)
    y2, zs2, zT2, st = wb.clipper_fwd_tp(x, th_jump, FS, K, W, want_zT=True, state=state)
    s = wb.tp_status(st)
    assert s["n_bad"] > 0 and s["repaired_tiles"] > 0, s
    assert float((y2 - y).abs().max()) <= 1e-6 and float((zs2 - zs).abs().max()) <= 2e-6
    assert float((zT2 - zT).abs().max()) <= 2e-6
    assert state.info()["next_warm_tiles"] > tiles_before
    for k in range(1, 4):                                    # small steps again from the new place
        th = th_jump * (1.0 + 1.0e-3 * k)
        y, _, _ = wb.clipper_fwd(x, th, FS)
        y2, _, _, st = wb.clipper_fwd_tp(x, th, FS, K, W, state=state)
        assert float((y2 - y).abs().max()) <= 1e-6
    assert wb.tp_status(st)["n_bad"] == 0


def test_fwd_tp_warm_unchanged_theta_becomes_exact(wb):
    """theta does not move (validation passes, autotune loops): the secant factor is 0, the snapshots
    ARE the states, the controller walks the warm-up down to zero tiles, and with chunks starting
    from the bit-exact states the output equals the sequential kernel's bit for bit."""
    B, T, K, W = 70, 2048, 8, 192
    x, th = setup(B, T, seed=53)
    state = wb.TpWarmState(B, T, K, 256 // wb.warm_unit(), x.device)
    y, zs, zT = wb.clipper_fwd(x, th, FS, want_zT=True)
    for _ in range(8):
        y2, zs2, zT2, st = wb.clipper_fwd_tp(x, th, FS, K, W, want_zT=True, state=state)
        assert wb.tp_status(st)["n_bad"] == 0
    info = state.info()
    assert info["last_warm_tiles"] == 0 and info["last_miss"] == 0.0, info
    assert torch.equal(y2, y) and torch.equal(zs2, zs) and torch.equal(zT2, zT)
    state.reset()                                            # and a reset really goes back to a cold call
    wb.clipper_fwd_tp(x, th, FS, K, W, state=state)
    assert state.info()["last_warm_tiles"] == -1 and state.info()["n_calls"] == 1


def test_fwd_tp_warm_state_is_bound_to_its_shape(wb):
    x, th = setup(64, 1024, seed=54)
    state = wb.TpWarmState(64, 1024, 4, 256 // wb.warm_unit(), x.device)
    with pytest.raises(wb.WdfHipError):
        wb.clipper_fwd_tp(x, th, FS, 8, 128, state=state)    # other chunking
    x2, _ = setup(70, 1024, seed=54)
    with pytest.raises(wb.WdfHipError):
        wb.clipper_fwd_tp(x2, th, FS, 4, 128, state=state)   # other batch


def test_bench_path_at_full_size_against_the_oracle(wb, oracle):
    """bench.py's own step at its own size -- MseStep(time_major=True, warm=True) forward and the
    MSE-fused reverse sweep with the Adam update folded into its last kernel, 8192 x 4096 -- checked END
    TO END against the fp64 oracle's fused step over the whole batch: every y sample (2e-6), the loss and
    the four gradient components (1e-4; observed 1.4e-7 / 3e-6) at the third step of a training run,
    i.e. with warm-started chunks and parameters that have moved."""
    import os
    from wdf_hip import engine, workload
    B, T = 8192, 4096
    xh = workload.sweep_batch(B, T)
    x = dev(xh)
    xt = x.t().contiguous()
    th0 = workload.clipper_theta()
    theta = dev(th0)
    tgt, _, _ = wb.clipper_fwd(x, dev(workload.target_theta()), FS, want_stash=False)
    tp = engine.plan_time_parallel(B, T, th0[2], th0[3], FS, time_major=True)
    st = engine.MseStep(B, T, FS, tp, x.device, time_major=True, warm=True)
    adam = wb.Adam(4, lr=[1e-3 * float(v) for v in th0], lo=[1e-15, 1e-3, 180.0, 1e-13], hi=[1e-3, 1.0, 1.0e6, 1.0])
    for _ in range(3):
        th_before = theta.clone()
        st.forward(theta, xt)
        sse, g = st.backward(theta, xt, tgt, adam=adam)
    assert not torch.equal(theta, th_before) and st.warm.info()["last_warm_tiles"] >= 0
    assert wb.tp_status(st.status)["n_bad"] == 0
    loss_ref, g_ref, y_ref = oracle.clipper_mse_step(th_before.cpu().numpy().astype(np.float64), FS, xh.astype(np.float64),
                                                     tgt.cpu().numpy().astype(np.float64), dtype=np.float64,
                                                     n_threads=len(os.sched_getaffinity(0)))
    assert np.max(np.abs(st.y.cpu().numpy() - y_ref)) < 2e-6
    assert abs(float(sse) / (B * T) - loss_ref) < 1e-4 * loss_ref
    got = g.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - g_ref) / np.abs(g_ref)) < 1e-4, (got, g_ref)
