"""GPU: the one-pass MSE training step of small trees with a DIODE-PAIR root and resident component values
(csrc/wdf_ss_nl_step.h through Circuit.to_device() + Circuit.mse()): HPFDiodeClipper.h:28-32's circuit in lpf.py:77-99's loop
shape -- forward, loss and the five gradients against the fp64 oracle (cold chunks, predicted warm starts, forced misses that
the finishing waves repair sequentially), a two-state tree against the host-probe path, and the device's warm-up control."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000
THETA = np.array([33.0e3, 1.0e3, 22.0e-9, 4.352e-9, 25.85e-3 * 1.906], dtype=np.float32).astype(np.float64)


def cuda(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.abs(b)))


@pytest.fixture
def wdf():
    import tf_wdf
    return tf_wdf


def hpf(wdf, n_up=2, n_down=3, theta=THETA):
    """HPFDiodeClipper.h:28-32: Parallel(R, Series(Vs, C)) + diode pair."""
    R = wdf.Resistor(float(theta[0]), True)
    Vs = wdf.ResistiveVoltageSource(float(theta[1]), trainable=True)
    C = wdf.Capacitor(float(theta[2]), FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs, C))
    dp = wdf.DiodePair(top, float(theta[3]), Vt=float(theta[4]), nDiodes=1.0, N_up=n_up, N_down=n_down, trainable=True)
    return wdf.Circuit(top, dp, R), [R.R, Vs.R, C.C, dp.Is, dp.nVt]


def oracle_hpf(O, theta, x, tgt, n_up, n_down):
    """-> (y [T,B], mean squared error, its gradient w.r.t. {R, Rs, C, Is, nVt}) in float64."""
    nodes = [(O.NODE_RESISTOR, -1, -1, 0, -1, -1), (O.NODE_RES_VSOURCE, -1, -1, 1, 0, -1),
             (O.NODE_CAPACITOR, -1, -1, 2, -1, -1), (O.NODE_SERIES, 1, 2, -1, -1, -1), (O.NODE_PARALLEL, 0, 3, -1, -1, -1)]
    oc = O.Circuit(nodes, top=4, probe=0, n_in=1, root_kind=O.ROOT_DIODE_PAIR, fs=FS, p_is=3, p_nvt=4, n_up=n_up, n_down=n_down)
    y = O.tree_fwd(oc, theta, x.astype(np.float64))
    e = y - tgt.astype(np.float64)
    gy = 2.0 * e / e.size
    return y, float(np.mean(e * e)), O.tree_grad(oc, theta, x.astype(np.float64), gy)


def one_call(wdf, circ, params, x, tgt):
    tf = wdf.tf
    with tf.GradientTape() as tape:
        loss = circ.mse(x, tgt)
    g = np.array([float(v) for v in tape.gradient(loss, params)])
    return float(loss), g, circ.last_output.detach().cpu().numpy()


@pytest.mark.parametrize("B,T,n_up,n_down", [(1, 40, 2, 3), (70, 1000, 2, 3), (130, 1501, 1, 1), (300, 4096, 2, 2), (64, 8192, 1, 2)])
def test_first_call_against_the_oracle(wdf, oracle, B, T, n_up, n_down):
    """Cold: chunks warmed up from z = 0 for the planner's estimate; odd batches take the one-sequence-per-lane kernels,
    T % 8 != 0 the tail, N_up != N_down the per-sign diode constants."""
    rng = np.random.default_rng(B + T)
    x = (rng.standard_normal((B, T)) * 1.2).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)
    circ, params = hpf(wdf, n_up, n_down)
    circ.to_device()
    loss, g, y = one_call(wdf, circ, params, cuda(x), cuda(tgt))
    yref, lref, gref = oracle_hpf(oracle, THETA, x, tgt, n_up, n_down)
    ctl = circ._tree.read_ctl(next(iter(circ._tree.cache.values())))
    print(f"B {B} T {T}: |y - oracle| {np.max(np.abs(y - yref)):.2e}, loss {loss:.6e} / {lref:.6e}, gradients {rel(g, gref):.2e}; {ctl}")
    assert np.max(np.abs(y - yref)) < 3e-6
    assert abs(loss - lref) < 2e-6 * lref
    assert rel(g, gref) < 3e-4


def test_training_loop_against_the_oracle_every_step(wdf, oracle):
    """lpf.py:86-99's loop on the HPF clipper, five optimizers: every step's output, loss and gradients against the oracle AT THE
    PARAMETERS OF THAT STEP (the chunks of step n start from step n - 1's states moved along their tangents); the warm-up
    the device settles on is a fraction of the cold one and no group needs the sequential repair."""
    tf = wdf.tf
    rng = np.random.default_rng(3)
    B, T = 128, 4096
    x = (rng.standard_normal((B, T)) * 1.2).astype(np.float32)
    ref, _ = hpf(wdf)
    tgt = (ref(cuda(x)) * 0.8).as_subclass(torch.Tensor).detach().cpu().numpy()
    circ, params = hpf(wdf)
    circ.to_device()
    opts = [tf.keras.optimizers.Adam(learning_rate=1.0e-3 * float(p)) for p in params]
    xd, td = cuda(x), cuda(tgt)
    worst_y = worst_g = 0.0
    used, gated = [], 0
    for step in range(24):
        theta = np.array([float(p) for p in params], dtype=np.float32).astype(np.float64)
        with tf.GradientTape() as tape:
            loss = circ.mse(xd, td)
        grads = tape.gradient(loss, params)
        g = np.array([float(v) for v in grads])
        y = circ.last_output.detach().cpu().numpy()
        ent = next(iter(circ._tree.cache.values()))
        ctl = circ._tree.read_ctl(ent)
        used.append(ctl["w_used"])
        gated += ctl["gated_groups"]
        if step in (0, 1, 2, 5, 11, 23):
            yref, lref, gref = oracle_hpf(oracle, theta, x, tgt, 2, 3)
            worst_y = max(worst_y, float(np.max(np.abs(y - yref))))
            worst_g = max(worst_g, rel(g, gref))
            assert abs(float(loss) - lref) < 1e-5 * lref, (step, float(loss), lref)
        for o, gr, p in zip(opts, grads, params):
            o.apply_gradients([(gr, p)])
    print(f"warm-ups used {used}; groups repaired {gated}; worst |y - oracle| {worst_y:.2e}, worst gradient error {worst_g:.2e}")
    assert worst_y < 4e-6 and worst_g < 5e-4
    assert used[0] >= 256 and max(used[2:]) <= 128
    assert gated == 0


def test_missed_boundaries_are_repaired_sequentially(wdf, oracle):
    """A tolerance no prediction can meet (negative: an unchanged circuit arrives bit for bit where its predecessor ended):
    every group's finishing wave runs its sequences again from t = 0 -- the sequential recursion itself (states, tangents,
    sums)."""
    rng = np.random.default_rng(8)
    B, T = 200, 2048
    x = (rng.standard_normal((B, T)) * 1.2).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)
    circ, params = hpf(wdf)
    circ.to_device()
    xd, td = cuda(x), cuda(tgt)
    one_call(wdf, circ, params, xd, td)
    ent = next(iter(circ._tree.cache.values()))
    from wdf_hip import binding
    binding._check(binding.lib().wdf_ss_nl_step_set(binding._ptr(ent["ws"]), 8, -1.0, binding._stream()), "set")
    loss, g, y = one_call(wdf, circ, params, xd, td)
    ctl = circ._tree.read_ctl(ent)
    yref, lref, gref = oracle_hpf(oracle, THETA, x, tgt, 2, 3)
    print(f"{ctl}; |y - oracle| {np.max(np.abs(y - yref)):.2e}, gradients {rel(g, gref):.2e}")
    assert ctl["gated_groups"] == 2 and ctl["n_bad"] > 0              # 200 sequences, two per lane: two groups
    assert np.max(np.abs(y - yref)) < 3e-6 and abs(loss - lref) < 2e-6 * lref and rel(g, gref) < 3e-4
    # and the call after it starts from the repaired call's snapshots
    binding._check(binding.lib().wdf_ss_nl_step_set(binding._ptr(ent["ws"]), 8, 1.0e-6, binding._stream()), "set")
    loss2, g2, y2 = one_call(wdf, circ, params, xd, td)
    ctl2 = circ._tree.read_ctl(ent)
    print(ctl2)
    assert ctl2["gated_groups"] == 0
    assert np.max(np.abs(y2 - yref)) < 3e-6 and rel(g2, gref) < 3e-4


def test_a_jump_of_the_components_is_caught(wdf, oracle):
    """The snapshots predict the next call's states to first order; a component assigned a very different value between two
    calls breaks the prediction -- the boundaries miss, the groups are repaired, the answer is the oracle's."""
    rng = np.random.default_rng(9)
    B, T = 128, 4096
    x = (rng.standard_normal((B, T)) * 1.2).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)
    circ, params = hpf(wdf)
    circ.to_device()
    xd, td = cuda(x), cuda(tgt)
    for _ in range(3):
        one_call(wdf, circ, params, xd, td)
    params[2].assign(40.0e-9)                                       # C: 22 nF -> 40 nF
    params[0].assign(20.0e3)
    theta = np.array([float(p) for p in params], dtype=np.float32).astype(np.float64)
    loss, g, y = one_call(wdf, circ, params, xd, td)
    ctl = circ._tree.read_ctl(next(iter(circ._tree.cache.values())))
    yref, lref, gref = oracle_hpf(oracle, theta, x, tgt, 2, 3)
    print(f"{ctl}; |y - oracle| {np.max(np.abs(y - yref)):.2e}, gradients {rel(g, gref):.2e}")
    assert np.max(np.abs(y - yref)) < 4e-6 and abs(loss - lref) < 1e-5 * lref and rel(g, gref) < 5e-4


def test_two_state_diode_tree_against_the_host_probe_path(wdf):
    """ns = 2, ni = 1 with a diode-pair root: 13 tangents of two components each, 2 x 2 Psi."""
    tf = wdf.tf
    rng = np.random.default_rng(6)
    B, T = 96, 3000
    x = (rng.standard_normal((B, T)) * 1.5).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)

    def build():
        Ra = wdf.Resistor(4.7e3, True)
        Vr = wdf.ResistiveVoltageSource(1.0e3, trainable=True)
        Ca, Cb = wdf.Capacitor(4.7e-8, FS, True), wdf.Capacitor(2.2e-8, FS, True)
        top = wdf.Parallel(wdf.Series(Ra, Ca), wdf.Series(Vr, Cb))
        dp = wdf.DiodePair(top, 2.52e-9, Vt=25.85e-3, nDiodes=1.752, trainable=True)
        return wdf.Circuit(top, dp, Ra), [Ra.R, Vr.R, Ca.C, Cb.C, dp.Is, dp.nVt]

    ref, pr = build()
    assert (ref.ns, ref.ni) == (2, 1)
    with tf.GradientTape() as tape:
        y = ref(cuda(x))
        l0 = tf.reduce_mean(tf.square(y - cuda(tgt)))
    g0 = np.array([float(v) for v in tape.gradient(l0, pr)])
    circ, p = build()
    circ.to_device()
    xd, td = cuda(x), cuda(tgt)
    for call in range(3):                                          # cold, then twice from the snapshots
        l1, g1, y1 = one_call(wdf, circ, p, xd, td)
        ctl = circ._tree.read_ctl(next(iter(circ._tree.cache.values())))
        e_y = float(np.max(np.abs(y1 - y.as_subclass(torch.Tensor).detach().cpu().numpy())))
        print(f"call {call}: loss {float(l0):.6e} / {l1:.6e}; |y - host path| {e_y:.2e}; gradients {rel(g1, g0):.2e}; {ctl}")
        assert e_y < 4e-6 and abs(l1 - float(l0)) < 1e-5 * float(l0) and rel(g1, g0) < 5e-4


def test_persistent_misses_halve_the_chunk_count(wdf, oracle, monkeypatch):
    """A tolerance below fp32's resolution of the states: every call of a training loop misses although the warm-up is as long
    as a chunk.  The host looks at the control block every 16 calls (16 calls late, no synchronisation) and plans half as many
    chunks (down to one, which has no boundary, if need be); results stay the oracle's throughout (missed groups are repaired)."""
    from wdf_hip import lowering
    monkeypatch.setattr(lowering, "NL_TOL", 1.0e-12)
    tf = wdf.tf
    rng = np.random.default_rng(21)
    B, T = 128, 2048
    x = (rng.standard_normal((B, T)) * 1.2).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)
    circ, params = hpf(wdf)
    circ.to_device()
    opts = [tf.keras.optimizers.Adam(learning_rate=1.0e-4 * float(p)) for p in params]
    xd, td = cuda(x), cuda(tgt)
    ks = []
    for step in range(224):
        theta = np.array([float(p) for p in params], dtype=np.float32).astype(np.float64) if step % 56 == 55 else None
        with tf.GradientTape() as tape:
            loss = circ.mse(xd, td)
        grads = tape.gradient(loss, params)
        ent = next(iter(circ._tree.cache.values()))
        ks.append(ent["k"])
        if theta is not None:
            torch.cuda.synchronize()
            yref, lref, gref = oracle_hpf(oracle, theta, x, tgt, 2, 3)
            g = np.array([float(v) for v in grads])
            assert np.max(np.abs(circ.last_output.detach().cpu().numpy() - yref)) < 3e-6 and rel(g, gref) < 3e-4
            assert abs(float(loss) - lref) < 2e-6 * lref
        for o, gr, p in zip(opts, grads, params):
            o.apply_gradients([(gr, p)])
        if step % 16 == 15:
            torch.cuda.synchronize()                               # (so that the 16-call-old copy has landed when it is looked at)
    print(f"chunks per call: {sorted(set(ks), reverse=True)}; re-plans {ent['replans']}")
    # (it stops halving where the longer chunks' warm-up brings a chunk to its predecessor's end BIT FOR BIT)
    assert ks[0] == 32 and ks[-1] < 32 and all(b <= a for a, b in zip(ks, ks[1:])) and 2 ** ent["replans"] == ks[0] // ks[-1]
    assert circ._tree.read_ctl(ent)["gated_groups"] == 0


def test_one_launch_for_the_optimizers_of_a_step(wdf):
    """One Adam per component (lpf.py:79-80,93-94): the updates of a step are queued on the parameter block and go out as ONE
    launch before anything reads the values -- same numbers as five separate launches, in the order they were asked for."""
    tf = wdf.tf
    rng = np.random.default_rng(4)
    B, T = 64, 1024
    x, tgt = cuda(rng.standard_normal((B, T)) * 1.2), cuda(0.3 * rng.standard_normal((T, B)))
    ends = []
    for deferred in (True, False):
        circ, params = hpf(wdf)
        circ.to_device()
        pb = circ._tree.pb
        opts = [tf.keras.optimizers.Adam(learning_rate=2.0e-3 * float(p)) for p in params]
        for step in range(6):
            with tf.GradientTape() as tape:
                loss = circ.mse(x, tgt)
            grads = tape.gradient(loss, params)
            for o, g, p in zip(opts, grads, params):
                o.apply_gradients([(g, p)])
                if not deferred:
                    pb.flush()
            if deferred:
                assert len(pb.pending) == 5
                if step == 2:
                    opts[0].apply_gradients([(grads[0], params[0])])   # the same optimizer again: the queue goes out first
                    assert len(pb.pending) == 1
                if step == 3:
                    v = float(params[2])                                # a read sends them
                    assert len(pb.pending) == 0 and v != THETA[2]
            elif step == 2:
                opts[0].apply_gradients([(grads[0], params[0])])
                pb.flush()
        ends.append(([float(p) for p in params], float(loss)))
        assert len(pb.pending) == 0
    assert ends[0] == ends[1], ends


def test_full_size_step_against_the_two_pass_path(wdf):
    """8192 x 4096 (the bench shape): the one-pass step -- cold, then two calls from the snapshots across Adam steps -- against
    the plain path on the same components (host probe, ss_fwd_tp + verified chunks, torch's loss, ss_bwd_tp): outputs of every
    sequence, loss and the five gradients; and linearity of the loss in the target: L(t) - 2 L((t + t2) / 2) + L(t2) =
    mean((t - t2)^2) / 2 whatever the circuit does."""
    tf = wdf.tf
    g = torch.Generator(device="cpu").manual_seed(77)
    B, T = 8192, 4096
    x = (torch.randn((B, T), generator=g) * 1.2).cuda()
    tgt = (0.3 * torch.randn((T, B), generator=g)).cuda()
    circ, params = hpf(wdf, 1, 1)
    circ.to_device()
    opts = [tf.keras.optimizers.Adam(learning_rate=1.0e-3 * float(p)) for p in params]
    for call in range(3):
        theta = [float(p) for p in params]
        with tf.GradientTape() as tape:
            loss = circ.mse(x, tgt)
        grads = tape.gradient(loss, params)
        g1 = np.array([float(v) for v in grads])
        ref, pr = hpf(wdf, 1, 1, theta=np.array(theta))
        with tf.GradientTape() as tape:
            y0 = ref(x)
            l0 = tf.reduce_mean(tf.square(y0 - tgt))
        g0 = np.array([float(v) for v in tape.gradient(l0, pr)])
        e_y = float((circ.last_output - y0.as_subclass(torch.Tensor).detach()).abs().max())
        ctl = circ._tree.read_ctl(next(iter(circ._tree.cache.values())))
        print(f"call {call}: |y - two-pass| {e_y:.2e}, loss {float(loss):.7e} / {float(l0):.7e}, gradients {rel(g1, g0):.2e}, "
              f"warm-up {ctl['w_used']}, miss {ctl['max_miss']:.1e}, repaired {ctl['gated_groups']}")
        assert e_y < 4e-6 and abs(float(loss) - float(l0)) < 2e-6 * float(l0) and rel(g1, g0) < 3e-4
        assert ctl["gated_groups"] == 0
        for o, gr, p in zip(opts, grads, params):
            o.apply_gradients([(gr, p)])
    t2 = (0.3 * torch.randn((T, B), generator=g)).cuda()
    la, lb, lm = float(circ.mse(x, tgt)), float(circ.mse(x, t2)), float(circ.mse(x, ((tgt + t2) * 0.5).contiguous()))
    want = float(((tgt - t2) ** 2).mean()) * 0.5
    print(f"L(t) - 2 L(mid) + L(t2) = {la - 2.0 * lm + lb:.7e}, mean((t - t2)^2) / 2 = {want:.7e}")
    assert abs((la - 2.0 * lm + lb) - want) < 2e-5 * want


def test_one_state_two_sources_diode_tree_against_the_host_probe_path(wdf):
    """ns = 1, ni = 2 with a diode-pair root (two resistive sources): nine tangents, x as [B, T, 2]."""
    tf = wdf.tf
    rng = np.random.default_rng(16)
    B, T = 192, 2048
    x = (rng.standard_normal((B, T, 2)) * 1.2).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)

    def build():
        Va = wdf.ResistiveVoltageSource(1.0e3, trainable=True)
        Vb = wdf.ResistiveVoltageSource(4.7e3, trainable=True)
        Ca = wdf.Capacitor(2.2e-8, FS, True)
        top = wdf.Parallel(wdf.Series(Va, Ca), Vb)
        dp = wdf.DiodePair(top, 2.52e-9, Vt=25.85e-3, nDiodes=1.752, trainable=True)
        return wdf.Circuit(top, dp, Ca), [Va.R, Vb.R, Ca.C, dp.Is, dp.nVt]

    ref, pr = build()
    assert (ref.ns, ref.ni) == (1, 2)
    with tf.GradientTape() as tape:
        y = ref(cuda(x))
        l0 = tf.reduce_mean(tf.square(y - cuda(tgt)))
    g0 = np.array([float(v) for v in tape.gradient(l0, pr)])
    circ, p = build()
    circ.to_device()
    xd, td = cuda(x), cuda(tgt)
    for call in range(3):
        l1, g1, y1 = one_call(wdf, circ, p, xd, td)
        ctl = circ._tree.read_ctl(next(iter(circ._tree.cache.values())))
        e_y = float(np.max(np.abs(y1 - y.as_subclass(torch.Tensor).detach().cpu().numpy())))
        print(f"call {call}: loss {float(l0):.6e} / {l1:.6e}; |y - host path| {e_y:.2e}; gradients {rel(g1, g0):.2e}; w {ctl['w_used']}")
        assert e_y < 4e-6 and abs(l1 - float(l0)) < 1e-5 * float(l0) and rel(g1, g0) < 5e-4
        assert ctl["gated_groups"] == 0


def test_general_root_evaluation_against_the_oracle(wdf, oracle):
    """A diode pair whose omega_1 argument leaves the series-only region (L - log N = -2.3: large Is, large port resistance):
    the chunk kernel takes the general evaluation with its per-step ballot; two calls (cold, then from the snapshots)."""
    rng = np.random.default_rng(31)
    theta = np.array([1.0e5, 1.0e4, 2.2e-8, 1.0e-6, 0.045], dtype=np.float32).astype(np.float64)
    B, T = 130, 1500
    x = (rng.standard_normal((B, T)) * 1.2).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)
    circ, params = hpf(wdf, 2, 2, theta=theta)
    circ.to_device()
    rp = float(circ._tree.host_coef()[1])
    assert np.log(rp * theta[3] / theta[4]) - np.log(2.0) > -4.0
    yref, lref, gref = oracle_hpf(oracle, theta, x, tgt, 2, 2)
    for call in range(2):
        loss, g, y = one_call(wdf, circ, params, cuda(x), cuda(tgt))
        print(f"call {call}: |y - oracle| {np.max(np.abs(y - yref)):.2e}, loss {abs(loss - lref) / lref:.1e}, gradients {rel(g, gref):.1e}")
        assert np.max(np.abs(y - yref)) < 3e-6 and abs(loss - lref) < 2e-6 * lref and rel(g, gref) < 3e-4
