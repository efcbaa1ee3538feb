"""GPU: the one-pass MSE training step of LINEAR trees with the component values resident on the device
(csrc/wdf_ss_step.h, Circuit.to_device() + Circuit.mse()): lpf.py:20-49,77-99 and voltage_divider.py:19-46,69-93 -- against
the goldens recorded from the reference's own Model classes (g1, g2), against the fp64 oracle, and against the host-probe path."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000


def cuda(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.abs(b)))


@pytest.fixture
def wdf():
    import tf_wdf
    return tf_wdf


def build_lpf(wdf):
    Vs = wdf.IdealVoltageSource()
    R1 = wdf.Resistor(1000, True)
    C1 = wdf.Capacitor(1.0e-6, FS, True)
    I1 = wdf.Inverter(wdf.Series(R1, C1))
    return Vs, R1, C1, I1


def test_resident_rc_lowpass_against_the_reference_golden(wdf, golden):
    """lpf.py's Model on its own sweep (g1: the reference's Model.forward and tape.gradient executed): y, loss, dMSE/dC,
    dMSE/dR from ONE pass, the component values on the device."""
    tf = wdf.tf
    g = golden("g1_rc_lowpass.npz")
    Vs, R1, C1, I1 = build_lpf(wdf)
    circ = wdf.Circuit(I1, Vs, C1).to_device()
    x, tgt = cuda(g["x"][None, :]), cuda(g["target"][:, None])
    with tf.GradientTape() as tape:
        loss = circ.mse(x, tgt)
    grads = tape.gradient(loss, [C1.C, R1.R])                    # lpf.py:90,98-99 order
    assert C1.C.is_cuda and R1.R.is_cuda
    assert np.max(np.abs(circ.last_output.cpu().numpy()[:, 0] - g["y_f64"])) < 2e-6
    assert abs(float(loss) - float(g["loss_f64"])) < 1e-6
    assert rel(grads[0].cpu().numpy(), g["dC_f64"]) < 2e-4
    assert rel(grads[1].cpu().numpy(), g["dR_f64"]) < 2e-4


def test_resident_rc_lowpass_never_reset_epochs_against_the_reference_golden(wdf, golden):
    """lpf.py:30-49 never resets C1: epoch 2 starts from epoch 1's final state (g1: the reference's Model.forward called
    twice, the second tape's gradient with the stored state as its constant).  circ.mse(..., carry_state=True) on the resident
    one-pass step: y, loss, the state handed over and both gradients of BOTH epochs."""
    tf = wdf.tf
    g = golden("g1_rc_lowpass.npz")
    Vs, R1, C1, I1 = build_lpf(wdf)
    circ = wdf.Circuit(I1, Vs, C1).to_device()
    x, tgt = cuda(g["x"][None, :]), cuda(g["target"][:, None])
    for tag, zkey in (("", "z_after_f64"), ("_second_call", "z_after_second_call_f64")):
        with tf.GradientTape() as tape:
            loss = circ.mse(x, tgt, carry_state=True)
        grads = tape.gradient(loss, [C1.C, R1.R])
        assert np.max(np.abs(circ.last_output.cpu().numpy()[:, 0] - g["y" + tag + "_f64"])) < 2e-6
        assert abs(float(loss) - float(g["loss" + tag + "_f64"])) < 1e-6
        assert abs(float(circ.last_state[0, 0]) - float(g[zkey][0])) < 2e-6
        assert rel(grads[0].cpu().numpy(), g["dC" + tag + "_f64"]) < 1e-4
        assert rel(grads[1].cpu().numpy(), g["dR" + tag + "_f64"]) < 1e-4
    # an explicit z0 is the same hand-over; reset_state() is Capacitor.reset (tf_wdf.py:117-118)
    z1 = torch.full((1, 1), float(g["z_after_f64"][0]), device="cuda")
    circ.mse(x, tgt, z0=z1)
    assert np.max(np.abs(circ.last_output.cpu().numpy()[:, 0] - g["y_second_call_f64"])) < 2e-6
    circ.reset_state()
    circ.mse(x, tgt, carry_state=True)
    assert np.max(np.abs(circ.last_output.cpu().numpy()[:, 0] - g["y_f64"])) < 2e-6


@pytest.mark.parametrize("B,T,resident", [(70, 1000, True), (300, 4096, True), (70, 1000, False)])
def test_rc_lowpass_three_never_reset_epochs_against_the_oracle(wdf, oracle, B, T, resident):
    """Three epochs of lpf.py's loop shape (forward from the previous epoch's final state, MSE, tape.gradient, Adam on R and
    C), every epoch against the oracle at that epoch's own component values with z0 = the oracle's own zT hand-over:
    y, the state handed on, and the gradient (central differences of the fp64 oracle, z0 held -- the new tape's constant).
    resident=False: the same call on host-resident Variables (composed from __call__(x, z0, return_state))."""
    tf = wdf.tf
    O = oracle
    rng = np.random.default_rng(B + T + 1)
    x = rng.standard_normal((B, T)).astype(np.float32)
    tgt = (0.5 * rng.standard_normal((T, B))).astype(np.float32)
    Vs, R1, C1, I1 = build_lpf(wdf)
    circ = wdf.Circuit(I1, Vs, C1)
    if resident:
        circ.to_device()
    opts = [tf.keras.optimizers.Adam(learning_rate=25.0), tf.keras.optimizers.Adam(learning_rate=1.0e-8)]   # lpf.py:79-80
    oc = O.rc_lowpass_circuit(float(FS))
    x64 = x.astype(np.float64)
    z_ref = None
    xd, td = cuda(x), cuda(tgt)
    for epoch in range(3):
        theta = np.array([float(R1.R), float(C1.C)], dtype=np.float32).astype(np.float64)
        with tf.GradientTape() as tape:
            loss = circ.mse(xd, td, carry_state=True)
        grads = tape.gradient(loss, [R1.R, C1.C])
        yref, zT = O.tree_fwd(oc, theta, x64, z0=z_ref, return_state=True)

        def loss_at(th):
            return np.mean((O.tree_fwd(oc, th, x64, z0=z_ref) - tgt) ** 2)

        gref = []
        for k in range(2):
            h = 1e-5 * theta[k]
            tp, tm = theta.copy(), theta.copy()
            tp[k] += h
            tm[k] -= h
            gref.append((loss_at(tp) - loss_at(tm)) / (2 * h))
        got = np.array([float(v) for v in grads])
        e_y = float(np.max(np.abs(circ.last_output.cpu().numpy() - yref)))
        e_z = float(np.max(np.abs(circ.last_state.cpu().numpy()[0] - zT[:, 1])))
        print(f"epoch {epoch}: |y - oracle| {e_y:.2e}, |zT - oracle| {e_z:.2e}, grads {rel(got, np.array(gref)):.2e}")
        assert e_y < 3e-6 and e_z < 3e-6
        assert abs(float(loss) - np.mean((yref - tgt) ** 2)) < 1e-5 * np.mean((yref - tgt) ** 2)
        assert rel(got, np.array(gref)) < 3e-4
        opts[0].apply_gradients([(grads[0], R1.R)])
        opts[1].apply_gradients([(grads[1], C1.C)])
        # the hand-over in the oracle's own precision: what the GPU carries differs by its fp32 rounding only (checked above)
        z_ref = zT
    assert abs(float(R1.R) - 1000.0) > 10.0                      # (the loop really moved the components)


def test_mse_esr_carries_the_state_like_mse(wdf, golden):
    """circ.mse_esr(..., carry_state=True): the second call starts where the first ended (g1's second-call output), the loss is
    clipper_pot.py:146-156,177's on that output."""
    g = golden("g1_rc_lowpass.npz")
    Vs, R1, C1, I1 = build_lpf(wdf)
    circ = wdf.Circuit(I1, Vs, C1)
    x, tgt = cuda(g["x"][None, :]), cuda(g["target"][:, None])
    for key in ("y_f64", "y_second_call_f64"):
        loss = circ.mse_esr(x, tgt, skip=50, carry_state=True)
        y = circ.last_output.cpu().numpy()[:, 0]
        assert np.max(np.abs(y - g[key])) < 2e-6
        o, t = g[key][50:], g["target"][50:]
        S, E = np.sum((o - t) ** 2), np.sum(o * o) + np.finfo(float).eps
        ref = S / o.size + np.sqrt(S / E / o.size)
        assert abs(float(loss) - ref) < 2e-5 * ref


def test_resident_voltage_divider_against_the_reference_golden(wdf, golden):
    tf = wdf.tf
    g = golden("g2_voltage_divider.npz")
    Vs = wdf.IdealVoltageSource()
    R1, R2 = wdf.Resistor(2.0e3, True), wdf.Resistor(100.0, True)
    circ = wdf.Circuit(wdf.Inverter(wdf.Series(R1, R2)), Vs, R1).to_device()
    assert circ.ns == 0
    with tf.GradientTape() as tape:
        loss = circ.mse(cuda(g["x"][None, :]), cuda(g["target"][:, None]))
    grads = tape.gradient(loss, [R1.R, R2.R])
    assert np.max(np.abs(circ.last_output.cpu().numpy()[:, 0] - g["x"] * 2000.0 / 2100.0)) < 1e-6      # analytic
    assert rel(grads[0].cpu().numpy(), g["dR1_f64"]) < 2e-4
    assert rel(grads[1].cpu().numpy(), g["dR2_f64"]) < 2e-4


@pytest.mark.parametrize("B,T", [(1, 40), (70, 1000), (300, 4096)])
def test_resident_rc_lowpass_against_the_oracle(wdf, oracle, B, T):
    """Ragged batches and chunk counts: y, the loss and both gradients against the oracle (tree interpreter, fp64,
    complex-step derivative)."""
    tf = wdf.tf
    O = oracle
    rng = np.random.default_rng(B + T)
    x = rng.standard_normal((B, T)).astype(np.float32)
    tgt = (0.5 * rng.standard_normal((T, B))).astype(np.float32)
    Vs, R1, C1, I1 = build_lpf(wdf)
    circ = wdf.Circuit(I1, Vs, C1).to_device()
    with tf.GradientTape() as tape:
        loss = circ.mse(cuda(x), cuda(tgt))
    grads = tape.gradient(loss, [R1.R, C1.C])
    nodes = [(O.NODE_RESISTOR, -1, -1, 0, -1, -1), (O.NODE_CAPACITOR, -1, -1, 1, -1, -1), (O.NODE_SERIES, 0, 1, -1, -1, -1),
             (O.NODE_INVERTER, 2, -1, -1, -1, -1)]
    oc = O.Circuit(nodes, top=3, probe=1, n_in=1, root_kind=O.ROOT_IDEAL_VSOURCE, fs=FS, root_vin=0)
    theta = np.array([1000.0, 1.0e-6], dtype=np.float32).astype(np.float64)
    yref = O.tree_fwd(oc, theta, x.astype(np.float64))
    gy = 2.0 * (yref - tgt) / (B * T)
    gref = O.tree_grad(oc, theta, x.astype(np.float64), gy)
    e_y = float(np.max(np.abs(circ.last_output.cpu().numpy() - yref)))
    got = np.array([float(v) for v in grads])
    print(f"B {B} T {T}: |y - oracle| {e_y:.2e}, loss {float(loss):.6e} vs {np.mean((yref - tgt) ** 2):.6e}, grads {rel(got, gref):.2e}")
    assert e_y < 3e-6
    assert abs(float(loss) - np.mean((yref - tgt) ** 2)) < 1e-5 * np.mean((yref - tgt) ** 2)
    assert rel(got, gref) < 2e-4


def test_resident_two_state_ladder_two_sources_against_the_host_probe_path(wdf):
    """ns = 2, ni = 2 (a resistive source inside the ladder and the ideal-source root): the resident one-pass step against the plain path (host probe, forward + reverse-sweep kernels, torch autograd)."""
    tf = wdf.tf
    rng = np.random.default_rng(5)
    B, T = 130, 1500
    x = rng.standard_normal((B, T, 2)).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)

    def build():
        Ra = wdf.Resistor(1.0e3, True)
        Vr = wdf.ResistiveVoltageSource(2.2e3, trainable=True)
        Ca, Cb = wdf.Capacitor(1.0e-7, FS, True), wdf.Capacitor(2.2e-7, FS, True)
        top = wdf.Inverter(wdf.Series(Ra, wdf.Parallel(Ca, wdf.Series(Vr, Cb))))
        return wdf.Circuit(top, wdf.IdealVoltageSource(), Cb), [Ra.R, Vr.R, Ca.C, Cb.C]

    ref, pr = build()
    assert (ref.ns, ref.ni) == (2, 2)
    with tf.GradientTape() as tape:
        y = ref(cuda(x))
        l0 = tf.reduce_mean(tf.square(y - cuda(tgt)))
    g0 = np.array([float(v) for v in tape.gradient(l0, pr)])
    circ, p = build()
    circ.to_device()
    with tf.GradientTape() as tape:
        l1 = circ.mse(cuda(x), cuda(tgt))
    g1 = np.array([float(v) for v in tape.gradient(l1, p)])
    print(f"loss {float(l0):.6e} / {float(l1):.6e}; grads {g0} / {g1}")
    assert float((circ.last_output - y.as_subclass(torch.Tensor).detach()).abs().max()) < 3e-6
    assert abs(float(l1) - float(l0)) < 1e-5 * float(l0)
    assert rel(g1, g0) < 3e-4


def test_resident_rc_lowpass_training_loop_follows_the_host_loop(wdf):
    """lpf.py:77-99: two Adam optimizers (R and C), 30 epochs: the resident loop (no host round trip) and the plain loop end
    at the same component values."""
    tf = wdf.tf
    rng = np.random.default_rng(2)
    B, T = 64, 2048
    x = cuda(rng.standard_normal((B, T)))
    # target: the same circuit at the values lpf.py aims for (fc = 720 Hz)
    tr = wdf.Circuit(*(lambda Vs, R1, C1, I1: (I1, Vs, C1))(*build_lpf(wdf)))
    Vs_, R_, C_, I_ = build_lpf(wdf)
    with torch.no_grad():
        R_.R.fill_(315.0)
        C_.C.fill_(0.7e-6)
    tgt = wdf.Circuit(I_, Vs_, C_)(x).as_subclass(torch.Tensor).detach()
    ends = []
    for resident in (False, True):
        Vs, R1, C1, I1 = build_lpf(wdf)
        circ = wdf.Circuit(I1, Vs, C1)
        if resident:
            circ.to_device()
        oR, oC = tf.keras.optimizers.Adam(learning_rate=25.0), tf.keras.optimizers.Adam(learning_rate=1.0e-8)   # lpf.py:79-80
        losses = []
        for _ in range(30):
            with tf.GradientTape() as tape:
                loss = circ.mse(x, tgt)
            gC, gR = tape.gradient(loss, [C1.C, R1.R])
            oC.apply_gradients([(gC, C1.C)])
            oR.apply_gradients([(gR, R1.R)])
            losses.append(float(loss))
        ends.append((float(R1.R), float(C1.C), losses))
    (R0, C0, l0), (R1_, C1_, l1) = ends
    print(f"R {R0:.3f} / {R1_:.3f}  C {C0:.4e} / {C1_:.4e}  loss {l0[0]:.4e} -> {l0[-1]:.4e} / {l1[-1]:.4e}")
    assert l1[-1] < 0.5 * l1[0]
    assert abs(R1_ - R0) < 2e-3 * R0 and abs(C1_ - C0) < 2e-3 * C0


def test_resident_linear_tree_refuses_a_replaced_component(wdf):
    from wdf_hip import binding as wb
    tf = wdf.tf
    Vs, R1, C1, I1 = build_lpf(wdf)
    circ = wdf.Circuit(I1, Vs, C1).to_device()
    x, t = cuda(np.zeros((2, 64))), cuda(np.zeros((64, 2)))
    circ.mse(x, t)
    R1.set_resistance(tf.constant(500.0))                         # tf_wdf.py:51-52 style replacement
    with pytest.raises(wb.WdfHipError):
        circ.mse(x, t)


def _hpf(wdf, n_up=2, n_down=3):
    """HPFDiodeClipper.h:28-32: Parallel(R, Series(Vs, C)) + diode pair."""
    R = wdf.Resistor(33.0e3, True)
    Vs = wdf.ResistiveVoltageSource(1.0e3, trainable=True)
    C = wdf.Capacitor(22.0e-9, FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs, C))
    dp = wdf.DiodePair(top, 4.352e-9, Vt=25.85e-3, nDiodes=1.906, N_up=n_up, N_down=n_down, trainable=True)
    return wdf.Circuit(top, dp, R), [R.R, Vs.R, C.C, dp.Is, dp.nVt]


def test_resident_hpf_clipper_against_the_oracle(wdf, oracle):
    """A diode-root tree with its component values on the device: the probe (coefficients, port resistance and their
    chain rule to R, Rs, C) runs there; forward and the five gradients against the oracle."""
    tf = wdf.tf
    O = oracle
    rng = np.random.default_rng(12)
    B, T = 40, 700
    x = (rng.standard_normal((B, T)) * 1.2).astype(np.float32)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    circ, params = _hpf(wdf)
    circ.to_device()
    assert all(p.is_cuda for p in params)
    with tf.GradientTape() as tape:
        y = circ(cuda(x))
        loss = tf.reduce_sum(y * cuda(gy))
    grads = tape.gradient(loss, params)
    nodes = [(O.NODE_RESISTOR, -1, -1, 0, -1, -1), (O.NODE_RES_VSOURCE, -1, -1, 1, 0, -1),
             (O.NODE_CAPACITOR, -1, -1, 2, -1, -1), (O.NODE_SERIES, 1, 2, -1, -1, -1), (O.NODE_PARALLEL, 0, 3, -1, -1, -1)]
    oc = O.Circuit(nodes, top=4, probe=0, n_in=1, root_kind=O.ROOT_DIODE_PAIR, fs=FS, p_is=3, p_nvt=4, n_up=2, n_down=3)
    theta = np.array([33.0e3, 1.0e3, 22.0e-9, 4.352e-9, 25.85e-3 * 1.906], dtype=np.float32).astype(np.float64)
    yref = O.tree_fwd(oc, theta, x.astype(np.float64))
    gref = O.tree_grad(oc, theta, x.astype(np.float64), gy.astype(np.float64))
    got = np.array([float(v) for v in grads])
    e_y = float(np.max(np.abs(y.as_subclass(torch.Tensor).detach().cpu().numpy() - yref)))
    print(f"resident HPF clipper: |y - oracle| {e_y:.2e}, gradients {rel(got, gref):.2e}")
    assert e_y < 3e-6 and rel(got, gref) < 3e-4


def test_resident_hpf_clipper_training_loop_follows_the_host_loop(wdf):
    tf = wdf.tf
    B, T = 256, 2048
    x = cuda((np.random.default_rng(11).standard_normal((B, T)) * 1.2))
    ref, _ = _hpf(wdf)
    tgt = (ref(x) * 0.8).as_subclass(torch.Tensor).detach()
    ends = []
    for resident in (False, True):
        circ, params = _hpf(wdf)
        if resident:
            circ.to_device()
        opts = [tf.keras.optimizers.Adam(learning_rate=2.0e-3 * float(p)) for p in params]
        for _ in range(12):
            with tf.GradientTape() as tape:
                loss = circ.mse(x, tgt)
            grads = tape.gradient(loss, params)
            for o, g, p in zip(opts, grads, params):
                o.apply_gradients([(g, p)])
        ends.append(([float(p) for p in params], float(loss)))
    (p0, l0), (p1, l1) = ends
    print(f"host loop {p0} loss {l0:.4e}\\nresident  {p1} loss {l1:.4e}")
    assert np.allclose(p1, p0, rtol=5e-4, atol=0) and abs(l1 - l0) < 1e-3 * l0


def test_saved_losses_and_gradients_of_a_resident_loop_stay_what_they_were(wdf):
    """lpf.py:86-101 keeps every epoch's loss (`losses.append(loss)`) and prints `grads` after the loop: what mse() and
    tape.gradient hand out must not be overwritten by later calls on the same batch (each call's results live in a row of
    their own)."""
    tf = wdf.tf
    rng = np.random.default_rng(3)
    x = cuda(rng.standard_normal((64, 512)))
    tgt = cuda(0.3 * rng.standard_normal((512, 64)))
    Vs, R1, C1, I1 = build_lpf(wdf)
    circ = wdf.Circuit(I1, Vs, C1).to_device()
    opts = [tf.keras.optimizers.Adam(learning_rate=1e-3 * float(p)) for p in (C1.C, R1.R)]
    losses, first_grads, first_vals = [], None, None
    for epoch in range(24):
        with tf.GradientTape() as tape:
            loss = circ.mse(x, tgt)
        grads = tape.gradient(loss, [C1.C, R1.R])
        for o, g, p in zip(opts, grads, (C1.C, R1.R)):
            o.apply_gradients([(g, p)])
        losses.append(loss)
        if epoch == 0:
            first_grads, first_vals = grads, ([float(g) for g in grads], float(loss))
    vals = [float(v) for v in losses]
    assert len(set(vals)) == 24 and vals[-1] < vals[0]          # 24 distinct values, still a descending history
    assert float(losses[0]) == first_vals[1]
    assert [float(g) for g in first_grads] == first_vals[0]     # the first epoch's gradients, read after 23 more epochs
    # more calls than one slab of rows holds: the earliest results still stand
    from wdf_hip import lowering
    for _ in range(lowering._ROWS + 8):
        circ.mse(x, tgt)
    assert float(losses[0]) == first_vals[1] and [float(g) for g in first_grads] == first_vals[0]


def test_to_device_refuses_derived_values_and_notices_later_changes(wdf):
    """What the resident step would silently freeze is refused: a component computed from another Variable (no gradient
    would reach it), a non-Variable value replaced after to_device(); and a circuit the device probe cannot hold leaves its
    Variables where they were (the tape is recorded before anything is adopted)."""
    from wdf_hip import binding as wb, probe_tape
    tf = wdf.tf
    base = tf.Variable(1000.0, dtype=tf.float32)
    R1 = wdf.Resistor(1000.0, True)
    R1.R = base * 2.0                                            # derived: requires grad through `base`
    C1 = wdf.Capacitor(1.0e-6, FS, True)
    circ = wdf.Circuit(wdf.Inverter(wdf.Series(R1, C1)), wdf.IdealVoltageSource(), C1)
    with pytest.raises(wb.WdfHipError, match="computed from other Variables"):
        circ.to_device()
    assert getattr(C1.C, "_wdf_block", None) is None and not C1.C.is_cuda      # nothing was adopted on the way out
    # a frozen (non-trainable python number) value changed after to_device()
    R2 = wdf.Resistor(1000.0, False)
    R2.R = 1000.0
    C2 = wdf.Capacitor(1.0e-6, FS, True)
    c2 = wdf.Circuit(wdf.Inverter(wdf.Series(R2, C2)), wdf.IdealVoltageSource(), C2).to_device()
    x, tgt = cuda(np.ones((2, 64))), cuda(np.zeros((64, 2)))
    c2.mse(x, tgt)
    R2.R = 1000.0                                                # the same number again: fine
    c2.mse(x, tgt)
    R2.R = 2200.0
    with pytest.raises(wb.WdfHipError, match="changed after to_device"):
        c2.mse(x, tgt)
    # a tree with more component values than the device probe holds: refused, Variables untouched
    rs = [wdf.Resistor(100.0 * (i + 1), True) for i in range(probe_tape.MAX_PARAMS + 1)]
    top = rs[0]
    for r_ in rs[1:]:
        top = wdf.Series(top, r_)
    big = wdf.Circuit(wdf.Inverter(top), wdf.IdealVoltageSource(), rs[0])
    with pytest.raises(wb.WdfHipError):
        big.to_device()
    assert all(getattr(r_.R, "_wdf_block", None) is None and not r_.R.is_cuda for r_ in rs)
    ok = wdf.Circuit(wdf.Inverter(wdf.Series(rs[0], rs[1])), wdf.IdealVoltageSource(), rs[0]).to_device()   # ... and still adoptable
    assert rs[0].R.is_cuda and ok._lin is not None
