"""GPU: per-sample impedance and the MLP root on ANY small tree (csrc/wdf_ss_dyn.h, lowering.Circuit._run_dyn).

In the reference set_resistance replaces R on any ResistiveVoltageSource / Resistor (tf_wdf.py:51-52,80-81), calc_impedance is
legal every step on any tree (clipper_pot.py:116-117) and DenseRootModel terminates any tree (layers.py:72-82).  Held against
the oracle's tree interpreter (fp64; per-sample resistance channel `rin`, MLP root, complex-step gradients):
  * HPFDiodeClipper.h:28-32's tree Parallel(R, Series(ResistiveVoltageSource, C)) with a pot channel on the source resistance,
    diode-pair root: y and dL/d{R, C, Is, nVt};
  * the same tree with the pot on the parallel RESISTOR;
  * the same tree under a DenseRootModel root, static and with the pot channel: y, dL/d{R, C} and weight gradients;
  * the RC lowpass of lpf.py:20-29 (ideal-source root) with a per-sample resistor;
  * the clipper topology forced through these kernels against the clipper kernels' golden (g6, pot channel).
Tolerances: y 3e-6 V (fp32 recursion), gradients 3e-4 relative (fp32 sweep, rows formed in fp64 and rounded to fp32).
"""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000.0


def cuda(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


@pytest.fixture
def wdf():
    import tf_wdf
    return tf_wdf


def pot_channel(B, T, lo, hi, seed):
    """a slowly moving pot per sequence: log-uniform level, a slow sine on top (every sample its own resistance)"""
    rng = np.random.default_rng(seed)
    base = np.exp(rng.uniform(np.log(lo), np.log(hi), B))
    wob = 1.0 + 0.3 * np.sin(2 * np.pi * np.arange(T)[None, :] / rng.uniform(200, 900, B)[:, None] + rng.uniform(0, 6, B)[:, None])
    return (base[:, None] * wob).astype(np.float32)


def hpf_oracle(O, root, pot_on, fs=FS, sizes=None, acts=None):
    """nodes: R (param 0), Vs (param 1, voltage channel 0), C (param 2), Series(Vs, C), Parallel(R, Series)"""
    rin_r = 1 if pot_on == "R" else -1
    rin_s = 1 if pot_on == "Vs" else -1
    nodes = [(O.NODE_RESISTOR, -1, -1, 0, -1, rin_r), (O.NODE_RES_VSOURCE, -1, -1, 1, 0, rin_s), (O.NODE_CAPACITOR, -1, -1, 2, -1, -1),
             (O.NODE_SERIES, 1, 2, -1, -1, -1), (O.NODE_PARALLEL, 0, 3, -1, -1, -1)]
    n_in = 2 if pot_on else 1
    if root == "diode":
        return O.Circuit(nodes, top=4, probe=0, n_in=n_in, root_kind=O.ROOT_DIODE_PAIR, fs=fs, p_is=3, p_nvt=4, n_up=1, n_down=2)
    return O.Circuit(nodes, top=4, probe=0, n_in=n_in, root_kind=O.ROOT_MLP, fs=fs, mlp_off=3, mlp_sizes=sizes, mlp_act=acts)


def build_hpf(wdf, root, pot_on, vals, net=None):
    R = wdf.Resistor(vals[0], True)
    Vs = wdf.ResistiveVoltageSource(vals[1], trainable=True)
    C = wdf.Capacitor(vals[2], FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs, C))
    if root == "diode":
        rt = wdf.DiodePair(top, vals[3], Vt=vals[4], nDiodes=1.0, N_up=1, N_down=2, trainable=True)
        params = [R.R, Vs.R, C.C, rt.Is, rt.nVt]
    else:
        from layers import DenseRootModel
        rt = DenseRootModel(net)
        params = [R.R, Vs.R, C.C]
    pot = {"Vs": Vs, "R": R, None: None}[pot_on]
    return wdf.Circuit(top, rt, R, per_sample_R=pot), params, rt


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.abs(b)))


# (the last three: one sequence of one sample; fewer samples than a row block or a chunk unit; a second wave of one sequence)
@pytest.mark.parametrize("pot_on,B,T", [("Vs", 37, 300), ("R", 70, 257), ("Vs", 130, 1024), ("Vs", 1, 1), ("R", 3, 7), ("Vs", 65, 9)])
def test_hpf_clipper_with_a_pot_channel_diode_root(wdf, oracle, pot_on, B, T):
    tf = wdf.tf
    O = oracle
    vals = [33.0e3, 1.0e3, 22.0e-9, 4.352e-9, 25.85e-3 * 1.906]
    rng = np.random.default_rng(B + T)
    x = (1.5 * rng.standard_normal((B, T))).astype(np.float32)
    r = pot_channel(B, T, 300.0, 5.0e3, 1) if pot_on == "Vs" else pot_channel(B, T, 5.0e3, 80.0e3, 2)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    circ, params, _ = build_hpf(wdf, "diode", pot_on, vals)
    xin = cuda(np.stack([x, r], axis=-1))
    with tf.GradientTape() as tape:
        y = circ(xin)
        loss = tf.reduce_sum(y * cuda(gy))
    grads = tape.gradient(loss, params)
    oc = hpf_oracle(O, "diode", pot_on)
    theta = np.array(vals, dtype=np.float32).astype(np.float64)
    xin64 = np.stack([x, r], axis=-1).astype(np.float64)
    y_ref = O.tree_fwd(oc, theta, xin64)
    assert np.max(np.abs(y.cpu().numpy() - y_ref)) < 3e-6
    live = [i for i in range(5) if not (i == 0 and pot_on == "R") and not (i == 1 and pot_on == "Vs")]   # the streamed one has no gradient
    g_ref = O.tree_grad(oc, theta, xin64, gy.astype(np.float64), params=live)
    got = np.array([float(grads[i]) for i in live])
    assert rel(got, g_ref) < 3e-4, (got, g_ref)
    dead = 0 if pot_on == "R" else 1
    assert grads[dead] is None or float(grads[dead]) == 0.0


@pytest.mark.parametrize("root,pot_on,B,T", [("diode", "Vs", 70, 1024), ("diode", "R", 37, 300), ("mlp", "Vs", 40, 512), ("diode", "Vs", 130, 8)])
def test_a_pot_that_is_constant_along_each_sequence_takes_one_row_per_sequence(wdf, oracle, golden, root, pot_on, B, T):
    """The reference's recordings hold ONE pot value per file (dataimport.py:96 repeats it down the channel; batch_data,
    clipper_pot.py:61-80, cuts the sequences out of it): the resistance channel is constant along every sequence.  The lowering
    notices (one comparison pass per input tensor) and hands the kernels one coefficient row per SEQUENCE -- rows [1,n,B], the
    tape run over B values instead of B x T, dL/d(row) summed over the steps inside the sweep (wdf_ss_dyn_bwd, per_sample = 2).
    Against the oracle (y, every gradient), and against the per-sample path on the same data (y bit for bit)."""
    tf = wdf.tf
    O = oracle
    rng = np.random.default_rng(B + T + 5)
    x = (1.5 * rng.standard_normal((B, T))).astype(np.float32)
    grid = np.array([300.0, 1.0e3, 2.5e3, 5.0e3]) if pot_on == "Vs" else np.array([10.0e3, 25.2e3, 45.2e3, 75.0e3, 99.1e3])
    r = np.repeat(grid[np.arange(B) % len(grid)][:, None], T, axis=1).astype(np.float32)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    if root == "diode":
        vals = [33.0e3, 1.0e3, 22.0e-9, 4.352e-9, 25.85e-3 * 1.906]
        net, oc, theta = None, hpf_oracle(O, "diode", pot_on), None
        theta = np.array(vals, dtype=np.float32).astype(np.float64)
    else:
        vals = [33.0e3, 1.0e3, 22.0e-9]
        net, wts, sizes = _net(golden, "2x8")
        acts = [O.ACT_TANH] * (len(sizes) - 2) + [O.ACT_NONE]
        oc = hpf_oracle(O, "mlp", pot_on, sizes=sizes, acts=acts)
        theta = np.concatenate([np.array(vals, dtype=np.float32).astype(np.float64), wts.astype(np.float32).astype(np.float64)])
    xin = cuda(np.stack([x, r], axis=-1))
    out = {}
    for per_seq in (True, False):
        circ, params, rt = build_hpf(wdf, root, pot_on, vals, net)
        circ.per_sequence_rows = per_seq
        with tf.GradientTape() as tape:
            y = circ(xin)
            loss = tf.reduce_sum(y * cuda(gy))
        grads = tape.gradient(loss, params)
        out[per_seq] = (y.as_subclass(torch.Tensor).detach().clone(), [None if g is None else float(g) for g in grads])
        if per_seq and T > 1:
            assert list(circ._dyn_chan_const.values()) == [True]
    assert torch.equal(out[True][0], out[False][0])                       # the same coefficients at every step: the same y
    xin64 = np.stack([x, r], axis=-1).astype(np.float64)
    y_ref = O.tree_fwd(oc, theta, xin64)
    assert np.max(np.abs(out[True][0].cpu().numpy() - y_ref)) < 3e-6
    n_comp = 5 if root == "diode" else 3
    live = [i for i in range(n_comp) if not (i == 0 and pot_on == "R") and not (i == 1 and pot_on == "Vs")]
    g_ref = O.tree_grad(oc, theta, xin64, gy.astype(np.float64), params=live)
    got = np.array([out[True][1][i] for i in live])
    got_ps = np.array([out[False][1][i] for i in live])
    print(f"{root} pot on {pot_on} {B} x {T}: gradients vs oracle {rel(got, g_ref):.2e} (per-sample path {rel(got_ps, g_ref):.2e})")
    assert rel(got, g_ref) < 3e-4, (got, g_ref)
    assert rel(got, got_ps) < 1e-4


@pytest.mark.parametrize("root,n_sec,B,T,resident", [("ideal", 6, 70, 600, False), ("diode", 6, 37, 300, False), ("ideal", 8, 130, 1024, False),
                                                     ("diode", 7, 24, 257, False), ("diode", 5, 70, 512, True)])
def test_trees_of_five_to_eight_capacitors_against_the_oracle(wdf, oracle, root, n_sec, B, T, resident):
    """Round 6: trees beyond four capacitors (tf_wdf.py:129-192 composes any binary tree; the static-coefficient kernels are
    compiled for at most four states) run on the streamed-coefficient kernels with ONE static row, compiled for eight state
    slots.  An RC ladder of n_sec sections -- Series(R_k, Parallel(C_k, rest)), the last section Series(R, C) -- under the ideal
    source (through an Inverter, as lpf.py:28 connects it) or behind a resistive source in front of a diode pair: y and the
    gradient of every component value against the oracle's tree interpreter (fp64, complex step).  resident: the same with the
    component values in a device block (Circuit.to_device)."""
    tf = wdf.tf
    O = oracle
    rng = np.random.default_rng(B + T + n_sec)
    x = (1.2 * rng.standard_normal((B, T))).astype(np.float32)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    rv = [1.0e3 * (1.0 + 0.7 * k) for k in range(n_sec)]
    cv = [1.0e-7 / (1.0 + 0.5 * k) for k in range(n_sec)]
    Rs = [wdf.Resistor(v, True) for v in rv]
    Cs = [wdf.Capacitor(v, FS, True) for v in cv]
    sec = wdf.Series(Rs[-1], Cs[-1])
    for k in range(n_sec - 2, -1, -1):
        sec = wdf.Series(Rs[k], wdf.Parallel(Cs[k], sec))
    nodes = []

    def leaf(kind, param, vin=-1):
        nodes.append((kind, -1, -1, param, vin, -1))
        return len(nodes) - 1

    def join(kind, a, b=-1):
        nodes.append((kind, a, b, -1, -1, -1))
        return len(nodes) - 1

    r = [leaf(O.NODE_RESISTOR, 2 * k) for k in range(n_sec)]
    c = [leaf(O.NODE_CAPACITOR, 2 * k + 1) for k in range(n_sec)]
    s_ = join(O.NODE_SERIES, r[-1], c[-1])
    for k in range(n_sec - 2, -1, -1):
        s_ = join(O.NODE_SERIES, r[k], join(O.NODE_PARALLEL, c[k], s_))
    theta = [v for pair in zip(rv, cv) for v in pair]
    params = [p for pair in zip([e.R for e in Rs], [e.C for e in Cs]) for p in pair]
    if root == "ideal":
        circ = wdf.Circuit(wdf.Inverter(sec), wdf.IdealVoltageSource(), Cs[-1])
        top = join(O.NODE_INVERTER, s_)
        oc = O.Circuit(nodes, top=top, probe=c[-1], n_in=1, root_kind=O.ROOT_IDEAL_VSOURCE, fs=FS, root_vin=0)
    else:
        Vs = wdf.ResistiveVoltageSource(2.2e3, trainable=True)
        topw = wdf.Parallel(Vs, sec)
        dp = wdf.DiodePair(topw, 4.352e-9, Vt=25.85e-3 * 1.906, nDiodes=1.0, trainable=True)
        circ = wdf.Circuit(topw, dp, Cs[-1])
        v = leaf(O.NODE_RES_VSOURCE, 2 * n_sec, vin=0)
        top = join(O.NODE_PARALLEL, v, s_)
        oc = O.Circuit(nodes, top=top, probe=c[-1], n_in=1, root_kind=O.ROOT_DIODE_PAIR, fs=FS, p_is=2 * n_sec + 1, p_nvt=2 * n_sec + 2,
                       n_up=1, n_down=1)
        theta = theta + [2.2e3, 4.352e-9, 25.85e-3 * 1.906]
        params = params + [Vs.R, dp.Is, dp.nVt]
    assert circ.ns == n_sec and circ._dyn
    if resident:
        circ.to_device()
        assert all(p.is_cuda for p in params)
    theta = np.array(theta, dtype=np.float32).astype(np.float64)
    with tf.GradientTape() as tape:
        y = circ(cuda(x))
        loss = tf.reduce_sum(y * cuda(gy))
    grads = tape.gradient(loss, params)
    x64 = x.astype(np.float64)
    e_y = float(np.max(np.abs(y.cpu().numpy() - O.tree_fwd(oc, theta, x64))))
    g_ref = O.tree_grad(oc, theta, x64, gy.astype(np.float64))
    got = np.array([float(g) for g in grads])
    print(f"{root} root, {n_sec} capacitors, {B} x {T}: |y - oracle| {e_y:.2e}, gradients {rel(got, g_ref):.2e}")
    assert e_y < 5e-6
    assert rel(got, g_ref) < 5e-4, (got, g_ref)


def _net(golden, name):
    from test_gpu_mlp_root import model_json
    g = golden("g3_mlp_clipper.npz")
    return model_json(g, name), g[f"{name}_theta"].astype(np.float64), [int(v) for v in g[f"{name}_sizes"]]


@pytest.mark.parametrize("pot_on,name", [(None, "2x8"), ("Vs", "2x8"), ("Vs", "2x16"), ("R", "4x4")])
def test_hpf_tree_under_an_mlp_root(wdf, oracle, golden, pot_on, name):
    """DenseRootModel on a tree that is NOT the clipper: b = -MLP(a, log R_port) (clipper_pot.py:119-121), R_port per sample
    when a pot channel moves it.  y, dL/d{R, Rs, C} and the weight gradient (every bias, first and last layer, every 7th
    hidden-kernel entry) against the oracle's complex-step derivative."""
    tf = wdf.tf
    O = oracle
    js, w64, sizes = _net(golden, name)
    B, T = 24, 200
    vals = [33.0e3, 4.0e3, 22.0e-9]
    rng = np.random.default_rng(len(name) + (0 if pot_on is None else 5))
    x = (0.8 * rng.standard_normal((B, T))).astype(np.float32)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    circ, params, model = build_hpf(wdf, "mlp", pot_on, vals, net=js)
    if pot_on is None:
        xin, xin64 = cuda(x), x.astype(np.float64)[:, :, None]
    else:
        r = pot_channel(B, T, 1.0e3, 20.0e3, 3) if pot_on == "Vs" else pot_channel(B, T, 8.0e3, 90.0e3, 4)
        xin, xin64 = cuda(np.stack([x, r], axis=-1)), np.stack([x, r], axis=-1).astype(np.float64)
    weights = list(model.trainable_variables)
    with tf.GradientTape() as tape:
        y = circ(xin)
        loss = tf.reduce_sum(y * cuda(gy))
    grads = tape.gradient(loss, params + weights)
    acts = [O.ACT_TANH] * (len(sizes) - 2) + [O.ACT_NONE]
    oc = hpf_oracle(O, "mlp", pot_on, sizes=sizes, acts=acts)
    theta = np.concatenate([np.array(vals, dtype=np.float32).astype(np.float64), w64.astype(np.float32).astype(np.float64)])
    y_ref = O.tree_fwd(oc, theta, xin64)
    assert np.max(np.abs(y.cpu().numpy() - y_ref)) < 3e-6
    live = [i for i in range(3) if not (i == 0 and pot_on == "R") and not (i == 1 and pot_on == "Vs")]
    g_ref = O.tree_grad(oc, theta, xin64, gy.astype(np.float64), params=live)
    got = np.array([float(grads[i]) for i in live])
    assert rel(got, g_ref) < 3e-4, (got, g_ref)
    # weights: the flat order of the oracle is kernel, bias per layer; trainable_variables lists bias before kernel per layer
    flat = {}
    o = 3
    for l in range(len(sizes) - 1):
        nk = sizes[l] * sizes[l + 1]
        flat[("k", l)] = (o, nk); o += nk
        flat[("b", l)] = (o, sizes[l + 1]); o += sizes[l + 1]
    dense = [d for d in model.layers if type(d).__name__ == "DenseLayer"]
    gw = np.zeros(len(theta))
    gmap = {id(v): g for v, g in zip(weights, grads[3:])}
    for l, d in enumerate(dense):
        ok, nk = flat[("k", l)]
        gw[ok:ok + nk] = gmap[id(d.kernel)].cpu().numpy().reshape(-1)
        ob, nb = flat[("b", l)]
        gw[ob:ob + nb] = gmap[id(d.bias)].cpu().numpy().reshape(-1)
    pick = sorted(set(list(range(3, 3 + 3 * sizes[1])) + list(range(len(theta) - sizes[-2] - 1, len(theta))) + list(range(3, len(theta), 7))))
    gw_ref = O.tree_grad(oc, theta, xin64, gy.astype(np.float64), params=pick)
    scale = np.max(np.abs(gw_ref))
    assert np.max(np.abs(gw[pick] - gw_ref)) < 3e-4 * scale, (np.max(np.abs(gw[pick] - gw_ref)), scale)


def test_rc_lowpass_with_a_per_sample_resistor(wdf, oracle):
    """lpf.py:20-29's tree Inverter(Series(R1, C1)) under the ideal source, R1 driven per sample (Resistor.set_resistance,
    tf_wdf.py:80-81): the ideal-source root folds into the rows like it folds into the static matrices."""
    tf = wdf.tf
    O = oracle
    B, T = 50, 400
    rng = np.random.default_rng(7)
    x = rng.standard_normal((B, T)).astype(np.float32)
    r = pot_channel(B, T, 300.0, 3.0e3, 9)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    R1, C1 = wdf.Resistor(1000.0, True), wdf.Capacitor(1.0e-6, FS, True)
    circ = wdf.Circuit(wdf.Inverter(wdf.Series(R1, C1)), wdf.IdealVoltageSource(), C1, per_sample_R=R1)
    with tf.GradientTape() as tape:
        y = circ(cuda(np.stack([x, r], axis=-1)))
        loss = tf.reduce_sum(y * cuda(gy))
    gC = tape.gradient(loss, [C1.C])[0]
    nodes = [(O.NODE_RESISTOR, -1, -1, 0, -1, 1), (O.NODE_CAPACITOR, -1, -1, 1, -1, -1), (O.NODE_SERIES, 0, 1, -1, -1, -1),
             (O.NODE_INVERTER, 2, -1, -1, -1, -1)]
    oc = O.Circuit(nodes, top=3, probe=1, n_in=2, root_kind=O.ROOT_IDEAL_VSOURCE, fs=FS, root_vin=0)
    theta = np.array([1000.0, np.float32(1.0e-6)], dtype=np.float64)
    xin64 = np.stack([x, r], axis=-1).astype(np.float64)
    assert np.max(np.abs(y.cpu().numpy() - O.tree_fwd(oc, theta, xin64))) < 3e-6
    g_ref = O.tree_grad(oc, theta, xin64, gy.astype(np.float64), params=[1])
    assert rel([float(gC)], g_ref) < 3e-4


def test_three_state_tree_with_a_pot_and_an_mlp_root(wdf, oracle, golden):
    """Beyond two states: input coupling R0-C0, a shelving section R1 || C1, the pot Rp in series with C2, under a
    DenseRootModel -- ns = 3, a per-sample resistor AND the network root on one tree; y and dL/d{R0, C0, R1, C1, C2}
    against the oracle."""
    tf = wdf.tf
    O = oracle
    from layers import DenseRootModel
    js, w64, sizes = _net(golden, "2x8")
    B, T = 20, 180
    rng = np.random.default_rng(33)
    x = (0.8 * rng.standard_normal((B, T))).astype(np.float32)
    r = pot_channel(B, T, 2.0e3, 50.0e3, 6)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    Vs = wdf.ResistiveVoltageSource(2.2e3, trainable=True)
    R1 = wdf.Resistor(15.0e3, True)
    Rp = wdf.Resistor(10.0e3, True)
    C0, C1, C2 = wdf.Capacitor(47.0e-9, FS, True), wdf.Capacitor(10.0e-9, FS, True), wdf.Capacitor(22.0e-9, FS, True)
    top = wdf.Parallel(wdf.Series(Rp, C2), wdf.Series(wdf.Series(Vs, C0), wdf.Parallel(R1, C1)))
    model = DenseRootModel(js)
    circ = wdf.Circuit(top, model, C2, per_sample_R=Rp)
    assert (circ.ns, circ.ni) == (3, 1)
    params = [Vs.R, R1.R, C0.C, C1.C, C2.C]
    with tf.GradientTape() as tape:
        y = circ(cuda(np.stack([x, r], axis=-1)))
        loss = tf.reduce_sum(y * cuda(gy))
    grads = tape.gradient(loss, params)
    # theta = [Rs, R1, Rp (streamed: channel 1), C0, C1, C2, weights...]
    nodes = [(O.NODE_RES_VSOURCE, -1, -1, 0, 0, -1), (O.NODE_RESISTOR, -1, -1, 1, -1, -1), (O.NODE_RESISTOR, -1, -1, 2, -1, 1),
             (O.NODE_CAPACITOR, -1, -1, 3, -1, -1), (O.NODE_CAPACITOR, -1, -1, 4, -1, -1), (O.NODE_CAPACITOR, -1, -1, 5, -1, -1),
             (O.NODE_SERIES, 2, 5, -1, -1, -1),                    # 6: Rp - C2
             (O.NODE_SERIES, 0, 3, -1, -1, -1),                    # 7: Vs - C0
             (O.NODE_PARALLEL, 1, 4, -1, -1, -1),                  # 8: R1 || C1
             (O.NODE_SERIES, 7, 8, -1, -1, -1),                    # 9
             (O.NODE_PARALLEL, 6, 9, -1, -1, -1)]                  # 10: top
    acts = [O.ACT_TANH] * (len(sizes) - 2) + [O.ACT_NONE]
    oc = O.Circuit(nodes, top=10, probe=5, n_in=2, root_kind=O.ROOT_MLP, fs=FS, mlp_off=6, mlp_sizes=sizes, mlp_act=acts)
    theta = np.concatenate([np.array([2.2e3, 15.0e3, 10.0e3, 47.0e-9, 10.0e-9, 22.0e-9], dtype=np.float32).astype(np.float64),
                            w64.astype(np.float32).astype(np.float64)])
    xin64 = np.stack([x, r], axis=-1).astype(np.float64)
    assert np.max(np.abs(y.cpu().numpy() - O.tree_fwd(oc, theta, xin64))) < 3e-6
    g_ref = O.tree_grad(oc, theta, xin64, gy.astype(np.float64), params=[0, 1, 3, 4, 5])
    assert rel(np.array([float(g) for g in grads]), g_ref) < 3e-4


def test_clipper_topology_through_the_streamed_kernels_equals_the_clipper_kernels(wdf, golden):
    """force_generic: Parallel(ResistiveVoltageSource, Capacitor) + diode pair + pot channel through wdf_ss_dyn_* against the
    golden of the clipper kernels' pot path (g6 rpot: fp64 restatement, Newton-verified) and against those kernels."""
    tf = wdf.tf
    g = golden("g6_diode_clipper.npz")
    Is, nVt, R, C = (float(v) for v in g["theta"])
    x, r = g["x"].astype(np.float32), g["r"].astype(np.float32)

    def build(generic):
        Vs = wdf.ResistiveVoltageSource(R)
        Cap = wdf.Capacitor(C, FS, trainable=True)
        P1 = wdf.Parallel(Vs, Cap)
        dp = wdf.DiodePair(P1, Is, Vt=nVt, trainable=True)
        return wdf.Circuit(P1, dp, Cap, per_sample_R=Vs, force_generic=generic), [dp.Is, dp.nVt, Cap.C]

    xin = cuda(np.stack([x, r], axis=-1))
    outs = []
    for generic in (True, False):
        circ, params = build(generic)
        with tf.GradientTape() as tape:
            y = circ(xin)
            loss = tf.reduce_mean(tf.square(y))
        outs.append((y.cpu().numpy(), np.array([float(v) for v in tape.gradient(loss, params)])))
    assert np.max(np.abs(outs[0][0] - g["y_1u1d_rpot_f64"])) < 3e-6
    assert np.max(np.abs(outs[0][0] - outs[1][0])) < 3e-6
    assert rel(outs[0][1], outs[1][1]) < 3e-4


def test_what_the_streamed_kernels_refuse(wdf, golden):
    from wdf_hip import binding as wb
    from layers import DenseRootModel
    R = [wdf.Resistor(1.0e3 * (i + 1)) for i in range(4)]
    caps = [wdf.Capacitor(1.0e-8 * (i + 1), FS) for i in range(5)]
    Vs = wdf.ResistiveVoltageSource(1.0e3)
    ladder = wdf.Series(Vs, caps[4])
    for k in range(4):
        ladder = wdf.Series(wdf.Parallel(caps[k], R[k]), ladder)
    top = ladder
    dp = wdf.DiodePair(top, 4.352e-9, Vt=0.049)
    c5 = wdf.Circuit(top, dp, caps[0], per_sample_R=Vs)           # five states: fine since round 6 (up to eight)
    assert c5.ns == 5
    big = wdf.Series(wdf.ResistiveVoltageSource(1e3), wdf.Capacitor(1e-8, FS))
    for k in range(8):
        big = wdf.Series(wdf.Parallel(wdf.Capacitor(1e-8 * (k + 1), FS), wdf.Resistor(1e3 * (k + 1))), big)
    with pytest.raises(wb.WdfHipError, match="eight capacitors"):
        wdf.Circuit(big, wdf.DiodePair(big, 4.352e-9, Vt=0.049), big.P2)   # nine states
    with pytest.raises(ValueError):
        wdf.Circuit(wdf.Parallel(wdf.Resistor(1e3), wdf.Series(wdf.ResistiveVoltageSource(1e3), wdf.Capacitor(1e-8, FS))), dp, caps[0])
    js, _, _ = _net(golden, "2x8")
    for layer in js["layers"][:-1]:
        layer["activation"] = "relu"
    Rr, Vr, Cr = wdf.Resistor(33e3), wdf.ResistiveVoltageSource(1e3), wdf.Capacitor(22e-9, FS)
    circ = wdf.Circuit(wdf.Parallel(Rr, wdf.Series(Vr, Cr)), DenseRootModel(js), Rr)
    with pytest.raises(wb.WdfHipError, match="tanh"):
        circ(cuda(np.zeros((2, 16))))
    with pytest.raises(wb.WdfHipError, match="tanh"):             # (to_device() of a streamed-coefficient circuit: round 6)
        circ.to_device()


@pytest.mark.parametrize("root", ["diode", "mlp"])
def test_streamed_coefficient_circuit_trains_with_resident_components(wdf, golden, root):
    """Circuit.to_device() on HPFDiodeClipper.h's tree with a pot channel (round 6): the component Variables (and the diode's
    Is, nVt / the network's weights) live on the device, tape.gradient hands back device tensors, tf.keras.optimizers.Adam
    updates them there.  lpf.py:86-99's loop shape (one Adam per component) for eight epochs ends at the parameters the same
    loop reaches with host-resident Variables."""
    tf = wdf.tf
    B, T = 48, 512
    rng = np.random.default_rng(11)
    x = (1.2 * rng.standard_normal((B, T))).astype(np.float32)
    grid = np.array([300.0, 1.0e3, 2.5e3, 5.0e3])
    r = np.repeat(grid[np.arange(B) % 4][:, None], T, axis=1).astype(np.float32)
    tgt = cuda(0.2 * rng.standard_normal((T, B)))
    xin = cuda(np.stack([x, r], axis=-1))
    vals = [33.0e3, 1.0e3, 22.0e-9, 4.352e-9, 25.85e-3 * 1.906] if root == "diode" else [33.0e3, 1.0e3, 22.0e-9]
    net = _net(golden, "2x8")[0] if root == "mlp" else None
    ends = {}
    for resident in (False, True):
        circ, params, rt = build_hpf(wdf, root, "Vs", vals, net)
        train = [p for i, p in enumerate(params) if i != 1]                       # (the source resistance is the streamed one)
        weights = list(rt.trainable_variables) if root == "mlp" else []
        if resident:
            circ.to_device()
            assert all(p.is_cuda for p in train + weights)
        opts = [tf.keras.optimizers.Adam(learning_rate=1.0e-3 * abs(float(p))) for p in train]
        wopt = tf.keras.optimizers.Adam(learning_rate=1.0e-4) if weights else None
        losses = []
        for epoch in range(8):
            with tf.GradientTape() as tape:
                y = circ(xin)
                loss = tf.reduce_mean(tf.square(y - tgt))
            g = tape.gradient(loss, train + weights)
            for o, gi, p in zip(opts, g, train):
                o.apply_gradients([(gi, p)])
            if weights:
                wopt.apply_gradients(zip(g[len(train):], weights))
            losses.append(float(loss))
        ends[resident] = (np.array([float(p) for p in train]), [w.detach().cpu().numpy().copy() for w in weights], losses)
    a, b = ends[False], ends[True]
    print(f"{root}: losses host {a[2][0]:.6e} -> {a[2][-1]:.6e}, resident {b[2][0]:.6e} -> {b[2][-1]:.6e}; components differ by {rel(b[0], a[0]):.2e}")
    assert rel(b[0], a[0]) < 2e-5 and abs(b[2][-1] - a[2][-1]) < 1e-4 * a[2][-1]
    assert np.max(np.abs(a[0] / np.array([v for i, v in enumerate(vals) if i != 1]) - 1.0)) > 1e-3   # (they really moved)
    for wa, wb_ in zip(a[1], b[1]):
        assert np.max(np.abs(wa - wb_)) < 2e-6


@pytest.mark.parametrize("root,pot_on,B,T,fast", [("diode", "Vs", 70, 1000, True), ("mlp", "Vs", 40, 515, True), ("mlp", None, 130, 2048, False),
                                                  ("diode", "R", 5, 131, True), ("diode", "Vs", 130, 2048, False)])
def test_streamed_kernels_in_time_chunks_equal_the_sequential_ones(wdf, golden, root, pot_on, B, T, fast):
    """wdf_ss_dyn_fwd_tp / _bwd_tp (round 5): the forward in verified chunks gives the sequential kernel's y within the verified
    tolerance with a clean verdict, the chunked reverse sweep its gradients up to fp32 summation order; a warm-up that is far
    too short is noticed on the device and the waves concerned are re-run sequentially (identical outputs then)."""
    from wdf_hip import binding as wb, lowering
    tf = wdf.tf
    js = _net(golden, "2x8")[0] if root == "mlp" else None
    # fast: a small capacitor -- the tree forgets its state in a few dozen steps, the planner cuts the forward into chunks;
    # else HPFDiodeClipper.h's values: with the diodes off the state decays by 2.7 % per step (672 warm-up steps), the forward
    # stays sequential (or nearly) and only the reverse sweep is chunked
    vals = [3.3e3, 1.0e3, 1.0e-9, 4.352e-9, 25.85e-3 * 1.906] if fast else [33.0e3, 1.0e3, 22.0e-9, 4.352e-9, 25.85e-3 * 1.906]
    rng = np.random.default_rng(B + T)
    x = (1.2 * rng.standard_normal((B, T))).astype(np.float32)
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    if pot_on is None:
        xin = cuda(x)
    else:
        r = pot_channel(B, T, 300.0, 5.0e3, 1) if pot_on == "Vs" else pot_channel(B, T, 2.0e3, 30.0e3, 2)
        xin = cuda(np.stack([x, r], axis=-1))

    def run(time_parallel):
        circ, params, model = build_hpf(wdf, root, pot_on, vals, net=js)
        circ.time_parallel = time_parallel
        plist = [p for i, p in enumerate(params) if not (i == 0 and pot_on == "R") and not (i == 1 and pot_on == "Vs")]
        if root == "mlp":
            plist = plist + list(model.trainable_variables)
        lowering.LAST_SS_TP_STATUS["status"] = None
        with tf.GradientTape() as tape:
            y = circ(xin)
            loss = tf.reduce_sum(y * cuda(gy))
        g = tape.gradient(loss, plist)
        st = lowering.LAST_SS_TP_STATUS["status"]
        return y.cpu().numpy(), np.concatenate([v.cpu().numpy().reshape(-1) for v in g]), (None if st is None else wb.ss_tp_status(st))

    y_seq, g_seq, st_seq = run(None)
    y_tp, g_tp, st_tp = run("auto")
    assert st_seq is None
    if root == "diode" and fast and T >= 512:
        assert st_tp is not None and lowering.LAST_SS_TP_STATUS["chunks_used"] >= 2
    if st_tp is not None:
        assert st_tp["n_bad"] == 0 and st_tp["max_miss"] <= 1e-6, st_tp
    assert np.max(np.abs(y_tp - y_seq)) <= 2e-6
    scale = np.max(np.abs(g_seq))
    assert np.max(np.abs(g_tp - g_seq) / (np.abs(g_seq) + 1e-3 * scale)) < 5e-4
    if T >= 512 and not fast:
        k = wb.dyn_chunks(T, 8)
        y_short, g_short, st_short = run(lowering.SsTpPlan(k, 8, 1.0e-6, k))      # 8 steps cannot forget the capacitor's state
        assert st_short["n_bad"] > 0 and st_short["gated_waves"] >= 1, st_short
        assert np.array_equal(y_short, y_seq)


def test_network_root_slope_sizes_the_planned_warm_up(wdf, golden):
    """lowering._plan_dyn under a network root: the warm-up outlasts the slowest mode of A + Da E ca^T with Da taken from the
    WEIGHTS over the batch's amplitude (mlp_root.slope_range), not from a diode's |db/da| <= 1.  The pretrained 2x8 network and
    the same network with its output layer scaled by 3 (a root three times as steep: the state is forgotten more slowly, or not
    at all) must get different plans; both still give the sequential kernel's outputs."""
    from wdf_hip import binding as wb, lowering, mlp_root
    tf = wdf.tf
    js = _net(golden, "2x8")[0]
    vals = [3.3e3, 1.0e3, 1.0e-9, 4.352e-9, 25.85e-3 * 1.906]
    B, T = 64, 1024
    rng = np.random.default_rng(11)
    xin = cuda((1.2 * rng.standard_normal((B, T))).astype(np.float32))
    plans, slopes = [], []
    for scale in (1.0, 3.0):
        outs = []
        for tp in (None, "auto"):
            circ, params, model = build_hpf(wdf, "mlp", None, vals, net=js)
            last = [l for l in model.layers if type(l).__name__ == "DenseLayer"][-1]
            with torch.no_grad():
                last.kernel.mul_(scale)
                last.bias.mul_(scale)
            circ.time_parallel = tp
            outs.append(circ(xin).cpu().numpy())
            if tp == "auto":
                plans.append(next(iter(circ._dyn_plans.values()))[0])
                dense = mlp_root.describe(model, with_activation=True)[0]
                slopes.append(mlp_root.slope_range(dense, [math.log(1.0e3)], 4.0 * float(xin.abs().max())))
        assert np.max(np.abs(outs[0] - outs[1])) <= 2e-6, scale
    assert abs(slopes[1][1] - 3.0 * slopes[0][1]) < 1e-6 * abs(slopes[1][1])      # (the weights are scaled in float32)
    assert plans[0] != plans[1], (plans, slopes)
    assert plans[1].warmup == 0 or plans[1].warmup > plans[0].warmup, (plans, slopes)     # (0: no contraction -> one chunk)


@pytest.mark.parametrize("root", ["diode", "mlp"])
def test_streamed_forward_starts_warm_when_a_batch_is_visited_again(wdf, golden, root):
    """A training loop on HPFDiodeClipper.h's tree with a pot channel (the tree whose cold warm-up, 672 steps, outlasts its
    chunks): the first call runs sequentially, later calls cut the forward into chunks started from the previous call's states
    (lowering.DynWarmStart) -- same losses and gradients as the same loop on the sequential kernels, every verdict clean or
    repaired, and the chunked calls really happened."""
    from wdf_hip import binding as wb, lowering
    tf = wdf.tf
    js = _net(golden, "2x8")[0] if root == "mlp" else None
    vals = [33.0e3, 1.0e3, 22.0e-9, 4.352e-9, 25.85e-3 * 1.906]
    B, T = 96, 2048
    rng = np.random.default_rng(5)
    x = (1.2 * rng.standard_normal((B, T))).astype(np.float32)
    r = pot_channel(B, T, 300.0, 5.0e3, 1)
    xin = cuda(np.stack([x, r], axis=-1))
    tgt = cuda(0.2 * rng.standard_normal((T, B)))

    def loop(time_parallel, steps=14):
        circ, params, model = build_hpf(wdf, root, "Vs", vals, net=js)
        circ.time_parallel = time_parallel
        plist = [params[0], params[2]] + (params[3:] if root == "diode" else list(model.trainable_variables))
        opts = [tf.keras.optimizers.Adam(learning_rate=(1.0e-3 * abs(float(p)) if p.numel() == 1 else 1.0e-4)) for p in plist]
        out, chunks = [], []
        for _ in range(steps):
            lowering.LAST_SS_TP_STATUS["status"] = None
            with tf.GradientTape() as tape:
                y = circ(xin)
                loss = tf.reduce_mean(tf.square(y - tgt))
            g = tape.gradient(loss, plist)
            st = lowering.LAST_SS_TP_STATUS["status"]
            chunks.append(0 if st is None else lowering.LAST_SS_TP_STATUS["chunks_used"])
            out.append((float(loss), np.concatenate([v.cpu().numpy().reshape(-1) for v in g])))
            for o, gi, p in zip(opts, g, plist):
                o.apply_gradients([(gi, p)])
        return out, chunks

    seq, c_seq = loop(None)
    warm, c_warm = loop("auto")
    assert all(c == 0 for c in c_seq)
    assert c_warm[0] <= 3 and min(c_warm[2:]) >= 2, c_warm                      # chunked from the second or third call on
    for (ls, gs), (lw, gw) in zip(seq, warm):
        assert abs(lw - ls) <= 2e-5 * ls
        scale = np.max(np.abs(gs))
        assert np.max(np.abs(gw - gs) / (np.abs(gs) + 1e-3 * scale)) < 2e-3


@pytest.mark.parametrize("tree,B,T", [("hpf", 70, 33), ("three", 130, 9), ("static", 1, 1)])
def test_rows_made_on_the_device_equal_the_tape_in_float64(wdf, tree, B, T):
    """wdf_ss_dyn_rows / _bwd (csrc/wdf_ss_dyn_rows.h) against the same tape run by torch in float64 with autograd: every row
    entry of every sample, and dLoss/d(component value) for a random row adjoint.  (The end-to-end tests above go through
    these kernels too; this one holds them alone, to float32 rounding.)"""
    from wdf_hip import binding as wb, lowering, probe_tape
    from tf_wdf import DiodePair
    if tree in ("hpf", "static"):
        circ, _, _ = build_hpf(wdf, "diode", "Vs" if tree == "hpf" else None, [33.0e3, 1.0e3, 22.0e-9, 4.0e-9, 0.049])
        chan_el = circ.per_sample_R
    else:
        Vs = wdf.ResistiveVoltageSource(2.2e3, trainable=True)
        R1, Rp = wdf.Resistor(15.0e3, True), wdf.Resistor(10.0e3, True)
        C0, C1, C2 = wdf.Capacitor(47.0e-9, FS, True), wdf.Capacitor(10.0e-9, FS, True), wdf.Capacitor(22.0e-9, FS, True)
        top = wdf.Parallel(wdf.Series(Rp, C2), wdf.Series(wdf.Series(Vs, C0), wdf.Parallel(R1, C1)))
        circ = wdf.Circuit(top, DiodePair(top, 4.0e-9, Vt=0.049, trainable=True), C2, per_sample_R=Rp)
        chan_el = Rp
    own = {"Resistor": "R", "ResistiveVoltageSource": "R", "Capacitor": "C"}
    els = [(e, own[lowering._kind(e)]) for e in circ.elements if lowering._kind(e) in own]
    tape, outs, rport = probe_tape.record(circ, [e.__dict__[n] for e, n in els], device_limits=False)
    outs = outs + [rport]
    chan = next((i for i, (e, _) in enumerate(els) if e is chan_el), -1)
    p64 = np.array([float(e.__dict__[n]) for e, n in els], dtype=np.float64)
    rng = np.random.default_rng(5)
    r = pot_channel(B, T, 300.0, 80.0e3, 9).T.copy() if chan >= 0 else None          # [T,B]
    rt = wb.RowsTape(*tape.packed(), outs)
    assert rt.fits(len(els))
    params = torch.as_tensor(p64, device="cuda")
    rows = wb.ss_dyn_rows(rt, params, chan, None if r is None else cuda(r))
    n = len(outs)
    assert tuple(rows.shape) == (T, n, B)
    # the same tape in torch, float64, with autograd
    pv = [torch.tensor(v, dtype=torch.float64, device="cuda", requires_grad=(i != chan)) for i, v in enumerate(p64)]
    vals = [p if i != chan else torch.as_tensor(r.astype(np.float64), device="cuda") for i, p in enumerate(pv)]
    nodes = tape.evaluate_torch(vals, outs)
    ref = torch.stack([torch.broadcast_to(v, (T, B)) for v in nodes], dim=1)          # [T,n,B]
    scale = ref.detach().abs().amax(dim=(0, 2), keepdim=True).clamp_min(1e-300)
    assert float(((rows.double() - ref.detach()).abs() / scale).max()) < 2e-7
    grows = torch.as_tensor((rng.standard_normal((T, n, B)) / scale.cpu().numpy()).astype(np.float32), device="cuda")
    gp = wb.ss_dyn_rows_bwd(rt, params, chan, None if r is None else cuda(r), grows)
    (ref * grows.double()).sum().backward()
    for i, p in enumerate(pv):
        if i == chan:
            assert float(gp[i]) == 0.0
        else:
            # (float32 node adjoints under a random-sign row adjoint: the sum over T n B terms cancels to ~1e-3 of their size)
            assert abs(float(gp[i]) - float(p.grad)) <= 2e-4 * abs(float(p.grad)) + 1e-30, (i, float(gp[i]), float(p.grad))
    # refusals: a tape that names a later operation; a channel without its values
    bad = wb.RowsTape(np.array([[2, 1, 0], [0, 0, 0]]), [1.0], [0])
    with pytest.raises(wb.WdfHipError, match="not one the probe records"):
        wb.ss_dyn_rows(bad, params, -1, None)
    if chan >= 0:
        with pytest.raises(wb.WdfHipError, match="come together"):
            wb.ss_dyn_rows(rt, params, chan, None)


def test_rows_at_the_largest_tape_the_device_takes(wdf):
    """192 operations, 15 component values, 32 constants, 48 row entries: the reverse kernel's 155 KB of LDS (asked for per
    launch) -- against the tape's host evaluation with forward tangents (probe_tape.Tape.evaluate), sample by sample."""
    from wdf_hip import binding as wb, probe_tape as pt
    rng = np.random.default_rng(12)
    P, n_ops, n_out, B, T = 15, 192, 48, 3, 5
    ops = [[pt.OP_PARAM, p, 0] for p in range(P)] + [[pt.OP_CONST, j, 0] for j in range(32)]
    while len(ops) < n_ops:
        i = len(ops)
        a, b = int(rng.integers(0, i)), int(rng.integers(0, i))
        ops.append([[pt.OP_ADD, a, b], [pt.OP_SUB, a, b], [pt.OP_MUL, a, b], [pt.OP_NEG, a, 0], [pt.OP_DIV, a, int(rng.integers(0, P))],
                    [pt.OP_RECIP, int(rng.integers(0, P)), 0]][int(rng.integers(0, 6))])
    consts = rng.uniform(0.5, 1.5, 32)
    outs = [int(v) for v in rng.integers(P, n_ops, n_out)]
    tape = pt.Tape()
    tape.ops, tape.consts = [tuple(o) for o in ops], list(consts)
    p64 = rng.uniform(0.7, 1.3, P)
    chan = 4
    r = rng.uniform(0.7, 1.3, (T, B)).astype(np.float32)
    rt = wb.RowsTape(ops, consts, outs)
    assert rt.fits(P) and not wb.RowsTape(ops + [[pt.OP_NEG, 0, 0]], consts, outs).fits(P)
    params = torch.as_tensor(p64, device="cuda")
    rows = wb.ss_dyn_rows(rt, params, chan, cuda(r)).cpu().numpy().astype(np.float64)
    grows = rng.standard_normal((T, n_out, B)).astype(np.float32)
    gp = wb.ss_dyn_rows_bwd(rt, params, chan, cuda(r), cuda(grows)).cpu().numpy()
    ref_rows, ref_gp, mag = np.zeros((T, n_out, B)), np.zeros(P), np.zeros(P)
    for t in range(T):
        for b in range(B):
            pv = p64.copy()
            pv[chan] = float(r[t, b])
            val, jac = tape.evaluate(pv, outs)
            ref_rows[t, :, b] = val
            ref_gp += grows[t, :, b].astype(np.float64) @ jac
            mag += np.abs(grows[t, :, b].astype(np.float64)) @ np.abs(jac)
    ref_gp[chan] = 0.0
    finite = np.isfinite(ref_rows) & (np.abs(ref_rows) < 1e30)
    assert finite.mean() > 0.5
    assert np.max(np.abs(rows[finite] - ref_rows[finite]) / (np.abs(ref_rows[finite]) + 1e-30)) < 3e-7
    ok = np.isfinite(mag) & (mag < 1e30)
    assert ok.sum() >= P // 2 and gp[chan] == 0.0
    assert np.all(np.abs(gp[ok] - ref_gp[ok]) <= 1e-5 * mag[ok] + 1e-30), (gp, ref_gp, mag)
