"""CPU: pin the oracle (oracle/wdf_oracle.c) against the reference-derived goldens.

Goldens come from tests/golden/gen_golden.py, which executed the reference's own files
(see its docstring).  Tolerances are stated per test.
"""
import numpy as np
import pytest


def rel_or_abs(a, b, floor):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))


# ---- g5: Wright omega (toms917.cpp:134-375) ------------------------------------------
def test_omega_vs_reference_toms917_and_mpmath(oracle, golden):
    g = golden("g5_omega.npz")
    x = g["x"]
    w = oracle.wright_omega(x)
    # vs the reference's own toms917 build: complex vs real arithmetic ordering only
    assert rel_or_abs(w, g["w_toms917"], 1e-300) < 4e-16
    assert np.mean(w == g["w_toms917"]) > 0.99
    assert rel_or_abs(w, g["w_scipy"], 1e-300) < 1e-14
    assert rel_or_abs(w, g["w_mpmath"], 1e-300) < 1e-14


def test_omega_live_reference_build(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no reference checkout)")
    x = np.random.default_rng(0).uniform(-110, 110, 20000)
    assert rel_or_abs(oracle.wright_omega(x), oracle.ref_wright_omega(x), 1e-300) < 4e-16


def test_omega_f32_instantiation(oracle, golden):
    g = golden("g5_omega.npz")
    x = g["x"]
    w32 = oracle.wright_omega(x, np.float32).astype(np.float64)
    x32 = x.astype(np.float32).astype(np.float64)
    w_of_x32 = oracle.wright_omega(x32)
    # fp32 result vs exact omega at the fp32-rounded argument: a few ulp, absolute floor
    # below the fp32 denormal range (exp underflow, x < -87).  The worst band is -20 < x < -2,
    # where the toms917 FSC step evaluates r = x - w - log(w) with |log w| ~ |x|: its fp32
    # rounding (ulp(|x|)/2) lands directly in w's relative error (measured 9e-7).
    err = np.abs(w32 - w_of_x32) / np.maximum(np.abs(w_of_x32), 1e-37)
    assert np.max(err) < 2e-6
    _, it = oracle.wright_omega_iters(x, np.float32)
    assert set(np.unique(it)) <= {0, 1, 2}
    assert np.mean(it == 2) < 0.05      # the second FSC iteration is the rare path in fp32


def test_omega_special_values(oracle):
    w = oracle.wright_omega(np.array([-np.inf, np.inf, np.nan, 0.0, 1.0]))
    assert w[0] == 0.0 and np.isinf(w[1]) and np.isnan(w[2])
    assert abs(w[3] - 0.5671432904097838) < 1e-15      # omega(0) = W(1)
    assert abs(w[4] - 1.0) < 1e-15                     # omega(1) = 1


# ---- g4: diode pair (diode_pretraining.py:39-60) --------------------------------------
def test_diode_pair_vs_reference_function(oracle, golden):
    g = golden("g4_diode_pair.npz")
    a = g["a"]
    for i in range(len(g["Is"])):
        for j, R in enumerate(g["R"]):
            b = oracle.diode_pair(a, float(R), float(g["Is"][i]), float(g["Vt"][i]), float(g["nabla"][i]),
                                  int(g["n_up"][i]), int(g["n_down"][i]))
            # the reference function returns np.float32: half-ulp of |b| <= 5.3
            assert np.max(np.abs(b - g["b"][i, j])) <= 2.4e-7 + 1e-12


def test_diode_pair_cpp_form(oracle):
    """Toms917DiodePair.h:51-59 (eqn 39, float state, omega through the reference toms917)
    equals the N_up = N_down = 1 case of eqn 45."""
    L = oracle.ref_lib()
    if L is None:
        pytest.skip("oracle/_ref not built")
    a = np.linspace(-5, 5, 201).astype(np.float32)
    bc = np.array([L.ref_toms917_diode_pair(ai, 2112.0, 4.352e-9, 25.85e-3, 1.906) for ai in a])
    bo = oracle.diode_pair(a.astype(np.float64), 2112.0, 4.352e-9, 25.85e-3, 1.906)
    assert np.max(np.abs(bc - bo)) < 3e-6      # float32 evaluation of a +- 2Vt*omega ~ 9 V


# ---- g1 / g2: linear trees (tf_wdf.py, lpf.py, voltage_divider.py) --------------------
def test_rc_lowpass_forward_and_grads(oracle, golden):
    g = golden("g1_rc_lowpass.npz")
    circ = oracle.rc_lowpass_circuit(48000.0)
    theta = np.array([float(g["R"]), float(g["C"])])
    x = g["x"][None, :]
    y = oracle.tree_fwd(circ, theta, x)[:, 0]
    assert np.max(np.abs(y - g["y_f64"])) < 1e-13
    assert np.max(np.abs(oracle.tree_fwd(circ, theta, x, np.float32)[:, 0] - g["y_f32"])) < 2e-6
    # MSE gradient (lpf.py:87-90): dL/dy = 2 (y - t)/N
    gy = (2.0 * (y - g["target"]) / y.size)[:, None]
    gr = oracle.tree_grad(circ, theta, x, gy)
    assert abs(gr[0] - float(g["dR_f64"])) < 1e-9 * abs(float(g["dR_f64"]))
    assert abs(gr[1] - float(g["dC_f64"])) < 1e-9 * abs(float(g["dC_f64"]))
    assert abs(np.mean((y - g["target"]) ** 2) - float(g["loss_f64"])) < 1e-14


def test_rc_lowpass_state_carries_across_calls(oracle, golden):
    """lpf.py never resets C1: forward() #2 starts from the final state of forward() #1."""
    g = golden("g1_rc_lowpass.npz")
    circ = oracle.rc_lowpass_circuit(48000.0)
    theta = np.array([float(g["R"]), float(g["C"])])
    x = g["x"][None, :]
    _, zT = oracle.tree_fwd(circ, theta, x, return_state=True)
    assert abs(zT[0, 1] - g["z_after_f64"][0]) < 1e-13
    y2, zT2 = oracle.tree_fwd(circ, theta, x, z0=zT, return_state=True)
    y2 = y2[:, 0]
    assert np.max(np.abs(y2 - g["y_second_call_f64"])) < 1e-13
    assert abs(zT2[0, 1] - g["z_after_second_call_f64"][0]) < 1e-13
    # the second epoch's gradient (round 6): the state handed over is a CONSTANT of the new tape -- central differences of the
    # oracle with z0 held fixed reproduce what the reference's second tape.gradient gives
    assert abs(np.mean((y2 - g["target"]) ** 2) - float(g["loss_second_call_f64"])) < 1e-14

    def loss_at(th):
        return np.mean((oracle.tree_fwd(circ, th, x, z0=zT)[:, 0] - g["target"]) ** 2)

    for k, key in ((0, "dR_second_call_f64"), (1, "dC_second_call_f64")):
        h = 1e-5 * theta[k]
        tp, tm = theta.copy(), theta.copy()
        tp[k] += h
        tm[k] -= h
        fd = (loss_at(tp) - loss_at(tm)) / (2 * h)
        assert abs(fd - float(g[key])) < 1e-7 * abs(float(g[key])), (key, fd, float(g[key]))
    assert abs(float(g["dC_second_call_f64"]) - float(g["dC_f64"])) > 3e-4 * abs(float(g["dC_f64"]))   # (the two epochs differ)


def test_rc_lowpass_is_bilinear_one_pole(oracle, golden):
    """Known answer: the WDF RC divider equals the bilinear-transformed 1-pole lowpass with
    fc = 1/(2 pi R C) pre-warped -- independent of any golden."""
    fs, R, Cc = 48000.0, 1000.0, 1.0e-6
    x = golden("g1_rc_lowpass.npz")["x"]
    y = oracle.tree_fwd(oracle.rc_lowpass_circuit(fs), np.array([R, Cc]), x[None, :])[:, 0]
    k = 2.0 * fs * R * Cc                      # s -> 2 fs (1-z^-1)/(1+z^-1)
    b0, a1 = 1.0 / (1.0 + k), (1.0 - k) / (1.0 + k)
    ref = np.zeros_like(x)
    xm1 = ym1 = 0.0
    for i, xi in enumerate(x):
        ref[i] = b0 * (xi + xm1) - a1 * ym1
        xm1, ym1 = xi, ref[i]
    # the WDF output is the trapezoidal capacitor voltage: same filter, half-sample aligned
    assert np.max(np.abs(y - ref)) < 1e-12


def test_voltage_divider(oracle, golden):
    g = golden("g2_voltage_divider.npz")
    circ = oracle.voltage_divider_circuit()
    theta = np.array([float(g["R1"]), float(g["R2"])])
    x = g["x"][None, :]
    y = oracle.tree_fwd(circ, theta, x)[:, 0]
    assert np.max(np.abs(y - g["y_f64"])) < 1e-14
    assert np.max(np.abs(y - x[0] * 2000.0 / 2100.0)) < 1e-14       # analytic: R1/(R1+R2)
    gy = (2.0 * (y - g["target"]) / y.size)[:, None]
    gr = oracle.tree_grad(circ, theta, x, gy)
    assert abs(gr[0] - float(g["dR1_f64"])) < 1e-10 * abs(float(g["dR1_f64"]))
    assert abs(gr[1] - float(g["dR2_f64"])) < 1e-10 * abs(float(g["dR2_f64"]))


# ---- g3: MLP-root clipper (clipper_pot.py:94-127,141-177, layers.py) -------------------
@pytest.mark.parametrize("name", ["2x4", "2x8", "2x16", "2x16_pre"])
def test_mlp_clipper_forward_loss_grads(oracle, golden, name):
    g = golden("g3_mlp_clipper.npz")
    sizes, acts = [int(s) for s in g[f"{name}_sizes"]], [int(a) for a in g[f"{name}_acts"]]
    circ = oracle.clipper_mlp_circuit(48000.0, sizes, acts)
    theta = np.concatenate([[45.0e3, float(g["C"])], g[f"{name}_theta"]])
    x = g["x"]                                   # [B,T,2]
    y = oracle.tree_fwd(circ, theta, x)          # [T,B]
    assert np.max(np.abs(y - g[f"{name}_y_f64"])) < 1e-12
    assert np.max(np.abs(oracle.tree_fwd(circ, theta, x, np.float32) - g[f"{name}_y_f32"])) < 2e-5
    # loss = MSE + ESR on [:, skip:, :] with (outs, target) passed as (target, pred)
    skip = int(g["skip"])
    outs = y.T[:, skip:, None]                   # [B,T-skip,1]
    tgt = g["target"][:, skip:, :]
    mse = oracle.mse_loss(outs, tgt)
    esr = oracle.esr_loss(outs, tgt)             # energy normaliser = sum(outs^2): the swap
    assert abs(mse - float(g[f"{name}_mse_f64"])) < 1e-13
    assert abs(esr - float(g[f"{name}_esr_f64"])) < 1e-13
    assert abs(mse + esr - float(g[f"{name}_loss_f64"])) < 1e-13
    # gradient of that loss w.r.t. y, then complex-step through the oracle
    n = outs.size
    d = outs - tgt
    S, E = np.sum(d * d), np.sum(outs * outs) + np.finfo(float).eps
    g_outs = 2.0 * d / n + (1.0 / (2.0 * esr)) * (2.0 * d / E - S * 2.0 * outs / (E * E)) / n
    gy = np.zeros_like(y)
    gy[skip:, :] = g_outs[:, :, 0].T
    ks = list(range(2, 2 + g[f"{name}_theta"].size))
    if name != "2x4":
        ks = ks[:: max(1, len(ks) // 40)]        # subsample the big nets to keep the CPU suite short
    gr = oracle.tree_grad(circ, theta, x, gy, params=ks)
    ref = g[f"{name}_grad_f64"][[k - 2 for k in ks]]
    assert np.max(np.abs(gr - ref)) < 1e-9 * max(1.0, np.max(np.abs(ref)))


def test_mlp_clipper_relu_network(oracle, golden):
    """g9: the reference's ClipperModel on a ReLU network (layers.py:63-67): the oracle's ACT_RELU against it -- outputs, the
    MSE + ESR loss and a spread of weight gradients (complex step does not see a kink: every unit keeps its side in fp64)."""
    g = golden("g9_mlp_relu.npz")
    name = "2x16_relu"
    sizes, acts = [int(s) for s in g[f"{name}_sizes"]], [int(a) for a in g[f"{name}_acts"]]
    assert acts[:-1] == [oracle.ACT_RELU] * (len(acts) - 1)
    circ = oracle.clipper_mlp_circuit(48000.0, sizes, acts)
    theta = np.concatenate([[45.0e3, float(g["C"])], g[f"{name}_theta"]])
    x = g["x"]
    y = oracle.tree_fwd(circ, theta, x)
    assert np.max(np.abs(y - g[f"{name}_y_f64"])) < 1e-12
    skip = int(g["skip"])
    outs, tgt = y.T[:, skip:, None], g["target"][:, skip:, :]
    mse, esr = oracle.mse_loss(outs, tgt), oracle.esr_loss(outs, tgt)
    assert abs(mse + esr - float(g[f"{name}_loss_f64"])) < 1e-13
    n, d = outs.size, outs - tgt
    S, E = np.sum(d * d), np.sum(outs * outs) + np.finfo(float).eps
    g_outs = 2.0 * d / n + (1.0 / (2.0 * esr)) * (2.0 * d / E - S * 2.0 * outs / (E * E)) / n
    gy = np.zeros_like(y)
    gy[skip:, :] = g_outs[:, :, 0].T
    ks = list(range(2, 2 + g[f"{name}_theta"].size))
    ks = ks[:: max(1, len(ks) // 40)]
    gr = oracle.tree_grad(circ, theta, x, gy, params=ks)
    ref = g[f"{name}_grad_f64"][[k - 2 for k in ks]]
    assert np.max(np.abs(gr - ref)) < 1e-9 * max(1.0, np.max(np.abs(ref)))


# ---- g6: diode-pair clipper ------------------------------------------------------------
@pytest.mark.parametrize("cfg,n_up,n_down", [("1u1d", 1, 1), ("2u3d", 2, 3)])
def test_diode_clipper_forward_vs_reference_pieces(oracle, golden, cfg, n_up, n_down):
    g = golden("g6_diode_clipper.npz")
    circ = oracle.clipper_diode_circuit(48000.0, n_up, n_down)
    y = oracle.tree_fwd(circ, g["theta"], g["x"])
    # (a) tf_wdf.py elements (f32) + the reference diode_pair_func (returns float32)
    assert np.max(np.abs(y - g[f"y_refpieces_{cfg}_f32"])) < 5e-7
    # (b) tf_wdf.py elements in f64 + autograd diode root
    assert np.max(np.abs(y - g[f"y_{cfg}_f64"])) < 1e-12
    # specialised twin == generic interpreter
    y2 = oracle.clipper_fwd(g["theta"], 48000.0, g["x"], n_up=n_up, n_down=n_down)
    assert np.max(np.abs(y - y2)) < 1e-13


@pytest.mark.parametrize("cfg,n_up,n_down", [("1u1d", 1, 1), ("2u3d", 2, 3)])
def test_diode_clipper_grads(oracle, golden, cfg, n_up, n_down):
    g = golden("g6_diode_clipper.npz")
    theta, x = g["theta"], g["x"]
    circ = oracle.clipper_diode_circuit(48000.0, n_up, n_down)
    y = oracle.tree_fwd(circ, theta, x)
    gy = 2.0 * (y - g["target"]) / y.size
    ref = g[f"grad_{cfg}_f64"]                    # torch autograd through tf_wdf.py
    cs = oracle.tree_grad(circ, theta, x, gy)     # complex step through the oracle
    assert np.max(np.abs(cs - ref) / np.abs(ref)) < 1e-8
    _, adj = oracle.clipper_fwd_bwd(theta, 48000.0, x, gy, n_up=n_up, n_down=n_down)   # hand adjoint
    assert np.max(np.abs(adj - ref) / np.abs(ref)) < 1e-8
    assert abs(np.mean((y - g["target"]) ** 2) - float(g[f"loss_{cfg}_f64"])) < 1e-14


def test_diode_clipper_per_sample_resistance(oracle, golden):
    g = golden("g6_diode_clipper.npz")
    theta, x, r = g["theta"], g["x"], g["r"]
    circ = oracle.clipper_diode_circuit(48000.0, per_sample_r=True)
    xin = np.stack([x, r], axis=-1)
    y = oracle.tree_fwd(circ, theta, xin)
    assert np.max(np.abs(y - g["y_1u1d_rpot_f64"])) < 1e-12
    gy = 2.0 * (y - g["target"]) / y.size
    y2, adj = oracle.clipper_fwd_bwd(theta, 48000.0, x, gy, r=r)
    assert np.max(np.abs(y - y2)) < 1e-13
    ref = g["grad_1u1d_rpot_f64"]                 # [dIs, dnVt, dC]
    got = adj[[0, 1, 3]]
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-8
    assert adj[2] == 0.0


def test_diode_clipper_f32_twin_and_fd(oracle, golden):
    """fp32 instantiation stays within rounding of fp64; central differences agree with the
    adjoint (pins dL/dIs, dL/dnVt, which no reference artefact pins)."""
    g = golden("g6_diode_clipper.npz")
    theta, x = g["theta"], g["x"]
    y64 = oracle.clipper_fwd(theta, 48000.0, x)
    y32 = oracle.clipper_fwd(theta, 48000.0, x, dtype=np.float32)
    assert np.max(np.abs(y32 - y64)) < 5e-6
    gy = 2.0 * (y64 - g["target"]) / y64.size
    _, adj = oracle.clipper_fwd_bwd(theta, 48000.0, x, gy)
    for k in range(4):
        h = 1e-6 * theta[k]
        tp, tm = theta.copy(), theta.copy()
        tp[k] += h
        tm[k] -= h
        fd = np.sum(gy * (oracle.clipper_fwd(tp, 48000.0, x) - oracle.clipper_fwd(tm, 48000.0, x))) / (2 * h)
        assert abs(fd - adj[k]) < 2e-6 * abs(adj[k]) + 1e-12
    loss, g4, _ = oracle.clipper_mse_step(theta, 48000.0, x, g["target"], dtype=np.float64)
    assert abs(loss - float(g["loss_1u1d_f64"])) < 1e-14
    assert np.max(np.abs(g4 - adj) / np.abs(adj)) < 1e-12


# ---- two different diodes (config C5): the oracle's own root, against mpmath -------------------
def test_asym_root_vs_mpmath(oracle):
    import mpmath as mp
    mp.mp.dps = 40
    Rp, Is1, V1, Is2, V2 = 2112.0, 4.352e-9, 0.0492701, 2.0e-6, 0.03619
    for a in (-5.0, -1.2, -0.3, -1e-3, 0.0, 1e-4, 0.25, 0.9, 4.0):
        f = lambda v: v + Rp * (Is1 * (mp.e ** (v / V1) - 1) - Is2 * (mp.e ** (-v / V2) - 1)) - a   # noqa: E731
        lo, hi = (mp.mpf(a), mp.mpf(0)) if a < 0 else (mp.mpf(0), mp.mpf(a))
        v = mp.findroot(f, (lo, hi), solver="illinois", tol=1e-60, maxsteps=400) if a != 0.0 else mp.mpf(0)
        b_ref = float(2 * v - a)
        b = oracle.asym_root(a, Rp, Is1, V1, Is2, V2)[0]
        assert abs(b - b_ref) < 1e-12 * max(1.0, abs(b_ref)), (a, b, b_ref)
    # equal diodes: the exact pair vs the reference's Wright-omega closed form (eqn 39/45),
    # which neglects the reverse diode's saturation current: ~2 Rp Is of difference
    a = np.linspace(-5, 5, 41)
    exact = oracle.asym_root(a, Rp, Is1, V1, Is1, V1)
    closed = oracle.diode_pair(a, Rp, Is1, V1, 1.0)
    assert np.max(np.abs(exact - closed)) < 4 * Rp * Is1
