"""CPU: the loop recorder turns a reference-style per-sample loop into the same state-space
matrices the Circuit probe derives -- checked with the kernel launch replaced by a capture
(no compute here; the GPU suite runs the real thing)."""
import ast
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

FS = 48000
REF = "/root/reference"


class Captured(Exception):
    pass


@pytest.fixture()
def capture(monkeypatch):
    from wdf_hip import lowering, trace
    got = {}

    class FakeFn:
        @staticmethod
        def apply(coef, rootp, x, z0, ns, ni, kind, n_up, n_down, want_zT):
            got.update(coef=coef, rootp=rootp, x=x, z0=z0, ns=ns, ni=ni, kind=kind)
            T, B = x.shape[1], x.shape[0]
            y = (coef.sum() * 0.0 + torch.zeros(T, B)).float()      # keeps the autograd graph alive
            return y, torch.zeros(ns, B)

    monkeypatch.setattr(trace, "_device", lambda: torch.device("cpu"))
    monkeypatch.setattr(lowering._StateSpaceFn, "apply", FakeFn.apply)
    return got


def test_recorded_ladder_loop_equals_probed_matrices(capture):
    """A hand-written loop over a two-capacitor ladder (tests/loops.py): what the recorder lowers it to
    is the state-space program the Circuit probe derives from the same tree."""
    import tf_wdf as wdf
    from loops import BridgedLadder
    m = BridgedLadder(wdf, FS)
    x = np.random.default_rng(0).standard_normal((3, 40))
    out = m.run(x)
    assert tuple(out.shape) == (40, 3, 1)                          # TensorArray.stack() layout
    ref, _ = wdf.Circuit(m.top, m.src, m.Cb).matrices()
    assert capture["ns"] == 2 and capture["ni"] == 1 and capture["kind"] == 0
    assert torch.allclose(capture["coef"].double(), ref, rtol=1e-6, atol=1e-9)
    assert torch.allclose(capture["x"][:, :, 0].double(), torch.as_tensor(x), atol=1e-6)
    # gradients reach the components through the recorded coefficients
    g = torch.autograd.grad(capture["coef"][0], m.params, allow_unused=True)
    assert sum(1 for v in g if v is not None and float(v.abs()) > 0) >= 2
    # the capacitors now hold numeric final states (carried into the next run unless reset() is called)
    assert isinstance(m.Cb.z, torch.Tensor) and tuple(m.Cb.z.shape) == (3, 1)
    m.run(x)                                                       # second run records again
    assert capture["z0"] is None or tuple(capture["z0"].shape) == (2, 3)


def test_recorder_rejects_nonuniform_loops(capture):
    import tf_wdf as wdf
    from loops import BridgedLadder
    from wdf_hip import trace
    tf = wdf.tf
    m = BridgedLadder(wdf, FS)
    x = tf.cast(tf.expand_dims(np.ones((2, 4)), axis=-1), dtype=tf.float32)
    ta = tf.TensorArray(dtype=tf.float32, size=4)
    m.top.calc_impedance()
    with pytest.raises(trace.WdfTraceError):
        for n in range(4):
            m.src.set_voltage(x[:, n])
            m.src.incident(m.top.reflected())
            down = m.src.reflected()
            m.top.incident(down * (2.0 if n == 2 else 1.0))        # step 2 is a different map
            ta = ta.write(n, wdf.voltage(m.Cb))
        ta.stack()
    trace._current = None
    # a product of two waves is not an adaptor operation
    m2 = BridgedLadder(wdf, FS)
    m2.top.calc_impedance()
    m2.src.set_voltage(x[:, 0])
    w = m2.top.reflected()
    with pytest.raises(trace.WdfTraceError):
        _ = (w + 0.0) * (w + 0.0)
    trace._current = None


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("script,probe", [("wdf_py/simple_circuits/lpf.py", "C1"),
                                          ("wdf_py/simple_circuits/voltage_divider.py", "R1")])
def test_reference_model_classes_run_unchanged(capture, script, probe):
    """The reference's own Model class (AST-extracted from the script, nothing edited) records
    and lowers through the drop-in tf_wdf."""
    import tf_wdf as wdf
    tree = ast.parse(open(os.path.join(REF, script)).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Model"]
    ns = {"tf": wdf.tf, "wdf": wdf, "FS": FS}
    exec(compile(ast.Module(cls, []), script, "exec"), ns)
    m = ns["Model"]()
    x = np.random.default_rng(1).standard_normal((1, 64))
    out = m.forward(x)
    assert tuple(out.shape) == (64, 1, 1)
    ref, _ = wdf.Circuit(m.I1, m.Vs, getattr(m, probe)).matrices()
    assert torch.allclose(capture["coef"].double(), ref, rtol=1e-6, atol=1e-9)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_clipper_model_runs_unchanged(monkeypatch):
    """clipper_pot.py's own ClipperModel (AST-extracted, unedited) with a committed reference
    weight file: the recorder recognises the per-sample-R clipper loop and hands it to the
    MLP-root kernel (captured here)."""
    import json
    import tf_wdf as wdf
    from layers import DenseRootModel
    from wdf_hip import mlp_root, trace
    got = {}

    class FakeFn:
        @staticmethod
        def apply(theta2, w, x, r, z0, fs, hidden, n_tanh, want_zT):
            got.update(theta2=theta2, w=w, x=x, r=r, hidden=hidden, n_tanh=n_tanh, fs=fs)
            B, T = x.shape
            return torch.zeros(T, B) + 0.0 * w.sum(), torch.zeros(B)

    monkeypatch.setattr(trace, "_device", lambda: torch.device("cpu"))
    monkeypatch.setattr(mlp_root._ClipperMlpFn, "apply", FakeFn.apply)
    script = "wdf_py/diode_clipper/clipper_pot.py"
    tree = ast.parse(open(os.path.join(REF, script)).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ClipperModel"]
    ns = {"tf": wdf.tf, "wdf": wdf, "FS": FS, "C_val": 4.7e-9, "DenseRootModel": DenseRootModel}
    exec(compile(ast.Module(cls, []), script, "exec"), ns)
    mj = json.load(open(os.path.join(REF, "wdf_py/diode_clipper/models/1N4148 (1U-1D)_2x16_training_2000.json")))
    model = ns["ClipperModel"](mj)
    B, T = 3, 20
    rng = np.random.default_rng(2)
    data = np.stack([rng.standard_normal((B, T)), np.full((B, T), 25.2e3)], axis=-1)     # [B,T,2]
    out = model.forward(data)
    assert tuple(out.shape) == (T, B, 1, 1)                              # clipper_pot.py:126
    assert got["hidden"] == 16 and got["n_tanh"] == 3 and got["w"].numel() == 609
    assert torch.allclose(got["x"].double(), torch.as_tensor(data[:, :, 0]), atol=1e-6)
    assert torch.allclose(got["r"].double(), torch.as_tensor(data[:, :, 1]), rtol=1e-6)
    assert len(model.trainable_variables) == 8
    g = torch.autograd.grad(out.sum(), model.trainable_variables[0], allow_unused=True)
    assert g[0] is not None


def test_replay_memo_stops_growing_after_the_second_step(capture, monkeypatch):
    """Replay: from the second step on every wave operation of a uniform loop is answered from the recorder's
    memo (same coefficient objects, no torch arithmetic), the memo therefore stops growing, and a step is
    confirmed by comparing object identities instead of coefficient values -- for a static tree and for the
    pot clipper, whose calc_impedance() runs every step (scalar component arithmetic memoised per variable
    version in compat_tf)."""
    import json
    import tf_wdf as wdf
    from loops import BridgedLadder, PotClipper
    from wdf_hip import mlp_root, trace
    sizes, by_value = [], []
    orig_close, orig_sig = trace.Recorder._close_step, trace.Recorder._sig

    def close(self):
        sizes.append(len(self.memo))
        return orig_close(self)

    monkeypatch.setattr(trace.Recorder, "_close_step", close)
    monkeypatch.setattr(trace.Recorder, "_sig", staticmethod(lambda waves: (by_value.append(1), orig_sig(waves))[1]))
    monkeypatch.setattr(mlp_root, "clipper_mlp",
                        lambda theta2, w, x, r, z0, fs, hidden, n_tanh, C, R_static=None, time_parallel="auto":
                        (torch.zeros(x.shape[1], x.shape[0]) + 0.0 * w.sum(), torch.zeros(x.shape[0])))
    T = 40
    BridgedLadder(wdf, FS).run(np.random.default_rng(0).standard_normal((2, T)))
    assert len(sizes) == T and len(set(sizes[1:])) == 1, sizes            # constant from the second step on
    assert len(by_value) <= 3                                             # steps 0, 1 (and the final check) by value, the rest by identity
    sizes.clear(); by_value.clear()
    js = {"in_shape": [None, 2], "layers": [
        {"type": "dense", "activation": "tanh", "shape": [None, 4], "weights": [np.ones((2, 4)).tolist(), [0.0] * 4]},
        {"type": "dense", "activation": "tanh", "shape": [None, 4], "weights": [np.eye(4).tolist(), [0.0] * 4]},
        {"type": "dense", "activation": "tanh", "shape": [None, 4], "weights": [np.eye(4).tolist(), [0.0] * 4]},
        {"type": "dense", "activation": "", "shape": [None, 1], "weights": [np.ones((4, 1)).tolist(), [0.0]]}]}
    data = np.stack([np.random.default_rng(1).standard_normal((3, T)), np.full((3, T), 25.2e3)], axis=-1)
    PotClipper(wdf, FS, 4.7e-9, mlp_json=js).run(data)
    assert len(sizes) == T and len(set(sizes[2:])) == 1, sizes
    assert len(by_value) <= 4


def test_scalar_memo_survives_in_place_mutation_of_a_result():
    """Scalar component arithmetic is memoised while a loop is being recorded (compat_tf.Tensor._scalar_memo).  A script
    that modifies a returned value in place must not poison later, identical operations: the entry is recomputed once
    its result's version has moved."""
    import types
    import torch
    from wdf_hip import compat_tf as ctf, trace
    rec = types.SimpleNamespace(scalar_memo={})
    saved, trace._current = trace._current, rec
    try:
        r = ctf.constant(4.0)
        a = r * 2.0
        assert float(a) == 8.0 and (r * 2.0) is a                  # memoised: the same object comes back
        a += 1.0                                                   # the script mutates ITS value ...
        b = r * 2.0
        assert float(b) == 8.0 and b is not a and float(a) == 9.0  # ... and the operation still means r * 2
        assert (r * 2.0) is b
    finally:
        trace._current = saved


def test_clamp_bounds_recognises_the_elements_constraints():
    """compat_tf.clamp_bounds: the constraints tf_wdf gives its Variables (tf_wdf.py:74,104) are clamps the resident
    optimizer can fold into its kernel; anything else is left to the per-Variable path."""
    import tf_wdf
    from wdf_hip import compat_tf as tf
    res = tf_wdf.Resistor(1000.0, trainable=True)
    cap = tf_wdf.Capacitor(1.0e-8, FS, trainable=True)
    assert tf.clamp_bounds(res.R.constraint) == (180.0, 1.0e6)
    assert tf.clamp_bounds(tf_wdf.ResistiveVoltageSource(1000.0, trainable=True).R.constraint) == (-float("inf"), float("inf"))
    lo, hi = tf.clamp_bounds(cap.C.constraint)
    assert lo == float(np.float32(0.1e-12)) and hi == 1.0
    assert tf.clamp_bounds(None) == (-float("inf"), float("inf"))
    assert tf.clamp_bounds(lambda z: z * 2.0) is None
    assert tf.clamp_bounds(lambda z: tf.abs(z)) is None
    assert tf.clamp_bounds(lambda z: tf.clip_by_value(z, 0.0, float("inf"))) == (0.0, float("inf"))


def test_adam_without_resident_variables_takes_the_per_variable_path():
    """Host Variables never reach the fused update (no GPU here): the optimizer's rule is the Keras one."""
    from wdf_hip import compat_tf as tf
    v = tf.Variable(1.0, constraint=lambda z: tf.clip_by_value(z, 0.0, 0.9995))
    opt = tf.keras.optimizers.Adam(learning_rate=1.0e-3)
    opt.apply_gradients([(tf.constant(2.0), v)])
    assert opt.iterations == 1 and not opt._resident
    assert abs(float(v) - 0.999) < 1e-6          # first Adam step moves by lr whatever the gradient's size
    opt.apply_gradients([(tf.constant(-2.0), v)])
    assert float(v) <= 0.9995


def test_warm_up_controller_policy():
    """warmstart.WarmUpController (host side of warm-started chunks, no device here: verdicts are fed by hand): repairs
    lengthen the warm-up by two units and bar what failed for 256 calls; clean calls shorten it by one unit after
    `wait_calls`; far inside the tolerance the bolder rule takes two units after four calls; a change only happens once
    the caller holds start states for the new value."""
    from wdf_hip import warmstart

    class Done:
        def query(self):
            return True

    def verdict(c, w, gated_total, miss=2.0e-6, n_bad=0):
        c.pin = torch.zeros(5, dtype=torch.int32)
        c.pin[0], c.pin[2], c.pin[4] = n_bad, 0, gated_total
        c.pin[1:2] = torch.tensor([miss], dtype=torch.float32).view(torch.int32)
        c.pending = (Done(), c.calls, w)

    c = warmstart.WarmUpController(cold=672, start=150, unit=16, floor=32, miss_waves=8, wait_calls=32)
    assert c.W == 160 and c.candidates() == [144, 160, 192]
    assert c.begin({}) is None                                   # nothing to start from: cold call
    rows = {w: None for w in c.candidates()}
    assert c.begin(rows) == 160
    verdict(c, 160, gated_total=3)                               # a wave or two: the fp32 floor, not a short warm-up
    assert c.begin(rows) == 160 and c.want is None
    verdict(c, 160, gated_total=40)                              # dozens of waves: too short
    assert c.begin(rows) == 192 and c.bad == 160
    rows = {w: None for w in c.candidates()}
    for _ in range(40):
        c.begin(rows)
    verdict(c, 192, gated_total=40)                              # clean long enough: one unit less -- but 176 > bad = 160 only
    assert c.begin(rows) == 176
    rows = {w: None for w in c.candidates()}
    for _ in range(40):
        c.begin(rows)
    verdict(c, 176, gated_total=40)
    assert c.begin(rows) == 176 and c.want is None               # 160 failed within the last 256 calls: not tried again
    for _ in range(260):
        c.begin(rows)
    verdict(c, 176, gated_total=40)
    assert c.begin(rows) == 160                                  # ... after 256 calls it is
    # the bolder rule (state-space forward): far inside the tolerance -> two units after four calls, down to the floor
    b = warmstart.WarmUpController(cold=664, start=166, unit=16, floor=16, miss_waves=1, wait_calls=16, bold_below=4.0, tol=1.0e-6)
    assert b.W == 176 and b.candidates() == [144, 160, 176, 208]
    rows = {w: None for w in b.candidates()}
    for _ in range(5):
        b.begin(rows)
    verdict(b, 176, gated_total=0, miss=1.0e-7)
    assert b.begin(rows) == 144
    rows = {w: None for w in b.candidates()}
    verdict(b, 144, gated_total=1, miss=3.0e-6, n_bad=2)         # one repaired wave counts here
    assert b.begin(rows) == 176 and b.bad == 144
    # a wanted value the caller has no states for yet is not taken
    b.want = 96
    assert b.begin({176: None}) == 176 and b.want == 96
