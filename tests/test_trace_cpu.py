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


def lpf_model(wdf):
    tf = wdf.tf

    class Model(tf.Module):                       # the shape of lpf.py:20-49
        def __init__(self):
            super().__init__()
            self.Vs = wdf.IdealVoltageSource()
            self.R1 = wdf.Resistor(1000, True)
            self.C1 = wdf.Capacitor(1.0e-6, FS, True)
            self.S1 = wdf.Series(self.R1, self.C1)
            self.I1 = wdf.Inverter(self.S1)

        def forward(self, input):  # noqa: A002
            sequence_length = input.shape[1]
            input = tf.cast(tf.expand_dims(input, axis=-1), dtype=tf.float32)  # noqa: A001
            output_sequence = tf.TensorArray(dtype=tf.float32, size=sequence_length, clear_after_read=False)
            self.I1.calc_impedance()
            for i in range(sequence_length):
                self.Vs.set_voltage(input[:, i])
                self.Vs.incident(self.I1.reflected())
                self.I1.incident(self.Vs.reflected())
                output = wdf.voltage(self.C1)
                output_sequence = output_sequence.write(i, output)
            return output_sequence.stack()

    return Model()


def test_recorded_lpf_loop_equals_probed_matrices(capture):
    import tf_wdf as wdf
    m = lpf_model(wdf)
    x = np.random.default_rng(0).standard_normal((3, 40))
    out = m.forward(x)
    assert tuple(out.shape) == (40, 3, 1)                          # TensorArray.stack() layout, lpf.py:48
    ref, _ = wdf.Circuit(m.I1, m.Vs, m.C1).matrices()
    assert capture["ns"] == 1 and capture["ni"] == 1 and capture["kind"] == 0
    assert torch.allclose(capture["coef"].double(), ref, rtol=1e-6, atol=1e-9)
    assert torch.allclose(capture["x"][:, :, 0].double(), torch.as_tensor(x), atol=1e-6)
    # gradients reach R and C through the recorded coefficients
    g = torch.autograd.grad(capture["coef"][0], [m.R1.R, m.C1.C])
    assert all(float(v.abs()) > 0 for v in g)
    # the capacitor now holds a numeric final state (carried into the next forward, lpf.py quirk)
    assert isinstance(m.C1.z, torch.Tensor) and tuple(m.C1.z.shape) == (3, 1)
    m.forward(x)                                                   # second call records again
    assert capture["z0"] is None or tuple(capture["z0"].shape) == (1, 3)


def test_recorder_rejects_nonuniform_loops(capture):
    import tf_wdf as wdf
    from wdf_hip import trace
    tf = wdf.tf
    m = lpf_model(wdf)
    x = tf.cast(tf.expand_dims(np.ones((2, 4)), axis=-1), dtype=tf.float32)
    ta = tf.TensorArray(dtype=tf.float32, size=4)
    m.I1.calc_impedance()
    with pytest.raises(trace.WdfTraceError):
        for i in range(4):
            m.Vs.set_voltage(x[:, i])
            m.Vs.incident(m.I1.reflected())
            up = m.Vs.reflected()
            m.I1.incident(up * (2.0 if i == 2 else 1.0))           # step 2 is a different map
            ta = ta.write(i, wdf.voltage(m.C1))
        ta.stack()
    trace._current = None
    # a product of two waves is not an adaptor operation
    m2 = lpf_model(wdf)
    m2.I1.calc_impedance()
    m2.Vs.set_voltage(x[:, 0])
    w = m2.I1.reflected()
    with pytest.raises(trace.WdfTraceError):
        _ = m2.Vs.reflected() * m2.Vs.reflected() if False else (w + 0.0) * (w + 0.0)
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
