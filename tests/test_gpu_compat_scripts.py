"""GPU: the compat tier.  Model classes written exactly like the reference scripts' (their own
`for i in range(sequence_length)` loop, TensorArray, GradientTape, Adam) run against the
drop-in tf_wdf / layers and are lowered to the HIP kernels by the loop recorder.  Checked
against the goldens recorded from the reference's code."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000


def make_lpf_model(wdf, tf):
    class Model(tf.Module):                                   # lpf.py:20-49
        def __init__(self):
            super(Model, self).__init__()
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
            output_sequence = output_sequence.stack()
            return output_sequence

    return Model()


def test_lpf_script_loop(golden):
    import tf_wdf as wdf
    from tf_wdf import tf
    g = golden("g1_rc_lowpass.npz")
    model = make_lpf_model(wdf, tf)
    data_in = np.array([g["x"]])
    data_target = np.transpose(np.array([g["target"]]))
    loss_func = tf.keras.losses.MeanSquaredError()
    with tf.GradientTape() as tape:                           # lpf.py:87-90
        outs = model.forward(data_in)[..., 0]
        loss = loss_func(outs, data_target)
    grads = tape.gradient(loss, model.trainable_variables)
    assert tuple(outs.shape) == (1280, 1)
    assert np.max(np.abs(outs.numpy()[:, 0] - g["y_f64"])) < 2e-6
    assert abs(float(loss) - float(g["loss_f64"])) < 1e-6
    assert abs(float(grads[0]) - float(g["dC_f64"])) < 2e-3 * abs(float(g["dC_f64"]))    # [C1.C, R1.R]
    assert abs(float(grads[1]) - float(g["dR_f64"])) < 2e-3 * abs(float(g["dR_f64"]))
    # second forward without reset: state carried like the reference (lpf.py has no reset())
    outs2 = model.forward(data_in)[..., 0]
    assert np.max(np.abs(outs2.numpy()[:, 0] - g["y_second_call_f64"])) < 2e-6


def test_lpf_script_training_converges(golden):
    """lpf.py:77-113 verbatim loop, 100 epochs: fc -> ~720 Hz (RC_lpf.png)."""
    import tf_wdf as wdf
    from tf_wdf import tf
    g = golden("g1_rc_lowpass.npz")
    model = make_lpf_model(wdf, tf)
    data_in = np.array([g["x"]])
    data_target = np.transpose(np.array([g["target"]]))
    loss_func = tf.keras.losses.MeanSquaredError()
    R_optimizer = tf.keras.optimizers.Adam(learning_rate=25.0)
    C_optimizer = tf.keras.optimizers.Adam(learning_rate=10.0e-9)
    for epoch in range(100):
        with tf.GradientTape() as tape:
            outs = model.forward(data_in)[..., 0]
            loss = loss_func(outs, data_target)
        grads = tape.gradient(loss, model.trainable_variables)
        R_optimizer.apply_gradients([(grads[1], model.R1.R)])
        C_optimizer.apply_gradients([(grads[0], model.C1.C)])
    final_freq = 1.0 / (2 * np.pi * model.R1.R * model.C1.C)
    assert float(loss) < 2e-4
    assert 650.0 < float(final_freq) < 800.0


def test_voltage_divider_script_loop(golden):
    import tf_wdf as wdf
    from tf_wdf import tf
    g = golden("g2_voltage_divider.npz")

    class Model(tf.Module):                                   # voltage_divider.py:17-46
        def __init__(self):
            super(Model, self).__init__()
            self.Vs = wdf.IdealVoltageSource()
            self.R1 = wdf.Resistor(2.0e3, True)
            self.R2 = wdf.Resistor(100.0, True)
            self.S1 = wdf.Series(self.R1, self.R2)
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
                output = wdf.voltage(self.R1)
                output_sequence = output_sequence.write(i, output)
            return output_sequence.stack()

    model = Model()
    data_in = np.array([g["x"]])
    with tf.GradientTape() as tape:
        outs = model.forward(data_in)[..., 0]
        loss = tf.keras.losses.MeanSquaredError()(outs, np.transpose(data_in * 0.5))
    grads = tape.gradient(loss, model.trainable_variables)
    assert np.max(np.abs(outs.numpy()[:, 0] - g["y_f64"])) < 1e-6
    assert abs(float(grads[0]) - float(g["dR1_f64"])) < 2e-3 * abs(float(g["dR1_f64"]))
    assert abs(float(grads[1]) - float(g["dR2_f64"])) < 2e-3 * abs(float(g["dR2_f64"]))


def test_clipper_pot_script_loop(golden):
    """ClipperModel.forward of clipper_pot.py:94-127 (per-sample R channel, DenseRootModel root)
    and its loss (clipper_pot.py:141-177, :245-248) against the golden of the 2x8 network."""
    import tf_wdf as wdf
    from tf_wdf import tf
    from layers import DenseRootModel, DenseLayer
    from test_gpu_mlp_root import model_json
    g = golden("g3_mlp_clipper.npz")
    name = "2x8"
    C_val = float(g["C"])

    class ClipperModel(tf.Module):
        def __init__(self, json):
            super(ClipperModel, self).__init__()
            self.Vs = wdf.ResistiveVoltageSource(45.0e3)
            self.C = wdf.Capacitor(C_val, FS)
            self.P1 = wdf.Parallel(self.Vs, self.C)
            self.model = DenseRootModel(json)

        def forward(self, input):  # noqa: A002
            sequence_length = input.shape[1]
            input = tf.cast(tf.expand_dims(input, axis=-1), dtype=tf.float32)  # noqa: A001
            output_sequence = tf.TensorArray(dtype=tf.float32, size=sequence_length, clear_after_read=False)
            self.Vs.reset()
            self.C.reset()
            for i in range(sequence_length):
                self.Vs.set_voltage(input[:, i, 0:1])
                self.Vs.set_resistance(input[:, i, 1:2])
                self.P1.calc_impedance()
                model_in = tf.concat((self.P1.reflected(), tf.math.log(self.P1.R)), axis=1)
                self.model.incident(tf.transpose(model_in, perm=[0, 2, 1]))
                self.P1.incident(-1 * self.model.reflected())
                output = wdf.voltage(self.C)
                output_sequence = output_sequence.write(i, output)
            output_sequence = output_sequence.stack()
            return output_sequence

    eps = np.finfo(float).eps

    def esr_loss(target_y, predicted_y, emphasis_func=lambda x: x):   # clipper_pot.py:148-156
        target_yp = emphasis_func(target_y)
        pred_yp = emphasis_func(predicted_y)
        mse = tf.math.reduce_sum(tf.math.square(target_yp - pred_yp))
        energy = tf.math.reduce_sum(tf.math.square(target_yp))
        loss_unnorm = mse / tf.cast(energy + eps, tf.float32)
        N = tf.cast((tf.shape(target_y)[0] * tf.shape(target_y)[1]), tf.float32)
        return tf.sqrt(loss_unnorm / N)

    mse_loss = tf.keras.losses.MeanSquaredError()
    loss_func = lambda target, pred: mse_loss(target, pred) + esr_loss(target, pred)  # noqa: E731

    model = ClipperModel(model_json(g, name))
    train_X, train_Y = g["x"], tf.constant(g["target"]).cuda()
    skip_samples = int(g["skip"])
    with tf.GradientTape() as tape:                           # clipper_pot.py:246-248
        outs = tf.transpose(model.forward(train_X)[..., 0], perm=[1, 0, 2])
        loss = loss_func(outs[:, skip_samples:, :], train_Y[:, skip_samples:, :])
    grads = tape.gradient(loss, model.trainable_variables)
    assert tuple(outs.shape) == (4, 256, 1)
    assert np.max(np.abs(outs.numpy()[:, :, 0].T - g[f"{name}_y_f64"])) < 3e-5
    assert abs(float(loss) - float(g[f"{name}_loss_f64"])) < 2e-5
    dense = [l for l in model.model.layers if isinstance(l, DenseLayer)]
    order = []
    for d in dense:
        order += [d.kernel, d.bias]
    tv = list(model.trainable_variables)
    got = np.concatenate([grads[next(i for i, v in enumerate(tv) if v is p)].numpy().ravel() for p in order])
    ref = g[f"{name}_grad_f64"]
    assert np.max(np.abs(got - ref)) < 2e-3 * np.max(np.abs(ref))
    # Adam step as clipper_pot.py:180,268
    optimizer = tf.keras.optimizers.Adam(learning_rate=0.0001, beta_1=0.5, beta_2=0.999)
    before = dense[0].kernel.numpy().copy()
    optimizer.apply_gradients(zip(grads, model.trainable_variables))
    assert np.max(np.abs(dense[0].kernel.numpy() - before)) > 0


def test_diode_pair_in_a_script_style_loop(golden):
    """The north-star variant written as a script loop (DiodeClipperWDF.cpp:24-28 order)."""
    import tf_wdf as wdf
    from tf_wdf import tf
    g = golden("g6_diode_clipper.npz")
    Is, nVt, R, C = [float(v) for v in g["theta"]]

    class Model(tf.Module):
        def __init__(self):
            super().__init__()
            self.Vs = wdf.ResistiveVoltageSource(R, trainable=True)
            self.C = wdf.Capacitor(C, FS, trainable=True)
            self.P1 = wdf.Parallel(self.Vs, self.C)
            self.dp = wdf.DiodePair(self.P1, Is, Vt=nVt, trainable=True)

        def forward(self, input):  # noqa: A002
            sequence_length = input.shape[1]
            input = tf.cast(tf.expand_dims(input, axis=-1), dtype=tf.float32)  # noqa: A001
            out = tf.TensorArray(dtype=tf.float32, size=sequence_length, clear_after_read=False)
            self.Vs.reset()
            self.C.reset()
            self.P1.calc_impedance()
            self.dp.calc_impedance()
            for i in range(sequence_length):
                self.Vs.set_voltage(input[:, i])
                self.dp.incident(self.P1.reflected())
                self.P1.incident(self.dp.reflected())
                out = out.write(i, wdf.voltage(self.C))
            return out.stack()

    m = Model()
    with tf.GradientTape() as tape:
        y = m.forward(g["x"])[..., 0]
        loss = tf.reduce_mean(tf.square(y - tf.constant(g["target"]).cuda()))
    grads = tape.gradient(loss, [m.dp.Is, m.dp.nVt, m.Vs.R, m.C.C])
    assert np.max(np.abs(y.numpy() - g["y_1u1d_f64"])) < 3e-5
    got, ref = np.array([float(v) for v in grads]), g["grad_1u1d_f64"]
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 2e-3, (got, ref)
