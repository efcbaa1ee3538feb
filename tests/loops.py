"""Per-sample loops written by hand for the loop-recorder tests (compat tier): circuits the two
simple_circuits scripts do NOT contain, driven sample by sample through the drop-in tf_wdf API the way
a user script would -- set the source, send the waves up and down the tree, read a voltage, write it
into a TensorArray.  The reference's own script classes are exercised elsewhere: executed in the
build container (tests/test_trace_cpu.py, AST-extracted from the checkout) and, on the GPU box,
through the programs recorded from them (tests/golden/g7_recorded_programs.npz)."""


def unroll(tf, n_steps, one_step):
    """The time loop of a user script: out[n] = one_step(n) for every sample, stacked [T, ...]."""
    ta = tf.TensorArray(dtype=tf.float32, size=n_steps, clear_after_read=False)
    for n in range(n_steps):
        ta = ta.write(n, one_step(n))
    return ta.stack()


class BridgedLadder:
    """Ideal source -> Ra -> (Ca parallel to (Rb -> Cb)): two capacitor states, no inverter, the
    voltage across Cb is the output.  Trainable Ra, Rb, Ca, Cb."""

    def __init__(self, wdf, fs, Ra=2.2e3, Ca=47.0e-9, Rb=6.8e3, Cb=10.0e-9):
        self.wdf = wdf
        self.Ra, self.Rb = wdf.Resistor(Ra, True), wdf.Resistor(Rb, True)
        self.Ca, self.Cb = wdf.Capacitor(Ca, fs, True), wdf.Capacitor(Cb, fs, True)
        self.top = wdf.Series(self.Ra, wdf.Parallel(self.Ca, wdf.Series(self.Rb, self.Cb)))
        self.src = wdf.IdealVoltageSource()
        self.params = [self.Ra.R, self.Ca.C, self.Rb.R, self.Cb.C]

    def reset(self):
        self.Ca.reset()
        self.Cb.reset()

    def run(self, x):
        """x [B,T] -> [T,B,1]"""
        wdf, tf = self.wdf, self.wdf.tf
        seq = tf.cast(tf.expand_dims(x, axis=-1), dtype=tf.float32)
        self.top.calc_impedance()

        def step(n):
            self.src.set_voltage(seq[:, n])
            self.src.incident(self.top.reflected())
            self.top.incident(self.src.reflected())
            return wdf.voltage(self.Cb)

        return unroll(tf, int(seq.shape[1]), step)


class HighPassClipper:
    """HPFDiodeClipper.h:28-32 tree, Parallel(R, Series(Vs, C)), under a diode-pair root; the output is
    the voltage across R.  Static impedances (calc_impedance once, before the loop)."""

    def __init__(self, wdf, fs, R=33.0e3, Rs=1.0e3, C=22.0e-9, Is=4.352e-9, nVt=25.85e-3 * 1.906, n_up=2, n_down=3):
        self.wdf = wdf
        self.R = wdf.Resistor(R, True)
        self.Vs = wdf.ResistiveVoltageSource(Rs, trainable=True)
        self.C = wdf.Capacitor(C, fs, True)
        self.top = wdf.Parallel(self.R, wdf.Series(self.Vs, self.C))
        self.dp = wdf.DiodePair(self.top, Is, Vt=nVt, N_up=n_up, N_down=n_down, trainable=True)
        self.params = [self.R.R, self.Vs.R, self.C.C, self.dp.Is, self.dp.nVt]

    def run(self, x):
        wdf, tf = self.wdf, self.wdf.tf
        seq = tf.cast(tf.expand_dims(x, axis=-1), dtype=tf.float32)
        self.Vs.reset()
        self.C.reset()
        self.top.calc_impedance()
        self.dp.calc_impedance()

        def step(n):
            self.Vs.set_voltage(seq[:, n])
            self.dp.incident(self.top.reflected())
            self.top.incident(self.dp.reflected())
            return wdf.voltage(self.R)

        return unroll(tf, int(seq.shape[1]), step)


class PotClipper:
    """The clipper with a pot: Parallel(ResistiveVoltageSource, Capacitor) whose source resistance
    arrives as the second input channel, so the port impedances are recomputed every sample; the root
    is either a layers.DenseRootModel fed (incident wave, log port resistance) -- its output is the
    NEGATED reflected wave -- or the analytic DiodePair."""

    def __init__(self, wdf, fs, C, mlp_json=None, diode=None):
        self.wdf = wdf
        self.Vs = wdf.ResistiveVoltageSource(45.0e3)
        self.C = wdf.Capacitor(C, fs, trainable=diode is not None)
        self.P = wdf.Parallel(self.Vs, self.C)
        self.mlp = self.dp = None
        if mlp_json is not None:
            from layers import DenseRootModel
            self.mlp = DenseRootModel(mlp_json)
        else:
            self.dp = wdf.DiodePair(self.P, diode[0], Vt=diode[1], trainable=True)

    @property
    def trainable_variables(self):
        return list(self.mlp.trainable_variables) if self.mlp is not None else [self.dp.Is, self.dp.nVt, self.C.C]

    def run(self, data):
        """data [B,T,2] = (input voltage, pot resistance) -> [T,B,1,1]"""
        wdf, tf = self.wdf, self.wdf.tf
        seq = tf.cast(tf.expand_dims(data, axis=-1), dtype=tf.float32)
        self.Vs.reset()
        self.C.reset()

        def step(n):
            self.Vs.set_voltage(seq[:, n, 0:1])
            self.Vs.set_resistance(seq[:, n, 1:2])
            self.P.calc_impedance()
            up = self.P.reflected()
            if self.mlp is not None:
                feats = tf.concat((up, tf.math.log(self.P.R)), axis=1)
                self.mlp.incident(tf.transpose(feats, perm=[0, 2, 1]))
                self.P.incident(-1 * self.mlp.reflected())
            else:
                self.dp.calc_impedance()
                self.dp.incident(up)
                self.P.incident(self.dp.reflected())
            return wdf.voltage(self.C)

        return unroll(tf, int(seq.shape[1]), step)


def mse_plus_esr(tf, first, second, eps):
    """MSE + error-to-signal ratio with the energy taken from the FIRST argument: S/n + sqrt(S/(E+eps)/n),
    S = sum (first - second)^2, E = sum first^2, n = the two leading dimensions' product.  (Called with
    (model output, target) it is the training loss of clipper_pot.py:146-156,177,248.)"""
    S = tf.math.reduce_sum(tf.math.square(first - second))
    E = tf.math.reduce_sum(tf.math.square(first))
    n = tf.cast(tf.shape(first)[0] * tf.shape(first)[1], tf.float32)
    count = 1
    for d in first.shape:
        count *= int(d)
    return S / float(count) + tf.sqrt(S / tf.cast(E + eps, tf.float32) / n)
