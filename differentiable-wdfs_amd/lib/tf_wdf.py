"""Drop-in for wdf_py/lib/tf_wdf.py: differentiable WDF elements, executed on MI355X.

Same module surface as the reference (tf_wdf.py:8-214): `voltage`, `IdealVoltageSource`,
`ResistiveVoltageSource`, `Resistor`, `Capacitor`, `Series`, `Parallel`, `Inverter`, the
same constructor arguments, the same duck-typed protocol (attributes a, b, R, C, FS, z, P1,
P2, p1R, b_diff, b_temp, Vs; methods calc_impedance / incident / reflected / reset /
set_voltage / set_resistance) and the same variable constraints.  `tf` is exported too,
because the reference scripts take TensorFlow from this module (`from tf_wdf import tf`).

New surface (not in the reference):
  DiodePair(next, Is, Vt=25.85e-3, nDiodes=1, N_up=1, N_down=1, trainable=False)
      analytic Wright-omega diode-pair root (formula diode_pretraining.py:39-60, element
      protocol Toms917DiodePair.h:21-59) with trainable Is and nVt;
  Circuit(top, root, probe) / run(...) / Circuit.mse(x, target)
      the fast tier: lowers the WHOLE per-sample loop the scripts own (lpf.py:39-46,
      clipper_pot.py:113-124) to one HIP kernel launch, and its tape.gradient to one reverse
      sweep.

How the elements execute.  Every method below is plain wave arithmetic; the values flowing
through are either
  * unit-vector probes (Circuit lowering): because every element is linear in the waves,
    one probed step yields the state-space matrices the HIP kernel runs (csrc/wdf_statespace.h),
    differentiably w.r.t. R and C; or
  * recorded symbols (wdf_hip.trace): a reference script's own `for i in range(T)` loop is
    recorded and lowered to the same kernel when it calls TensorArray.stack().
Nothing here computes audio samples on the host.
"""
import numpy as np
import torch

from wdf_hip import compat_tf as tf
from wdf_hip import lowering as _lowering
from wdf_hip import trace as _trace

tf.get_logger().setLevel("WARN")


def voltage(wdf):
    '''Voltage across a WDF element: (a + b)/2  (tf_wdf.py:8-10).'''
    return (wdf.a + wdf.b) * 0.5


class _Element(tf.Module):
    """What every one-port / adaptor shares (tf_wdf.py:16-17 and the like): zero-initialised
    incident and reflected waves, `incident` storing the wave, a no-op `calc_impedance`."""

    def __init__(self):
        super().__init__()
        self.a = tf.Variable(initial_value=tf.zeros(1), name="incident_wave", trainable=False)
        self.b = tf.Variable(initial_value=tf.zeros(1), name="reflected_wave", trainable=False)

    def calc_impedance(self):
        pass

    def incident(self, x):
        self.a = x

    def _emit(self, wave):
        self.b = wave
        return wave


class _Source:
    def set_voltage(self, voltage):
        # a sample of a sequence tensor starts / continues a recorded loop (wdf_hip.trace)
        self.Vs = _trace.bind_voltage(voltage)


def _clipped(lo, hi):
    return lambda z: tf.clip_by_value(z, lo, hi)


class IdealVoltageSource(_Source, _Element):
    '''Ideal voltage source ROOT: b = -a + 2 Vs  (tf_wdf.py:13-28).'''

    def reflected(self):
        return self._emit(-self.a + 2.0 * self.Vs)


class ResistiveVoltageSource(_Source, _Element):
    '''Resistive voltage source leaf: b = Vs  (tf_wdf.py:31-59).  R defaults to 1e-9, may be
    trainable, has no constraint; set_resistance REPLACES R by a tensor (:51-52).'''

    def __init__(self, initial_R=1.0e-9, trainable=False):
        super().__init__()
        self.R = tf.Variable(initial_value=initial_R, name="resistance", trainable=trainable)

    def reset(self):
        self.a = tf.zeros(1)

    def set_resistance(self, resistance):
        self.R = _trace.bind_resistance(self, resistance)

    def reflected(self):
        return self._emit(self.Vs * tf.ones_like(self.a))


class Resistor(_Element):
    '''Resistor leaf: b = 0  (tf_wdf.py:62-88).  R is clipped to [180, 1e6] by the optimizer.'''

    def __init__(self, initial_R, trainable=False):
        super().__init__()
        self.R = tf.Variable(initial_value=initial_R, name="resistance", dtype=tf.float32, trainable=trainable,
                             constraint=_clipped(180.0, 1.0e6))

    def set_resistance(self, resistance):
        self.R = resistance

    def reflected(self):
        return self._emit(tf.zeros_like(self.a))


class Capacitor(_Element):
    '''Capacitor leaf with one state: b = z, z <- a; port resistance 1/(2 C FS) recomputed
    differentiably in calc_impedance (tf_wdf.py:91-126).  C is clipped to [1e-13, 1].'''

    def __init__(self, initial_C, FS, trainable=False):
        super().__init__()
        self.FS = FS
        self.C = tf.Variable(initial_value=initial_C, name="capacitance", dtype=tf.float32, trainable=trainable,
                             constraint=_clipped(0.1e-12, 1.0))
        self.R = tf.Variable(initial_value=1.0 / (2.0 * initial_C * FS), name="impedance", trainable=False)
        self.z = tf.Variable(initial_value=0.0, name="state", trainable=False)

    def calc_impedance(self):
        self.R = tf.math.reciprocal(self.C * (2.0 * self.FS))

    def reset(self):
        self.z = tf.zeros(1)

    def incident(self, x):
        self.a = self.z = x

    def reflected(self):
        return self._emit(_trace.state(self))    # = self.z (the step's state symbol while a loop is recorded)


class _Adaptor(_Element):
    def __init__(self, *ports):
        super().__init__()
        for i, port in enumerate(ports, start=1):
            setattr(self, f"P{i}", port)
        self._n_ports = len(ports)

    def _children_impedance(self):
        for i in range(1, self._n_ports + 1):
            getattr(self, f"P{i}").calc_impedance()


class Series(_Adaptor):
    '''3-port series adaptor, port 3 reflection-free (tf_wdf.py:129-155).'''

    def __init__(self, P1, P2):
        super().__init__(P1, P2)

    def calc_impedance(self):
        self._children_impedance()
        self.R = self.P1.R + self.P2.R
        self.p1R, self.p2R = self.P1.R / self.R, self.P2.R / self.R

    def incident(self, x):
        # uses the children's waves stored by the last reflected() (:148)
        down1 = self.P1.b - self.p1R * (x + self.P1.b + self.P2.b)
        self.P1.incident(down1)
        self.P2.incident(-(x + down1))
        self.a = x

    def reflected(self):
        return self._emit(-(self.P1.reflected() + self.P2.reflected()))


class Parallel(_Adaptor):
    '''3-port parallel adaptor (tf_wdf.py:158-192); b_diff / b_temp carry from reflected()
    to incident().'''

    def __init__(self, P1, P2):
        super().__init__(P1, P2)

    def calc_impedance(self):
        self._children_impedance()
        G1, G2 = 1.0 / self.P1.R, 1.0 / self.P2.R
        self.R = 1.0 / (G1 + G2)
        self.p1R = G1 / (G1 + G2)

    def incident(self, x):
        down2 = x + self.b_temp
        self.P1.incident(self.b_diff + down2)
        self.P2.incident(down2)
        self.a = x

    def reflected(self):
        up1, up2 = self.P1.reflected(), self.P2.reflected()
        self.b_diff = up2 - up1
        self.b_temp = -self.p1R * self.b_diff
        return self._emit(up2 + self.b_temp)


class Inverter(_Adaptor):
    '''2-port polarity inverter (tf_wdf.py:195-214).'''

    def __init__(self, P1):
        super().__init__(P1)

    def calc_impedance(self):
        self._children_impedance()
        self.R = self.P1.R

    def incident(self, x):
        self.P1.incident(-x)
        self.a = x

    def reflected(self):
        return self._emit(-self.P1.reflected())


class DiodePair(_Element):
    '''Analytic diode-pair ROOT (new API, wdf_py style).

    b = a - 2 nVt lam (mu0 w(log(R Is/(mu0 nVt)) + lam a/(mu0 nVt))
                       - mu1 w(log(R Is/(mu1 nVt)) - lam a/(mu1 nVt)))
    with w = Wright omega, lam = sign(a), (mu0, mu1) = (N_down, N_up) if a >= 0 else
    (N_up, N_down): diode_pretraining.py:39-60 (Werner et al. eqn 45).  N_up = N_down = 1 is
    the C++ Toms917DiodePairT (Toms917DiodePair.h:51-59, eqn 39) with Vt <- nDiodes*Vt (:31).

    `next` is the tree the pair terminates (like the C++ constructor, :21); R is its port
    resistance.  Trainable variables: Is and nVt (= nDiodes * Vt).  The solve itself runs in
    the HIP kernel (csrc/wdf_omega.h); reflected() here only records the root.'''

    def __init__(self, next, Is, Vt=25.85e-3, nDiodes=1.0, N_up=1, N_down=1, trainable=False):  # noqa: A002
        super().__init__()
        if int(N_up) < 1 or int(N_down) < 1:
            raise ValueError("N_up and N_down must be >= 1")
        self.next = next
        self.N_up, self.N_down = int(N_up), int(N_down)
        self.Is = tf.Variable(initial_value=Is, name="saturation_current", dtype=tf.float32, trainable=trainable,
                              constraint=_clipped(1.0e-15, 1.0e-3))
        self.nVt = tf.Variable(initial_value=float(nDiodes) * float(Vt), name="n_thermal_voltage",
                               dtype=tf.float32, trainable=trainable,
                               constraint=_clipped(1.0e-3, 1.0))

    def calc_impedance(self):
        # Toms917DiodePair.h:37-42: the root's constants follow the tree's port resistance
        self.R = self.next.R

    def reflected(self):
        return self._emit(_lowering.diode_pair_reflected(self))


# ---- fast tier ------------------------------------------------------------------------------
Circuit = _lowering.Circuit


def run(top, root, probe, x, **kwargs):
    """One-shot fast tier: Circuit(top, root, probe)(x).  x: [B,T] or [B,T,n_in] float32 on
    the GPU; returns the probed voltage [T,B] (TensorArray.stack() layout)."""
    return Circuit(top, root, probe)(x, **kwargs)
