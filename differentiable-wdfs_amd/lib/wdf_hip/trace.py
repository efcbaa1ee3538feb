"""Loop recorder: runs a reference script's OWN per-sample loop symbolically and lowers it to
one HIP launch when the script calls TensorArray.stack().

The reference scripts own the time loop (lpf.py:39-46, voltage_divider.py:35-42,
clipper_pot.py:113-124):

    for i in range(sequence_length):
        self.Vs.set_voltage(input[:, i])
        self.Vs.incident(self.I1.reflected())
        self.I1.incident(self.Vs.reflected())
        output_sequence = output_sequence.write(i, wdf.voltage(self.C1))
    output_sequence = output_sequence.stack()

To run them unchanged, `input[:, i]` (an index into a compat_tf.Tensor) returns a sample that
remembers where it came from; the elements of tf_wdf.py, handed such a sample, compute with
`Wave`s instead of numbers.  A Wave is an affine form over the symbols of ONE time step --
capacitor states z_k at the start of the step, the step's input samples x_j, the root's
output b, and 1 -- with coefficients that are float64 torch scalars (so they carry autograd
back to R and C).  Every adaptor / one-port is linear in the waves (tf_wdf.py:31-214), so the
loop body maps Waves to Waves, and after one iteration the recorder holds exactly the
matrices of   a = ca.z + da.x ;  b = root(a) ;  z' = A z + Bx x + E b ;  y = cy.z + dy.x + fy b.
The remaining T-1 iterations of the script's loop only confirm that every step is that same
map.  stack() then launches the state-space / clipper kernels (csrc/) for all T steps and
returns a real tensor shaped like TensorArray.stack()'s; Capacitor.z is left holding the
final state so a script that never calls reset() (lpf.py) carries it into its next forward().

A per-sample resistance (Vs.set_resistance(input[:, i, 1:2]); P1.calc_impedance() every step,
clipper_pot.py:116-117) makes the adaptor coefficient p = G1/(G1+G2) data dependent; Waves
then carry coefficients c0 + c1 p, which is enough for the clipper topology, and the
lowering target is the clipper kernel that streams r.

Anything the kernels cannot express (a product of two waves outside the root, reading a
wave left over from the previous step, steps that differ) raises WdfTraceError: the loop is
never silently evaluated sample by sample on the host.
"""
import torch

from . import binding
from . import compat_tf as tf

NB = 16                      # basis slots: 0 = constant one, then states, inputs, root output


class WdfTraceError(binding.WdfHipError):
    pass


_current = None              # the Recorder of the forward() being recorded, if any


def _device():
    """Device the recorded loop is lowered to (tests on the CPU box replace this)."""
    binding.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def current():
    return _current


# ------------------------------------------------------------------------------ dynamic resistance
class RExpr:
    """Symbolic per-sample resistance algebra: just enough structure for
    Parallel.calc_impedance (G1 = 1/R; G = G1 + G2; R = 1/G; p1R = G1/G) and log(R)."""

    def __init__(self, kind, src=None, elem=None):
        self.kind, self.src, self.elem = kind, src, elem      # kind in {r, g1, g, rp, p, logrp}

    def __rtruediv__(self, other):                             # 1.0 / R ,  1.0 / G
        if self.kind == "r":
            return RExpr("g1", self.src, self.elem)
        if self.kind == "g":
            return RExpr("rp", self.src, self.elem)
        raise WdfTraceError(f"unsupported arithmetic on a per-sample resistance ({self.kind})")

    def __add__(self, other):                                  # G1 + G2
        if self.kind == "g1" and not isinstance(other, RExpr):
            return RExpr("g", self.src, self.elem)
        raise WdfTraceError("unsupported arithmetic on a per-sample resistance")

    __radd__ = __add__

    def __truediv__(self, other):                              # G1 / G
        if self.kind == "g1" and isinstance(other, RExpr) and other.kind == "g":
            return RExpr("p", self.src, self.elem)
        raise WdfTraceError("unsupported arithmetic on a per-sample resistance")

    def __neg__(self):
        if self.kind == "p":
            return RExpr("-p", self.src, self.elem)
        raise WdfTraceError("unsupported arithmetic on a per-sample resistance")

    def __wdf_log__(self):
        if self.kind == "rp":
            return RExpr("logrp", self.src, self.elem)
        raise WdfTraceError("log of a per-sample resistance expression other than the port resistance")


# ------------------------------------------------------------------------------ waves
def _scalar(v):
    """Accept python numbers and 1-element tensors as scalar coefficients."""
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, torch.Tensor) and v.numel() == 1:
        return v.as_subclass(torch.Tensor).double().reshape(())
    return None


# shared constants: coefficient vectors are never modified in place (every operation builds a new one)
_EYE = torch.eye(NB, dtype=torch.float64)
_EYE_ROWS = [_EYE[i] for i in range(NB)]      # ONE object per basis vector: the replay memo keys on identity
_ZERO = torch.zeros(NB, dtype=torch.float64)


class Wave:
    __slots__ = ("rec", "step", "c0", "c1")
    __array_priority__ = 1000

    def __init__(self, rec, c0, c1=None):
        self.rec, self.step, self.c0, self.c1 = rec, rec.step, c0, c1

    # -- construction helpers
    @staticmethod
    def basis(rec, slot):
        return Wave(rec, _EYE_ROWS[slot])

    def _fresh(self):
        if self.step != self.rec.step:
            raise WdfTraceError("a wave computed in a previous time step is read in this one; only "
                                "Capacitor.z may carry values across steps (tf_wdf.py:120-126)")
        return self

    def _coerce(self, other):
        if isinstance(other, Wave):
            return other._fresh()
        if isinstance(other, (int, float)):                   # constants: one vector per value (replay memo keys on identity)
            return Wave(self.rec, self._memo(("const", float(other)), (), lambda: _unit0() * float(other)))
        s = _scalar(other)
        if s is None:
            raise WdfTraceError(f"cannot combine a recorded wave with {type(other).__name__} of shape "
                                f"{tuple(getattr(other, 'shape', ()))}")
        return Wave(self.rec, _unit0() * s)

    # -- affine arithmetic.  REPLAY: every time step of a script's loop repeats the first one's operations on
    # waves whose coefficient vectors are the SAME objects (basis rows are shared, results come out of this
    # memo), so from the second step on an operation is one dictionary look-up keyed on operand identity
    # instead of torch arithmetic on 16-vectors.  The memo keeps its operands alive (no id is recycled).
    def _memo(self, key, keep, make):
        memo = self.rec.memo
        hit = memo.get(key)
        if hit is None:
            hit = memo[key] = (make(), keep)
        return hit[0]

    def __add__(self, other):
        if isinstance(other, Wave):
            o = other
            if o.step != self.rec.step or self.step != self.rec.step:
                o._fresh(); self._fresh()
            hit = self.rec.memo.get(("add", id(self.c0), id(self.c1), id(o.c0), id(o.c1)))
            if hit is not None:
                return Wave(self.rec, hit[0][0], hit[0][1])
            c0, c1 = self._memo(("add", id(self.c0), id(self.c1), id(o.c0), id(o.c1)), (self.c0, self.c1, o.c0, o.c1),
                                lambda: (self.c0 + o.c0,
                                         self.c1 if o.c1 is None else (o.c1 if self.c1 is None else self.c1 + o.c1)))
            return Wave(self.rec, c0, c1)
        return self + self._coerce(other)

    __radd__ = __add__

    def __neg__(self):
        self._fresh()
        c0, c1 = self._memo(("neg", id(self.c0), id(self.c1)), (self.c0, self.c1),
                            lambda: (-self.c0, None if self.c1 is None else -self.c1))
        return Wave(self.rec, c0, c1)

    def __sub__(self, other):
        return self + (-self._coerce(other))

    def __rsub__(self, other):
        return self._coerce(other) + (-self)

    def __mul__(self, other):
        self._fresh()
        if isinstance(other, RExpr):
            if other.kind not in ("p", "-p"):
                raise WdfTraceError("a wave may only be scaled by the adaptor coefficient p1R of a per-sample "
                                    "resistance (tf_wdf.py:190)")
            if self.c1 is not None and self._memo(("deg2", id(self.c1)), (self.c1,), lambda: bool(torch.any(self.c1 != 0))):
                raise WdfTraceError("per-sample resistance: coefficient of degree 2 in p1R (only the diode-clipper "
                                    "topology of clipper_pot.py is supported)")
            self.rec.dyn_elem = other.elem
            sgn = -1.0 if other.kind == "-p" else 1.0
            c1 = self._memo(("mulr", id(self.c0), sgn), (self.c0,), lambda: sgn * self.c0)
            return Wave(self.rec, _ZERO, c1)
        if isinstance(other, Wave):
            raise WdfTraceError("product of two waves outside the root: not a WDF adaptor operation")
        if isinstance(other, (int, float)):
            key, keep = ("mulf", id(self.c0), id(self.c1), float(other)), (self.c0, self.c1)
            hit = self.rec.memo.get(key)
            if hit is not None:
                return Wave(self.rec, hit[0][0], hit[0][1])
        else:                       # a scalar tensor (a component value / adaptor coefficient): same object, same version
            with torch._C.DisableTorchFunctionSubclass():      # (a Tensor property read through the subclass protocol costs 2 us)
                ver = getattr(other, "_version", 0)
            key, keep = ("mult", id(self.c0), id(self.c1), id(other), ver), (self.c0, self.c1, other)

        def make():
            s = _scalar(other)
            if s is None:
                raise WdfTraceError(f"cannot scale a recorded wave by {type(other).__name__} of shape "
                                    f"{tuple(getattr(other, 'shape', ()))}")
            return self.c0 * s, None if self.c1 is None else self.c1 * s

        c0, c1 = self._memo(key, keep, make)
        return Wave(self.rec, c0, c1)

    __rmul__ = __mul__

    def __truediv__(self, other):
        s = _scalar(other)
        if s is None:
            raise WdfTraceError("a recorded wave can only be divided by a scalar")
        return self * (1.0 / s)

    # torch.Tensor <op> Wave lands here
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", "")
        a = args
        if name in ("mul", "__mul__", "multiply") and len(a) == 2:
            return a[1] * a[0] if isinstance(a[1], Wave) else a[0] * a[1]
        if name in ("add", "__add__") and len(a) == 2:
            return a[1] + a[0] if isinstance(a[1], Wave) else a[0] + a[1]
        if name in ("sub", "__sub__", "subtract") and len(a) == 2:
            return a[1].__rsub__(a[0]) if isinstance(a[1], Wave) else a[0] - a[1]
        if name in ("rsub", "__rsub__") and len(a) == 2:
            return a[0].__rsub__(a[1]) if isinstance(a[0], Wave) else a[1] - a[0]
        if name in ("neg", "negative"):
            return -a[0]
        if name in ("div", "true_divide", "__truediv__") and isinstance(a[0], Wave):
            return a[0] / a[1]
        raise WdfTraceError(f"torch.{name} on a recorded wave is not a WDF adaptor operation")

    # -- hooks the compat tf namespace and the elements call
    def __wdf_like__(self, fn_name):                       # tf.zeros_like / tf.ones_like
        return 0.0 if "zeros" in fn_name else 1.0

    def __wdf_cast__(self, dtype):
        return self

    def __wdf_root__(self, root):                          # DiodePair.reflected()
        return self.rec.root_output(root, self._fresh(), None)

    def __wdf_concat__(self, values, axis):                # tf.concat((P1.reflected(), tf.math.log(P1.R)), axis=1)
        if len(values) != 2 or values[0] is not self:
            raise WdfTraceError("only concat((wave, log R)) -- the DenseRootModel input of clipper_pot.py:119 -- is supported")
        return ModelIn(self._fresh(), values[1])

    def __wdf_stack__(self, items):                        # TensorArray.stack()
        return self.rec.lower(items)

    @property
    def shape(self):
        return self.rec.sample_shape


_U0 = None


def _unit0():
    global _U0
    if _U0 is None:
        _U0 = torch.zeros(NB, dtype=torch.float64)
        _U0[0] = 1.0
    return _U0


class ModelIn:
    """The [B,1,2] input of DenseRootModel: (incident wave, log port resistance)."""

    def __init__(self, a, lr):
        self.a, self.lr = a, lr

    def __wdf_transpose__(self, perm):
        return self

    def __getitem__(self, idx):                            # layers.py:73  self.a = x[:, :, 0]
        return self.a

    def __wdf_root__(self, model):
        return self.a.rec.root_output(model, self.a, self.lr)


# ------------------------------------------------------------------------------ the recorder
class Recorder:
    def __init__(self, parent):
        self.parent = parent            # the [B,T,...] tensor the script indexes with [:, i(, c)]
        self.step = -1                  # index i of the current time step
        self.n_steps = 0
        self.states = []                # [(capacitor, z0 numeric tensor or None)]
        self.inputs = []                # channel keys, slot order
        self.rin = None                 # (elem, channel key) of a per-sample resistance
        self.dyn_elem = None
        self.root = None
        self.root_in = None             # Wave: a  (this step)
        self.root_lr = None
        self.sample_shape = None
        self.canon = None               # signature + waves of the first completed step
        self.pending_out = None
        self.slots = 1
        self.memo = {}                  # operation memo of the Wave arithmetic (replay)
        self.scalar_memo = {}           # component arithmetic of this forward pass (compat_tf.Tensor._scalar_memo)

    # -- slots
    def _slot(self):
        if self.slots >= NB:
            raise WdfTraceError("too many states / inputs for the recorder")
        s = self.slots
        self.slots += 1
        return s

    def begin_step(self, i):
        if i == self.step:
            return
        if self.step >= 0:
            self._close_step()
        self.step = i
        self.n_steps += 1
        self.root_in = self.root_lr = None

    # -- inputs
    def input_wave(self, sample):
        parent, i, key = sample._wdf_src
        self.begin_step(i)
        if self.sample_shape is None:
            self.sample_shape = tuple(sample.shape)
        for k, slot in self.inputs:
            if k == key:
                return Wave.basis(self, slot)
        slot = self._slot()
        self.inputs.append((key, slot))
        return Wave.basis(self, slot)

    def resistance(self, elem, sample):
        parent, i, key = sample._wdf_src
        self.begin_step(i)
        if self.rin is None:
            self.rin = (elem, key)
        elif self.rin[0] is not elem or self.rin[1] != key:
            raise WdfTraceError("only one per-sample resistance channel is supported (clipper_pot.py:116)")
        return RExpr("r", key, elem)

    # -- states
    def state(self, cap):
        z = cap.__dict__.get("z")
        for k, (c, _z0, slot) in enumerate(self.states):
            if c is cap:
                if isinstance(z, Wave) and z.step == self.step:
                    return z                                # incident() already ran this step
                if isinstance(z, Wave):
                    self._note_transition(k, z)
                w = Wave.basis(self, slot)
                cap.__dict__["z"] = w
                return w
        if isinstance(z, Wave):
            raise WdfTraceError("capacitor state is a wave recorded by another loop")
        slot = self._slot()
        self.states.append((cap, z, slot))
        w = Wave.basis(self, slot)
        cap.__dict__["z"] = w
        return w

    def _note_transition(self, k, z):
        self._trans = getattr(self, "_trans", {})
        self._trans[k] = z

    # -- root
    def root_output(self, root, a, lr):
        if self.root is None:
            self.root = root
        elif self.root is not root:
            raise WdfTraceError("more than one root in the recorded loop")
        self.root_in, self.root_lr = a, lr
        if not hasattr(self, "root_slot"):
            self.root_slot = self._slot()
        return Wave.basis(self, self.root_slot)

    # -- steps
    def note_output(self, wave):
        self.pending_out = wave

    def _step_record(self, final=False):
        """(state transition waves, root input, output) of the step that just finished."""
        trans = []
        for k, (cap, _z0, slot) in enumerate(self.states):
            z = cap.__dict__.get("z")
            if not isinstance(z, Wave) or z.step != self.step:
                raise WdfTraceError("a capacitor did not receive an incident wave in this step")
            trans.append(z)
        if self.pending_out is None or self.pending_out.step != self.step:
            raise WdfTraceError("no output was written for this time step")
        return trans, self.root_in, self.pending_out

    @staticmethod
    def _sig(waves):
        out = []
        for w in waves:
            if w is None:
                out.append(None)
                continue
            out.append(tuple(w.c0.tolist()))
            out.append(None if w.c1 is None else tuple(w.c1.tolist()))
        return tuple(out)

    @staticmethod
    def _ids(waves):
        return tuple((None if w is None else (id(w.c0), id(w.c1))) for w in waves)

    def _close_step(self):
        trans, a, y = self._step_record()
        waves = trans + [a, y]
        if self.canon is None:
            self.canon = (self._sig(waves), trans, a, y, self.root_lr)
            self.canon_ids = self._ids(waves)
            return
        ids = self._ids(waves)
        if ids == self.canon_ids:                    # replayed from the memo: literally the vectors of a step already checked
            return
        if self._sig(waves) != self.canon[0]:
            raise WdfTraceError(f"time step {self.step} is not the same linear map as the first step: the loop "
                                "cannot be lowered to one recursion")
        self.canon_ids = ids                         # (the first step differs in identity: its elements still held numbers)

    # -- lowering -------------------------------------------------------------------------------
    def lower(self, items):
        global _current
        try:
            return self._lower(items)
        finally:
            _current = None

    def _gather_inputs(self, keys, dev):
        """x [B,T,len(keys)] float32 on the device from the script's input tensor."""
        p = self.parent.as_subclass(torch.Tensor)
        cols = []
        for key in keys:
            idx = (slice(None), slice(None)) + key
            cols.append(p[idx].reshape(p.shape[0], p.shape[1]))
        return torch.stack(cols, dim=-1).to(device=dev, dtype=torch.float32).contiguous()

    def _lower(self, items):
        from . import engine, lowering, mlp_root
        dev = _device()
        T = len(items)
        if any(not isinstance(w, Wave) for w in items):
            raise WdfTraceError("TensorArray mixes recorded waves and tensors")
        self._close_step()
        if self.n_steps != T:
            raise WdfTraceError(f"{T} outputs were written but {self.n_steps} time steps were recorded")
        _sig, trans, a, y, lr = self.canon
        ns, ni = len(self.states), len(self.inputs)
        B = int(self.parent.shape[0])
        s_slots = [s for (_c, _z, s) in self.states]
        x_slots = [s for (_k, s) in self.inputs]
        b_slot = getattr(self, "root_slot", None)
        for w in trans + [y] + ([a] if a is not None else []):
            if bool(w.c0[0] != 0) or (w.c1 is not None and bool(w.c1[0] != 0)):
                raise WdfTraceError("constant (non-zero) wave sources are not supported by the kernels")
        # initial states: numeric z the capacitors held when the loop started
        z0 = None
        if ns:
            cols = []
            for (cap, zinit, _s) in self.states:
                zt = tf.convert(zinit if zinit is not None else 0.0).as_subclass(torch.Tensor).float().reshape(-1)
                cols.append(zt.expand(B) if zt.numel() == 1 else zt)
            z0 = torch.stack(cols).to(dev).contiguous()
            if not bool(torch.any(z0 != 0)):
                z0 = None
        root_kind = None if self.root is None else type(self.root).__name__

        def pick(w, slots):
            return w.c0[slots] if slots else torch.zeros(0, dtype=torch.float64)

        if self.rin is not None or any(w.c1 is not None for w in trans + [y] + ([a] if a is not None else [])):
            y_tb, zT = self._lower_dynamic(trans, a, y, lr, z0, dev, engine, mlp_root)
        elif root_kind == "DenseRootModel":
            y_tb, zT = self._lower_static_mlp(trans, a, y, lr, z0, dev, mlp_root)
        else:
            A = torch.stack([pick(w, s_slots) for w in trans]) if ns else torch.zeros(0, 0, dtype=torch.float64)
            Bx = torch.stack([pick(w, x_slots) for w in trans]) if ns else torch.zeros(0, ni, dtype=torch.float64)
            zero = torch.zeros((), dtype=torch.float64)
            E = torch.stack([w.c0[b_slot] if b_slot is not None else zero for w in trans]) if ns else torch.zeros(0, dtype=torch.float64)
            ca = pick(a, s_slots) if a is not None else torch.zeros(ns, dtype=torch.float64)
            da = pick(a, x_slots) if a is not None else torch.zeros(ni, dtype=torch.float64)
            cy, dy = pick(y, s_slots), pick(y, x_slots)
            fy = (y.c0[b_slot] if b_slot is not None else zero).reshape(1)
            coef = torch.cat([A.reshape(-1), Bx.reshape(-1), E, ca, da, cy, dy, fy]).to(device=dev, dtype=torch.float32)
            x = self._gather_inputs([k for (k, _s) in self.inputs], dev)
            if root_kind == "DiodePair":
                dp = self.root
                rootp = torch.stack([dp.Is.as_subclass(torch.Tensor).double().reshape(()),
                                     dp.nVt.as_subclass(torch.Tensor).double().reshape(()),
                                     tf.convert(dp.R).as_subclass(torch.Tensor).double().reshape(())]).to(device=dev, dtype=torch.float32)
                kind, n_up, n_down = binding.ROOT_DIODE_PAIR, dp.N_up, dp.N_down
            elif root_kind is None:
                rootp, kind, n_up, n_down = None, binding.ROOT_NONE, 1, 1
            else:
                raise WdfTraceError(f"unsupported root {root_kind}")
            y_tb, zT = lowering._StateSpaceFn.apply(coef, rootp, x, z0, ns, ni, kind, n_up, n_down, ns > 0)
        # leave the final states in the capacitors (a script that never resets carries them on)
        for k, (cap, _z, _s) in enumerate(self.states):
            cap.__dict__["z"] = zT[k].reshape((B,) + (1,) * (len(self.sample_shape) - 1)).as_subclass(tf.Tensor)
        out = y_tb.reshape((T,) + tuple(self.sample_shape))
        return out.as_subclass(tf.Tensor)

    # clipper topology with a per-sample resistance: coefficients are c0 + c1 p
    def _check_clipper_maps(self, trans, a, y):
        if len(self.states) != 1 or len(self.inputs) != 1 or a is None:
            raise WdfTraceError("a per-sample resistance is supported on the diode-clipper topology only "
                                "(clipper_pot.py:94-101)")
        s, x, b = self.states[0][2], self.inputs[0][1], self.root_slot

        def coefs(w):
            c1 = w.c1 if w.c1 is not None else torch.zeros(NB, dtype=torch.float64)
            return [float(w.c0[s]), float(w.c0[x]), float(w.c0[b]), float(c1[s]), float(c1[x]), float(c1[b])]

        # the root symbol is the diode pair's reflected wave b, or the MLP's OUTPUT, which the
        # script negates before sending it down (clipper_pot.py:121; the kernel does the same)
        sb = -1.0 if type(self.root).__name__ == "DenseRootModel" else 1.0
        want_a = [1.0, 0.0, 0.0, -1.0, 1.0, 0.0]              # a  = z - p (z - x)
        want_z = [0.0, 0.0, sb, -1.0, 1.0, 0.0]               # z' = b - p (z - x)
        want_y = [0.5, 0.0, 0.5 * sb, -0.5, 0.5, 0.0]         # y  = (z' + z)/2
        for got, want, name in ((coefs(a), want_a, "root input"), (coefs(trans[0]), want_z, "state update"),
                                (coefs(y), want_y, "output")):
            if any(abs(g - w) > 1e-12 for g, w in zip(got, want)):
                raise WdfTraceError(f"per-sample resistance: the recorded {name} is not the diode-clipper map")

    def _lower_dynamic(self, trans, a, y, lr, z0, dev, engine, mlp_root):
        self._check_clipper_maps(trans, a, y)
        cap = self.states[0][0]
        vs, rkey = self.rin
        x = self._gather_inputs([self.inputs[0][0]], dev)[:, :, 0].contiguous()
        r = self._gather_inputs([rkey], dev)[:, :, 0].contiguous()
        z0v = None if z0 is None else z0[0].contiguous()
        root_kind = type(self.root).__name__
        if root_kind == "DiodePair":
            dp = self.root
            theta = torch.stack([dp.Is.as_subclass(torch.Tensor).float().reshape(()),
                                 dp.nVt.as_subclass(torch.Tensor).float().reshape(()),
                                 torch.tensor(1.0), cap.C.as_subclass(torch.Tensor).float().reshape(())]).to(dev)
            y_tb, zT = engine.clipper_stateful(theta, x, float(cap.FS), r=r, n_up=dp.N_up, n_down=dp.N_down, z0=z0v)
        elif root_kind == "DenseRootModel":
            if not (isinstance(lr, RExpr) and lr.kind == "logrp"):
                raise WdfTraceError("DenseRootModel input must be (wave, log P1.R) (clipper_pot.py:119)")
            dense, hidden, n_tanh = mlp_root.describe(self.root)
            theta2 = torch.stack([torch.tensor(1.0), cap.C.as_subclass(torch.Tensor).float().reshape(())]).to(dev)
            w = mlp_root.flat_weights(dense).float().to(dev)
            y_tb, zT = mlp_root.clipper_mlp(theta2, w, x, r, z0v, float(cap.FS), hidden, n_tanh, float(cap.C))
        else:
            raise WdfTraceError(f"unsupported root {root_kind} with a per-sample resistance")
        return y_tb, zT.reshape(1, -1)

    def _lower_static_mlp(self, trans, a, y, lr, z0, dev, mlp_root):
        """MLP root with static impedances: still the clipper topology (only one the kernels cover)."""
        if len(self.states) != 1 or len(self.inputs) != 1:
            raise WdfTraceError("the MLP root is supported on the diode-clipper topology only")
        cap = self.states[0][0]
        s, xs, b = self.states[0][2], self.inputs[0][1], self.root_slot
        p = a.c0[xs]                                          # a = (1-p) z + p x
        ok = (abs(float(a.c0[s] + p - 1.0)) < 1e-12 and abs(float(trans[0].c0[b] + 1.0)) < 1e-12   # z' = -MLP + ...
              and abs(float(trans[0].c0[s] + p)) < 1e-12 and abs(float(trans[0].c0[xs] - p)) < 1e-12)
        if not ok:
            raise WdfTraceError("MLP root: the recorded step is not the diode-clipper map")
        # recover R from p = Rc/(R+Rc): differentiable through p (and C)
        C = cap.C.as_subclass(torch.Tensor).double().reshape(())
        Rc = 1.0 / (2.0 * C * float(cap.FS))
        R = Rc * (1.0 - p) / p
        dense, hidden, n_tanh = mlp_root.describe(self.root)
        theta2 = torch.stack([R, C]).to(device=dev, dtype=torch.float32)
        w = mlp_root.flat_weights(dense).float().to(dev)
        x = self._gather_inputs([self.inputs[0][0]], dev)[:, :, 0].contiguous()
        z0v = None if z0 is None else z0[0].contiguous()
        y_tb, zT = mlp_root.clipper_mlp(theta2, w, x, None, z0v, float(cap.FS), hidden, n_tanh, float(C), R_static=float(R))
        return y_tb, zT.reshape(1, -1)


# ------------------------------------------------------------------------------ element hooks
def _recorder_for(sample):
    global _current
    parent = sample._wdf_src[0]
    if _current is None or _current.parent is not parent:
        _current = Recorder(parent)
    return _current


def bind_voltage(value):
    """set_voltage(): a sample taken from a sequence tensor becomes this step's input symbol."""
    if getattr(value, "_wdf_src", None) is not None:
        return _recorder_for(value).input_wave(value)
    return value


def bind_resistance(elem, value):
    if getattr(value, "_wdf_src", None) is not None:
        return _recorder_for(value).resistance(elem, value)
    return value


def state(cap):
    """Capacitor.reflected(): b = z -- as this step's state symbol when a loop is being recorded."""
    if _current is not None:
        return _current.state(cap)
    return cap.z


def note_voltage(value):
    if isinstance(value, Wave):
        value.rec.note_output(value)
    return value
