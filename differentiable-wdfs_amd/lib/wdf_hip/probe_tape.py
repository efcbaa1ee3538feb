"""The probed step of a tf_wdf tree as a straight-line scalar program ("tape") the DEVICE can run.

lowering.Circuit.matrices() probes one time step of the element tree with unit vectors on the host (float64 torch scalars
with autograd): fine for a call now and then, but a training loop whose component values live on the device
(Circuit.to_device) would need them back on the host every step.  Here the same probe -- the elements' own
calc_impedance / reflected / incident code (tf_wdf.py:31-214) -- runs ONCE on tracing scalars and leaves a tape of
+, -, *, /, negate, reciprocal over the component values; csrc/wdf_ss_step.h (ss_probe_kernel) evaluates it on the
device in float64 with forward-mode tangents: the step's coefficients AND their Jacobian w.r.t. every component value,
which is all the chain rule from dLoss/d(coefficients) to dLoss/d{R, C} needs (what tape.gradient through
calc_impedance is in the reference, lpf.py:38,87-90).
"""
import numpy as np

OP_CONST, OP_PARAM, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_RECIP = range(8)
MAX_OPS, MAX_PARAMS = 384, 15


class Tape:
    def __init__(self):
        self.ops = []              # (op, a, b): node index operands; CONST: a = index into consts; PARAM: a = parameter
        self.consts = []
        self._const_node = {}
        self._param_node = {}
        self._seen = {}

    def const(self, v):
        v = float(v)
        n = self._const_node.get(v)
        if n is None:
            self.consts.append(v)
            self.ops.append((OP_CONST, len(self.consts) - 1, 0))
            n = self._const_node[v] = len(self.ops) - 1
        return PVal(self, n)

    def param(self, p):
        n = self._param_node.get(p)
        if n is None:
            self.ops.append((OP_PARAM, int(p), 0))
            n = self._param_node[p] = len(self.ops) - 1
        return PVal(self, n)

    def _const_of(self, n):
        op, a, _ = self.ops[n]
        return self.consts[a] if op == OP_CONST else None

    def emit(self, op, a, b=0):
        """A new node -- or an existing one: the probe runs on unit vectors, so most of its arithmetic is with 0 and 1.
        Constants fold, x + 0 / x - 0 / x * 1 / x / 1 return x, x * 0 is 0, and an operation seen before is not repeated
        (the device evaluates the tape one dependent operation after the other: every node saved is ~100 ns of the probe)."""
        a, b = int(a), int(b)
        ca = self._const_of(a)
        cb = self._const_of(b) if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV) else None
        if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV):
            if ca is not None and cb is not None and not (op == OP_DIV and cb == 0.0):
                return self.const({OP_ADD: ca + cb, OP_SUB: ca - cb, OP_MUL: ca * cb}[op] if op != OP_DIV else ca / cb)
            if op == OP_ADD and ca == 0.0:
                return PVal(self, b)
            if op in (OP_ADD, OP_SUB) and cb == 0.0:
                return PVal(self, a)
            if op == OP_SUB and ca == 0.0:
                return self.emit(OP_NEG, b)
            if op == OP_MUL and (ca == 0.0 or cb == 0.0):
                return self.const(0.0)
            if op == OP_MUL and ca == 1.0:
                return PVal(self, b)
            if op in (OP_MUL, OP_DIV) and cb == 1.0:
                return PVal(self, a)
            if op == OP_DIV and ca == 0.0:
                return self.const(0.0)
            if op in (OP_ADD, OP_MUL) and a > b:
                a, b = b, a                                       # (commutative: one key for both orders)
        elif op == OP_NEG:
            if ca is not None:
                return self.const(-ca)
            if self.ops[a][0] == OP_NEG:
                return PVal(self, self.ops[a][1])
        elif op == OP_RECIP and ca is not None and ca != 0.0:
            return self.const(1.0 / ca)
        key = (op, a, b)
        n = self._seen.get(key)
        if n is None:
            self.ops.append(key)
            n = self._seen[key] = len(self.ops) - 1
        return PVal(self, n)

    def pruned(self, outputs):
        """(Tape, outputs) with only the nodes the outputs depend on, renumbered."""
        keep = set()
        stack = [int(o) for o in outputs]
        while stack:
            n = stack.pop()
            if n in keep:
                continue
            keep.add(n)
            op, a, b = self.ops[n]
            if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV):
                stack += [a, b]
            elif op in (OP_NEG, OP_RECIP):
                stack.append(a)
        t = Tape()
        new = {}
        for n in sorted(keep):
            op, a, b = self.ops[n]
            if op == OP_CONST:
                new[n] = t.const(self.consts[a]).n
            elif op == OP_PARAM:
                new[n] = t.param(a).n
            else:
                t.ops.append((op, new[a], new[b] if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV) else 0))
                new[n] = len(t.ops) - 1
        return t, [new[int(o)] for o in outputs]

    def evaluate(self, params, outputs):
        """Host reference of the device kernel: float64 values of the output nodes and their Jacobian [len(outputs), P]."""
        P = len(params)
        val = np.zeros(len(self.ops))
        tan = np.zeros((len(self.ops), P))
        for i, (op, a, b) in enumerate(self.ops):
            if op == OP_CONST:
                val[i] = self.consts[a]
            elif op == OP_PARAM:
                val[i] = params[a]
                tan[i, a] = 1.0
            elif op == OP_ADD:
                val[i], tan[i] = val[a] + val[b], tan[a] + tan[b]
            elif op == OP_SUB:
                val[i], tan[i] = val[a] - val[b], tan[a] - tan[b]
            elif op == OP_MUL:
                val[i], tan[i] = val[a] * val[b], tan[a] * val[b] + val[a] * tan[b]
            elif op == OP_DIV:
                q = val[a] / val[b]
                val[i], tan[i] = q, (tan[a] - q * tan[b]) / val[b]
            elif op == OP_NEG:
                val[i], tan[i] = -val[a], -tan[a]
            elif op == OP_RECIP:
                r = 1.0 / val[a]
                val[i], tan[i] = r, -r * r * tan[a]
        out = np.asarray(outputs, dtype=np.int64)
        return val[out], tan[out]

    def evaluate_torch(self, params, outputs):
        """The tape on torch values, with autograd: params are tensors -- 0-dim (a component value) or of any common shape
        (a per-sample resistance channel [B,T]: every node that depends on it takes that shape, the others stay scalars).
        -> the output nodes' tensors, in order.  What wdf_ss_dyn_* rows are made of (lowering.Circuit._run_dyn)."""
        import torch
        ref = next((p for p in params if isinstance(p, torch.Tensor)), None)
        kw = {} if ref is None else {"dtype": ref.dtype, "device": ref.device}
        val = [None] * len(self.ops)
        for i, (op, a, b) in enumerate(self.ops):
            if op == OP_CONST:
                val[i] = torch.tensor(self.consts[a], **kw)
            elif op == OP_PARAM:
                val[i] = params[a]
            elif op == OP_ADD:
                val[i] = val[a] + val[b]
            elif op == OP_SUB:
                val[i] = val[a] - val[b]
            elif op == OP_MUL:
                val[i] = val[a] * val[b]
            elif op == OP_DIV:
                val[i] = val[a] / val[b]
            elif op == OP_NEG:
                val[i] = -val[a]
            elif op == OP_RECIP:
                val[i] = torch.reciprocal(val[a])
        return [val[int(o)] for o in outputs]

    def packed(self):
        """(int32 [n_ops, 3], float64 [n_consts]) for the device."""
        return np.asarray(self.ops, dtype=np.int32).reshape(-1, 3), np.asarray(self.consts, dtype=np.float64)


class PVal:
    """A scalar of the probe: a node of the tape.  Supports exactly the arithmetic tf_wdf's elements use."""
    __slots__ = ("tape", "n")
    __array_priority__ = 1000

    def __init__(self, tape, n):
        self.tape, self.n = tape, n

    def _c(self, other):
        if isinstance(other, PVal):
            return other
        if hasattr(other, "numel"):
            if other.numel() != 1:
                raise TypeError("the probe works on scalars")
            other = float(other)
        return self.tape.const(other)

    def __add__(self, o): return self.tape.emit(OP_ADD, self.n, self._c(o).n)          # noqa: E704
    def __radd__(self, o): return self.tape.emit(OP_ADD, self._c(o).n, self.n)         # noqa: E704
    def __sub__(self, o): return self.tape.emit(OP_SUB, self.n, self._c(o).n)          # noqa: E704
    def __rsub__(self, o): return self.tape.emit(OP_SUB, self._c(o).n, self.n)         # noqa: E704
    def __mul__(self, o): return self.tape.emit(OP_MUL, self.n, self._c(o).n)          # noqa: E704
    def __rmul__(self, o): return self.tape.emit(OP_MUL, self._c(o).n, self.n)         # noqa: E704
    def __truediv__(self, o): return self.tape.emit(OP_DIV, self.n, self._c(o).n)      # noqa: E704
    def __rtruediv__(self, o): return self.tape.emit(OP_DIV, self._c(o).n, self.n)     # noqa: E704
    def __neg__(self): return self.tape.emit(OP_NEG, self.n)                           # noqa: E704

    # torch.Tensor <op> PVal lands here (an element's initial waves are one-element tensors)
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", "")
        a = args
        me = lambda v: isinstance(v, PVal)  # noqa: E731
        if name in ("mul", "__mul__", "multiply", "__rmul__") and len(a) == 2:
            return a[1] * a[0] if me(a[1]) else a[0] * a[1]
        if name in ("add", "__add__", "__radd__") and len(a) == 2:
            return a[1] + a[0] if me(a[1]) else a[0] + a[1]
        if name in ("sub", "__sub__", "subtract") and len(a) == 2:
            return a[1].__rsub__(a[0]) if me(a[1]) else a[0] - a[1]
        if name in ("rsub", "__rsub__") and len(a) == 2:
            return a[0].__rsub__(a[1]) if me(a[0]) else a[1] - a[0]
        if name in ("neg", "negative"):
            return -a[0]
        if name in ("div", "true_divide", "__truediv__") and len(a) == 2:
            return a[0] / a[1] if me(a[0]) else a[1].__rtruediv__(a[0])
        if name in ("__rtruediv__",) and len(a) == 2:
            return a[0].__rtruediv__(a[1]) if me(a[0]) else a[1] / a[0]
        if name == "reciprocal":
            return a[0].__wdf_reciprocal__()
        raise TypeError(f"torch.{name} on a probe value is not a WDF element operation")

    # compat_tf hooks (tf.math.reciprocal, tf.zeros_like / tf.ones_like)
    def __wdf_reciprocal__(self):
        return self.tape.emit(OP_RECIP, self.n)

    def __wdf_like__(self, fn_name):
        return self.tape.const(0.0 if fn_name == "zeros_like" else 1.0)


def record(circ, param_vars, device_limits=True):
    """Trace one probed step of lowering.Circuit `circ`.  param_vars: the component Variables (each an attribute `R` or `C`
    of an element) in parameter order.  -> (Tape, output node of every entry of the coefficient vector, in the layout of
    Circuit.matrices(): A, Bx, E, ca, da, cy, dy, fy).  Linear trees (ideal-source root folded in) and diode-pair roots."""
    from . import lowering, trace
    tape = Tape()
    ns, ni = circ.ns, circ.ni
    K = ns + ni + 1
    elements = circ.elements + [circ.root]
    saved = lowering._Saved(elements)
    pidx = {id(v): i for i, v in enumerate(param_vars)}
    swapped = []
    rec, trace._current = trace._current, None
    try:
        for e in elements:                                       # component values -> parameter nodes
            for name in ("R", "C"):
                v = e.__dict__.get(name)
                if v is not None and id(v) in pidx:
                    swapped.append((e, name, v))
                    e.__dict__[name] = tape.param(pidx[id(v)])
        circ.top.calc_impedance()
        cols = []
        for k in range(K):                                       # one pass per unit vector (the host probe does all K at once)
            for s, cap in enumerate(circ.caps):
                cap.z = tape.const(1.0 if s == k else 0.0)
            for i, src in enumerate(circ.sources):
                src.Vs = tape.const(1.0 if ns + i == k else 0.0)
            up = circ.top.reflected()
            circ.top.incident(tape.const(1.0 if k == K - 1 else 0.0))
            znew = [cap.z for cap in circ.caps]
            yv = (circ.probe.a + circ.probe.b) * 0.5
            cols.append((tape._c_of(up), [tape._c_of(z) for z in znew], tape._c_of(yv)))
        r_port = tape._c_of(circ.top.R)
    finally:
        for e, name, v in swapped:
            e.__dict__[name] = v
        saved.restore()
        trace._current = rec
    # column k of [up | znew | y] = response to unit k of (z_0.., x_0.., b)
    up = [c[0] for c in cols]
    Z = [[cols[k][1][s] for k in range(K)] for s in range(ns)]
    yv = [c[2] for c in cols]
    A = [[Z[s][j] for j in range(ns)] for s in range(ns)]
    Bx = [[Z[s][ns + i] for i in range(ni)] for s in range(ns)]
    E = [Z[s][K - 1] for s in range(ns)]
    ca, da = up[:ns], up[ns:ns + ni]
    cy, dy, fy = yv[:ns], yv[ns:ns + ni], yv[K - 1]
    zero = tape.const(0.0)
    if circ.root_kind == "IdealVoltageSource":
        # b = -a + 2 Vs (tf_wdf.py:26-28) is linear: fold it in (Vs is the LAST channel) -- lowering.Circuit.matrices
        er = [tape.const(1.0 if i == ni - 1 else 0.0) for i in range(ni)]
        A = [[A[s][j] - E[s] * ca[j] for j in range(ns)] for s in range(ns)]
        Bx = [[Bx[s][i] - E[s] * da[i] + (E[s] * er[i]) * 2.0 for i in range(ni)] for s in range(ns)]
        cy = [cy[s] - fy * ca[s] for s in range(ns)]
        dy = [dy[i] - fy * da[i] + (fy * er[i]) * 2.0 for i in range(ni)]
        E, ca, da, fy = [zero] * ns, [zero] * ns, [zero] * ni, zero
    flat = [v for row in A for v in row] + [v for row in Bx for v in row] + E + ca + da + cy + dy + [fy]
    n_recorded = len(tape.ops)
    tape, nodes = tape.pruned([v.n for v in flat] + [r_port.n])
    tape.n_recorded = n_recorded
    flat_n, rport_n = nodes[:-1], nodes[-1]
    if device_limits and (len(tape.ops) > MAX_OPS or len(param_vars) > MAX_PARAMS):
        from . import binding
        raise binding.WdfHipError(f"the probed step needs {len(tape.ops)} operations on {len(param_vars)} component values; "
                                  f"the device probe holds {MAX_OPS} on {MAX_PARAMS}")
    return tape, flat_n, rport_n


def _c_of(self, v):
    """A probe value as a node (python numbers / 1-element tensors the elements may hand back become constants)."""
    if isinstance(v, PVal):
        return v
    return self.const(float(v))


Tape._c_of = _c_of
