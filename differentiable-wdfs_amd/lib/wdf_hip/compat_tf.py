"""The `tf` namespace the drop-in tf_wdf module exports (`from tf_wdf import tf`).

The reference scripts take TensorFlow itself from the library module (lpf.py:8-9,
voltage_divider.py:8-9, clipper_pot.py:11-12), so whoever provides `tf_wdf` provides `tf`.
This is the subset of the TF 2.5 eager API those scripts and wdf_py/lib touch (SURVEY 8b:
39 names), mapped onto torch: torch is the plumbing that owns parameters, the loss and the
optimizer; the per-sample WDF recursion itself never runs here -- it is lowered to the HIP
kernels (tf_wdf.Circuit / the loop recorder in wdf_hip.trace).

Semantics kept from TF because the scripts depend on them:
  * tf.Module.trainable_variables: depth-first over attributes sorted by name
    (lpf.py:98-99 relies on [C1.C, R1.R]);
  * tf.Variable(constraint=...) is applied by the optimizer after each update
    (tf_wdf.py:69-75, :99-105 clip R and C);
  * tf.GradientTape().gradient(loss, variables) -> list aligned with `variables`;
  * keras Adam: m/v moments with epsilon OUTSIDE the sqrt bias correction, lr_t = lr *
    sqrt(1-b2^t)/(1-b1^t), epsilon = 1e-7.
"""
import numpy as np
import torch

float32 = torch.float32
float64 = torch.float64
int32 = torch.int32

_DEFAULT = torch.float32


def _dtype(d):
    return _DEFAULT if d is None else d


_PENDING = []      # ParamBlocks with deferred optimizer updates (ParamBlock.defer)
_TORCH_FUNCTION = torch.Tensor.__torch_function__.__func__      # (the default dispatch, without a super() object per call)


class Tensor(torch.Tensor):
    """torch.Tensor with the two TF-isms the scripts use on results: .numpy() on tensors that
    require grad, and .assign() on variables."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # (Variables are instances of this class; one adopted by a ParamBlock may have optimizer updates queued: a value is
        #  about to be read or written, so they go out first)
        if _PENDING and not getattr(getattr(func, "__self__", None), "_wdf_sends_pending", False):
            flush_pending()                                  # (an autograd Function marked so sends them with its own first launch)
        return _TORCH_FUNCTION(cls, func, types, args, kwargs or {})

    def numpy(self):
        return torch.Tensor.numpy(self.detach().cpu().as_subclass(torch.Tensor))

    def assign(self, value):
        with torch.no_grad():
            self.copy_(convert(value, dtype=self.dtype, device=self.device).reshape(self.shape))
        return self

    def __float__(self):
        return float(self.detach().as_subclass(torch.Tensor))

    # Scalar component arithmetic is MEMOISED while a script's loop is being recorded (wdf_hip.trace): a script
    # that calls calc_impedance() every time step (clipper_pot.py:117) recomputes 1/(2 C FS), 1/R, -p1R ...
    # thousands of times from unchanged variables; handing back the SAME result object costs a dictionary
    # look-up instead of a torch dispatch, and lets the recorder's replay memo recognise the coefficient.
    # The memo lives in the Recorder -- one forward pass, one autograd graph, gone when the loop is lowered --
    # and is keyed on operand identity and version.
    def _scalar_memo(self, name, other, fn):
        rec = _trace_current()
        if rec is None:
            return fn()
        d = self.__dict__
        with torch._C.DisableTorchFunctionSubclass():       # plain attribute reads: no subclass dispatch (2 us each)
            n = d.get("_wdf_n")
            if n is None:
                n = d["_wdf_n"] = self.numel()
            if n != 1:
                return fn()
            ver = self._version
            if isinstance(other, (int, float)):
                ko = other
            elif other is None:
                ko = None
            elif isinstance(other, torch.Tensor) and other.numel() == 1:
                ko = (id(other), other._version)
            else:
                return fn()
            key = (id(self), ver, name, ko)
            hit = rec.scalar_memo.get(key)
            # (a result the script has since modified in place -- `v += ...`, `.mul_()` -- is not the operation's value any
            #  more: its version moved, compute afresh.  Read in here too: a Tensor property outside costs a dispatch)
            if hit is not None and hit[0]._version == hit[3]:
                return hit[0]
        out = fn()
        with torch._C.DisableTorchFunctionSubclass():
            rec.scalar_memo[key] = (out, self, other, out._version)           # keeps the operands alive: their ids stay their own
        return out

    def __neg__(self):
        return self._scalar_memo("neg", None, lambda: torch.Tensor.__neg__(self))

    def __mul__(self, other):
        if isinstance(other, (int, float, torch.Tensor)):
            return self._scalar_memo("mul", other, lambda: torch.Tensor.__mul__(self, other))
        return torch.Tensor.__mul__(self, other)

    def __rmul__(self, other):
        if isinstance(other, (int, float)):
            return self._scalar_memo("mul", other, lambda: torch.Tensor.__rmul__(self, other))
        return torch.Tensor.__rmul__(self, other)

    def __truediv__(self, other):
        if isinstance(other, (int, float, torch.Tensor)):
            return self._scalar_memo("div", other, lambda: torch.Tensor.__truediv__(self, other))
        return torch.Tensor.__truediv__(self, other)

    def __rtruediv__(self, other):
        if isinstance(other, (int, float)):
            return self._scalar_memo("rdiv", other, lambda: torch.Tensor.__rtruediv__(self, other))
        return torch.Tensor.__rtruediv__(self, other)

    def __add__(self, other):
        if isinstance(other, (int, float, torch.Tensor)):
            return self._scalar_memo("add", other, lambda: torch.Tensor.__add__(self, other))
        return torch.Tensor.__add__(self, other)

    def __getitem__(self, idx):
        # inside a recorded loop (from its second step on) the slice itself is never looked at: the recorder
        # needs to know WHICH sample this is, not its values -- hand back a stand-in that materialises on demand
        if (type(idx) is tuple and len(idx) >= 2 and type(idx[1]) is int and type(idx[0]) is slice and idx[0] == _FULL):
            rec = _trace_current()
            if rec is not None and rec.parent is self and rec.step >= 0 and rec.sample_shape is not None:
                return _LazySample(self, idx)
        out = super().__getitem__(idx)
        # `input[:, i]` / `input[:, i, 0:1]` (lpf.py:40, clipper_pot.py:114-116): remember which
        # sample of which sequence tensor this is, so the loop recorder (wdf_hip.trace) can
        # recognise the script's time loop.  The result is still the real slice.
        if (isinstance(idx, tuple) and len(idx) >= 2 and isinstance(idx[0], slice) and idx[0] == slice(None)
                and isinstance(idx[1], int) and isinstance(out, torch.Tensor)):
            out._wdf_src = (self, idx[1], tuple(idx[2:]))
        return out

    def __format__(self, spec):
        if self.numel() == 1:
            return format(float(self.detach()), spec)
        return str(self)


_FULL = slice(None)


_trace_mod = None


def _trace_current():
    global _trace_mod
    if _trace_mod is None:
        from . import trace
        _trace_mod = trace
    return _trace_mod._current


class _LazySample:
    """`input[:, i(, c)]` inside a recorded loop: carries where the sample comes from (_wdf_src) and its shape;
    anything else a script asks of it is answered by the real slice, made on first use."""
    __slots__ = ("_wdf_src", "_parent", "_idx", "_real")

    def __init__(self, parent, idx):
        self._wdf_src = (parent, idx[1], tuple(idx[2:]))
        self._parent, self._idx, self._real = parent, idx, None

    def _materialise(self):
        if self._real is None:
            self._real = torch.Tensor.__getitem__(self._parent, self._idx)
            self._real._wdf_src = self._wdf_src
        return self._real

    @property
    def shape(self):
        rec = _trace_current()
        if rec is not None and rec.parent is self._parent and rec.sample_shape is not None:
            return rec.sample_shape
        return self._materialise().shape

    def __getattr__(self, name):
        return getattr(self._materialise(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        args = tuple(a._materialise() if isinstance(a, _LazySample) else a for a in args)
        return func(*args, **(kwargs or {}))


def _wrap(t):
    return t.as_subclass(Tensor) if isinstance(t, torch.Tensor) and not isinstance(t, Tensor) else t


def convert(x, dtype=None, device=None):
    """tf.convert_to_tensor: numpy / python numbers / nested lists of tensors -> Tensor."""
    if isinstance(x, torch.Tensor):
        t = x
        if dtype is not None and t.dtype != dtype and t.is_floating_point():
            t = t.to(dtype)
        if device is not None and t.device != torch.device(device):
            t = t.to(device)
        return _wrap(t)
    if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], torch.Tensor):
        return _wrap(torch.stack([convert(v, dtype, device) for v in x]))
    arr = np.asarray(x)
    if arr.dtype.kind == "f" or dtype is not None:
        t = torch.as_tensor(arr, dtype=dtype if dtype is not None else _DEFAULT)
    else:
        t = torch.as_tensor(arr)
    if device is not None:
        t = t.to(device)
    return _wrap(t)


# ------------------------------------------------------------------------------ variables
class _VariableMeta(type(torch.Tensor)):
    def __instancecheck__(cls, obj):
        return isinstance(obj, torch.Tensor) and getattr(obj, "_is_tf_variable", False)


class Variable(Tensor, metaclass=_VariableMeta):
    """tf.Variable(initial_value, name=, trainable=, dtype=, constraint=)."""

    def __new__(cls, initial_value=None, name=None, trainable=True, dtype=None, constraint=None, **_):
        t = convert(initial_value, dtype=_dtype(dtype)).detach().clone().as_subclass(Tensor)
        t.requires_grad_(bool(trainable))
        t._is_tf_variable = True
        t.trainable = bool(trainable)
        t.constraint = constraint
        t.var_name = name
        return t


def flush_pending():
    for pb in list(_PENDING):
        pb.flush()


class ParamBlock:
    """Scalar float32 Variables of one circuit kept in ONE device-resident vector (tf_wdf.Circuit.to_device): the kernels
    read the component values where they live, tape.gradient hands back device scalars and the optimizer updates the
    block with one kernel launch -- a training step (lpf.py:86-99, clipper_pot.py:245-269) makes no host round trip.
    Each adopted Variable keeps its identity, name, constraint and autograd leaf; only its storage moves: it becomes a
    0-dim view of `block`.  `host` is a lagging pinned mirror for decisions that need numbers on the host (the
    time-parallel plan): refreshed by an asynchronous copy every `refresh_every` peeks, never by a synchronisation."""

    refresh_every = 64

    def __init__(self, values, device):
        vals = [float(v) for v in values]
        self.block = torch.tensor(vals, dtype=torch.float32, device=device)
        self.n = len(vals)
        self._host_vals = list(vals)
        self._pinned = torch.empty(self.n, dtype=torch.float32).pin_memory()
        self._event, self._peeks = None, 0
        self.members = {}                       # index -> adopted Variable
        self.pending = []                       # deferred optimizer updates: (binding.Adam, theta view, gradient)

    def defer(self, opt, theta, grad):
        """Queue one optimizer's update of (a slice of) the block instead of launching it: a script with one Adam per
        component (lpf.py:79-80,93-94) then pays ONE launch for all of them -- sent (flush) before anything reads the values:
        the next step's kernels, a torch operation on an adopted Variable, the host mirror."""
        from . import binding
        if len(self.pending) >= binding.ADAM_MULTI_MAX or any(j[0] is opt for j in self.pending):
            self.flush()
        self.pending.append((opt, theta, grad))
        if self not in _PENDING:
            _PENDING.append(self)

    def take_pending(self):
        """The queued updates, for a caller that sends them with its own launch (the device probe, lowering._LinResident)."""
        jobs, self.pending = self.pending, []
        if self in _PENDING:
            _PENDING.remove(self)
        return jobs

    def flush(self):
        if not self.pending:
            return
        from . import binding
        jobs = self.take_pending()
        with torch.no_grad(), torch._C.DisableTorchFunctionSubclass():
            binding.adam_step_multi(jobs)

    def adopt(self, i, v):
        with torch.no_grad():
            v.data = self.block[i]
        v._wdf_block = (self, int(i))
        self.members[int(i)] = v

    def host_values(self):
        """The block as python floats, at most a few dozen steps old; costs no synchronisation."""
        self.flush()
        if self._event is not None and self._event.query():
            self._host_vals = [float(x) for x in self._pinned.tolist()]
            self._event = None
        self._peeks += 1
        if self._event is None and self._peeks % self.refresh_every == 0:
            self._pinned.copy_(self.block, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        return self._host_vals


def clamp_bounds(constraint):
    """(lo, hi) if `constraint` is an element-wise clamp -- what tf_wdf.py:74,104 give their Variables
    (`lambda z: tf.clip_by_value(z, lo, hi)`) -- found by probing it; None for anything else (the optimizer then
    applies the callable as Keras does)."""
    if constraint is None:
        return -float("inf"), float("inf")
    try:
        f = lambda p: float(convert(constraint(_wrap(torch.tensor(p, dtype=torch.float32)))))  # noqa: E731
        lo, hi = f(-float("inf")), f(float("inf"))
        if not lo <= hi:
            return None
        for p in (-1.0e30, -1.0e3, -1.0, -1.0e-3, -1.0e-9, -1.0e-20, 0.0, 1.0e-20, 1.0e-14, 1.0e-9, 1.0e-3, 1.0, 1.0e3, 1.0e30):
            want = float(torch.clamp(torch.tensor(p, dtype=torch.float32), lo, hi))
            if f(p) != want:
                return None
        return lo, hi
    except Exception:
        return None


class Module:
    """tf.Module: attribute container with TF's variable traversal order."""

    def __init__(self, name=None):
        self._tf_name = name

    def _walk_variables(self):
        out, seen = [], set()

        def visit(obj):
            if id(obj) in seen:
                return
            seen.add(id(obj))
            if isinstance(obj, torch.Tensor):
                if getattr(obj, "_is_tf_variable", False):
                    out.append(obj)
                return
            if isinstance(obj, Module):
                for k in sorted(vars(obj)):
                    visit(vars(obj)[k])
            elif isinstance(obj, (list, tuple)):
                for v in obj:
                    visit(v)
            elif isinstance(obj, dict):
                for k in sorted(obj):
                    visit(obj[k])

        visit(self)
        return out

    @property
    def variables(self):
        return tuple(self._walk_variables())

    @property
    def trainable_variables(self):
        return tuple(v for v in self._walk_variables() if v.trainable)

    @property
    def submodules(self):
        out, seen = [], set()

        def visit(obj):
            if id(obj) in seen:
                return
            seen.add(id(obj))
            if isinstance(obj, Module):
                if obj is not self:
                    out.append(obj)
                for k in sorted(vars(obj)):
                    visit(vars(obj)[k])
            elif isinstance(obj, (list, tuple)):
                for v in obj:
                    visit(v)

        visit(self)
        return tuple(out)


# ------------------------------------------------------------------------------ creation ops
def constant(value, dtype=None, shape=None):
    t = convert(value, dtype=_dtype(dtype) if not isinstance(value, torch.Tensor) else dtype)
    return t if shape is None else _wrap(t.reshape(shape))


def convert_to_tensor(value, dtype=None):
    return convert(value, dtype=dtype)


def zeros(shape, dtype=None):
    return _wrap(torch.zeros(shape, dtype=_dtype(dtype)))


def ones(shape, dtype=None):
    return _wrap(torch.ones(shape, dtype=_dtype(dtype)))


def _like(x, fn):
    if hasattr(x, "__wdf_like__"):
        return x.__wdf_like__(fn.__name__)
    return _wrap(fn(convert(x)))


def zeros_like(x, dtype=None):
    return _like(x, torch.zeros_like)


def ones_like(x, dtype=None):
    return _like(x, torch.ones_like)


def cast(x, dtype=None):
    if hasattr(x, "__wdf_cast__"):
        return x.__wdf_cast__(dtype)
    t = convert(x)
    return _wrap(t.to(_dtype(dtype)))


def shape(x):
    return _wrap(torch.tensor(list(x.shape), dtype=torch.int32))


def expand_dims(x, axis):
    if hasattr(x, "__wdf_expand_dims__"):
        return x.__wdf_expand_dims__(axis)
    return _wrap(convert(x).unsqueeze(axis))


def squeeze(x, axis=None):
    t = convert(x)
    return _wrap(t.squeeze() if axis is None else t.squeeze(axis))


def reshape(x, shape):
    return _wrap(convert(x).reshape(tuple(shape)))


def concat(values, axis):
    if any(hasattr(v, "__wdf_concat__") for v in values):
        first = next(v for v in values if hasattr(v, "__wdf_concat__"))
        return first.__wdf_concat__(values, axis)
    vs = [convert(v) for v in values]
    dev = next((v.device for v in vs if v.is_cuda), vs[0].device)
    return _wrap(torch.cat([v.to(dev) for v in vs], dim=axis))


def stack(values, axis=0):
    vs = [convert(v) for v in values]
    return _wrap(torch.stack(vs, dim=axis))


def transpose(x, perm=None):
    if hasattr(x, "__wdf_transpose__"):
        return x.__wdf_transpose__(perm)
    t = convert(x)
    return _wrap(t.permute(*perm) if perm is not None else t.permute(*reversed(range(t.dim()))))


def matmul(a, b):
    return _wrap(torch.matmul(convert(a), convert(b)))


def clip_by_value(x, lo, hi):
    return _wrap(torch.clamp(convert(x), lo, hi))


def sqrt(x):
    return _wrap(torch.sqrt(convert(x)))


def square(x):
    return _wrap(torch.square(convert(x)))


def abs(x):  # noqa: A001
    return _wrap(torch.abs(convert(x)))


def exp(x):
    return _wrap(torch.exp(convert(x)))


def tanh(x):
    return _wrap(torch.tanh(convert(x)))


def reduce_sum(x, axis=None):
    t = convert(x)
    return _wrap(t.sum() if axis is None else t.sum(dim=axis))


def reduce_mean(x, axis=None):
    t = convert(x)
    return _wrap(t.mean() if axis is None else t.mean(dim=axis))


def reduce_min(x, axis=None):
    t = convert(x)
    return _wrap(t.min() if axis is None else t.amin(dim=axis))


def reduce_max(x, axis=None):
    t = convert(x)
    return _wrap(t.max() if axis is None else t.amax(dim=axis))


def _reciprocal(x):
    if hasattr(x, "__wdf_reciprocal__"):
        return x.__wdf_reciprocal__()
    if isinstance(x, Tensor):
        return x._scalar_memo("recip", None, lambda: _wrap(torch.reciprocal(x)))
    return _wrap(torch.reciprocal(convert(x)))


def _log(x):
    if hasattr(x, "__wdf_log__"):
        return x.__wdf_log__()
    if isinstance(x, Tensor):
        return x._scalar_memo("log", None, lambda: _wrap(torch.log(x)))
    return _wrap(torch.log(convert(x)))


class math:  # noqa: N801  (tf.math)
    reciprocal = staticmethod(_reciprocal)
    log = staticmethod(_log)
    exp = staticmethod(exp)
    square = staticmethod(square)
    sqrt = staticmethod(sqrt)
    abs = staticmethod(abs)
    tanh = staticmethod(tanh)
    reduce_sum = staticmethod(reduce_sum)
    reduce_mean = staticmethod(reduce_mean)
    reduce_min = staticmethod(reduce_min)
    reduce_max = staticmethod(reduce_max)


def _relu(x):
    return _wrap(torch.relu(convert(x)))


class nn:  # noqa: N801  (tf.nn) -- identity matters: clipper_pot.py:307 tests `layer == tf.nn.tanh`
    tanh = staticmethod(tanh)
    relu = staticmethod(_relu)


class _Logger:
    def setLevel(self, *_):  # noqa: N802
        pass


def get_logger():
    return _Logger()


# ------------------------------------------------------------------------------ TensorArray
class TensorArray:
    """tf.TensorArray(dtype, size, clear_after_read): write(i, v) returns the array,
    stack() -> [size, ...].  If the written values are recorded WDF outputs (wdf_hip.trace),
    stack() is where the whole loop is lowered to the HIP kernel."""

    def __init__(self, dtype=None, size=0, dynamic_size=False, clear_after_read=True, **_):
        self._items = [None] * int(size)
        self._dtype = dtype

    def write(self, index, value):
        if index >= len(self._items):
            self._items.extend([None] * (index + 1 - len(self._items)))
        self._items[index] = value
        if hasattr(value, "__wdf_stack__"):          # a recorded wave: this step's output
            value.rec.note_output(value)
        return self

    def read(self, index):
        return self._items[index]

    def size(self):
        return len(self._items)

    def stack(self):
        first = next((v for v in self._items if v is not None), None)
        if first is not None and hasattr(first, "__wdf_stack__"):
            return first.__wdf_stack__(self._items)
        return _wrap(torch.stack([convert(v) for v in self._items]))


# ------------------------------------------------------------------------------ autodiff
class GradientTape:
    """with tf.GradientTape() as tape: ...; tape.gradient(loss, variables).
    torch records the graph eagerly, so the tape only has to run autograd.grad."""

    def __init__(self, persistent=False, watch_accessed_variables=True):
        self._persistent = persistent

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def watch(self, t):
        if isinstance(t, torch.Tensor) and not t.requires_grad:
            t.requires_grad_(True)

    def gradient(self, target, sources):
        single = isinstance(sources, torch.Tensor)
        srcs = [sources] if single else list(sources)
        fused = getattr(target, "_wdf_fused", None)
        with torch._C.DisableTorchFunctionSubclass():           # plain dispatch: no per-call subclass protocol
            where = None if fused is None else [fused[1].get(id(v)) for v in srcs]
            if where is not None and all(i is not None for i in where):
                # the loss of a resident circuit's one-pass step (lowering.Circuit._loss_resident) carries its own gradient:
                # the pass that produced the loss produced d loss / d Variable with it -- nothing to back-propagate
                vec = fused[0]
                grads = []
                for i in where:
                    if isinstance(i, tuple):                    # a tensor Variable living in a flat vector: (offset, count, shape)
                        g = vec[1 + i[0]:1 + i[0] + i[1]].view(i[2]).as_subclass(Tensor)
                    else:
                        g = vec[1 + i].as_subclass(Tensor)
                    g._wdf_gvec = (vec, i)                      # _Adam._apply_resident: the update reads the vector itself
                    grads.append(g)
            else:
                grads = torch.autograd.grad(target, srcs, allow_unused=True, retain_graph=self._persistent)
                grads = [None if g is None else _wrap(g) for g in grads]
        return grads[0] if single else grads


# ------------------------------------------------------------------------------ keras
class _MeanSquaredError:
    def __call__(self, y_true, y_pred):
        a, b = convert(y_true), convert(y_pred)
        dev = a.device if a.is_cuda else b.device
        return _wrap(torch.mean(torch.square(a.to(dev) - b.to(dev, a.dtype))))


class _Adam:
    """tf.keras.optimizers.Adam (TF 2.5 defaults: beta_1 0.9, beta_2 0.999, epsilon 1e-7);
    honours Variable.constraint after the update like Keras does."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **_):
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self._slots = {}
        self._resident = {}     # ids of a group of block-resident Variables -> (binding.Adam, block, index or slice)

    def _apply_resident(self, gv):
        """Variables living in a ParamBlock, gradients on the same device: the whole update -- moments, bias correction,
        step, clip constraints -- is ONE launch of wdf_adam_step on the block (csrc/wdf_optim.h, this class's rule).
        False when the group does not qualify; the caller then takes the per-Variable path."""
        if not gv or any(g is None or not g.is_cuda or getattr(v, "_wdf_block", None) is None or id(v) in self._slots
                         for g, v in gv):
            return False
        key = tuple(id(v) for _, v in gv)
        st = self._resident.get(key)
        if st is None:
            if any(id(v) in k for k in self._resident for _, v in gv):
                return False                                 # the same Variable in another grouping: keep one set of moments
            pb = gv[0][1]._wdf_block[0]
            if any(v._wdf_block[0] is not pb for _, v in gv):
                return False
            bounds = [clamp_bounds(getattr(v, "constraint", None)) for _, v in gv]
            if any(b is None for b in bounds):
                return False
            from . import binding
            idx = [v._wdf_block[1] for _, v in gv]
            opt = binding.Adam(len(idx), self.lr, self.b1, self.b2, self.eps, lo=[b[0] for b in bounds],
                               hi=[b[1] for b in bounds], device=pb.block.device)
            if self.iterations:
                opt.step.fill_(self.iterations)
            sel = slice(idx[0], idx[0] + len(idx)) if idx == list(range(idx[0], idx[0] + len(idx))) else \
                torch.tensor(idx, dtype=torch.int64, device=pb.block.device)
            st = self._resident[key] = (opt, pb, sel, [v for _, v in gv])        # (keeps the Variables' ids their own)
        opt, pb, sel, _ = st
        with torch.no_grad(), torch._C.DisableTorchFunctionSubclass():
            tags = [getattr(g, "_wdf_gvec", None) for g, _ in gv]
            if isinstance(sel, slice) and all(t is not None and t[0] is tags[0][0] for t in tags) and \
                    [t[1] for t in tags] == list(range(sel.start, sel.stop)):
                g = tags[0][0][1 + sel.start:1 + sel.stop]       # the gradients ARE a slice of the one-pass step's output
            else:
                g = torch.stack([g.reshape(()) for g, _ in gv]).to(torch.float32)
            if isinstance(sel, slice):
                pb.defer(opt, pb.block[sel], g)
            else:
                pb.flush()
                th = pb.block[sel]
                opt.apply(th, g)
                pb.block[sel] = th
        return True

    def _apply_flat(self, gv):
        """The weights of a network that lives in one flat device vector (mlp_root.MlpResident), their gradients slices of
        one gradient vector, all of them: the whole update is ONE launch of wdf_adam_step on the vector."""
        if not gv or any(g is None or getattr(v, "_wdf_flat", None) is None or id(v) in self._slots for g, v in gv):
            return False
        res = gv[0][1]._wdf_flat[0]
        tags = [getattr(g, "_wdf_gvec", None) for g, _ in gv]
        if any(v._wdf_flat[0] is not res for _, v in gv) or any(t is None or t[0] is not tags[0][0] for t in tags):
            return False
        spans = [(v._wdf_flat[1], v._wdf_flat[2]) for _, v in gv]
        if any(not isinstance(t[1], tuple) or (t[1][0], t[1][1]) != sp for t, sp in zip(tags, spans)):
            return False
        o = 0
        for off, n in sorted(spans):                            # together they cover the vector, whatever the order
            if off != o:                                        # (tf.Module lists bias before kernel)
                return False
            o += n
        if o != res.w.numel():
            return False
        st = self._resident.get(("flat", id(res)))
        if st is None:
            from . import binding
            opt = binding.Adam(o, self.lr, self.b1, self.b2, self.eps, device=res.w.device)
            if self.iterations:
                opt.step.fill_(self.iterations)
            st = self._resident[("flat", id(res))] = (opt, res)
        with torch.no_grad(), torch._C.DisableTorchFunctionSubclass():
            st[0].apply(res.w, tags[0][0][1:1 + o])
        return True

    def apply_gradients(self, grads_and_vars):
        grads_and_vars = list(grads_and_vars)
        with torch._C.DisableTorchFunctionSubclass():       # (looking at .is_cuda etc. must not send the queued updates)
            done = self._apply_flat(grads_and_vars) or self._apply_resident(grads_and_vars)
        if done:
            self.iterations += 1
            return
        self.iterations += 1
        t = self.iterations
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is None:
                    continue
                g = convert(g).to(v.device, v.dtype).reshape(v.shape).as_subclass(torch.Tensor)
                key = id(v)
                if key not in self._slots:
                    self._slots[key] = (torch.zeros_like(g), torch.zeros_like(g))
                m, s = self._slots[key]
                m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
                s.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
                vt = v.as_subclass(torch.Tensor)
                vt.sub_(lr_t * m / (torch.sqrt(s) + self.eps))
                c = getattr(v, "constraint", None)
                if c is not None:
                    vt.copy_(convert(c(v)).as_subclass(torch.Tensor))


class _SGD:
    def __init__(self, learning_rate=0.01, **_):
        self.lr = float(learning_rate)

    def apply_gradients(self, grads_and_vars):
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is None:
                    continue
                vt = v.as_subclass(torch.Tensor)
                vt.sub_(self.lr * convert(g).to(v.device, v.dtype).reshape(v.shape).as_subclass(torch.Tensor))
                c = getattr(v, "constraint", None)
                if c is not None:
                    vt.copy_(convert(c(v)).as_subclass(torch.Tensor))


class _Orthogonal:
    def __init__(self, gain=1.0, seed=None):
        self.gain = gain

    def __call__(self, shape, dtype=None):
        w = torch.empty(*shape, dtype=_dtype(dtype))
        torch.nn.init.orthogonal_(w, gain=self.gain)
        return _wrap(w)


class _Zeros:
    def __call__(self, shape, dtype=None):
        return zeros(shape, dtype)


class keras:  # noqa: N801
    class losses:  # noqa: N801
        MeanSquaredError = _MeanSquaredError

    class optimizers:  # noqa: N801
        Adam = _Adam
        SGD = _SGD

    class initializers:  # noqa: N801
        Orthogonal = _Orthogonal
        Zeros = _Zeros
