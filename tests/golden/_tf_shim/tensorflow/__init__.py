"""Minimal stand-in for the `tensorflow` NAME, used ONLY by tests/golden/gen_golden.py.

TensorFlow 2.5 (requirements.txt:1 of the reference) is not installed in the build
container and cannot be installed (no network).  The reference's WDF library only uses
elementwise ops (+ matmul/tanh for the MLP root), so to execute the reference's own
tf_wdf.py / layers.py / Model / ClipperModel code and record golden vectors, this module
maps the ~40 `tf.*` names those files touch onto torch CPU tensors.  The OP SEQUENCE that
runs is the reference's; the fp kernels and the autodiff engine are torch's (IEEE + - * /
are bit-identical; log/tanh/matmul may differ from TF by a few ulp -- stated in the
golden tolerances).  This is test tooling: it is not shipped, not imported by the product
and never runs on the GPU box.

Set tensorflow._DTYPE = torch.float64 before building a model to record high-precision
goldens (tf.float32 then means "the working dtype").
"""
import numpy as np
import torch

_DTYPE = torch.float32


class _DT:
    def __repr__(self):
        return "tf.float32(shim)"


float32 = _DT()


def _dt(d=None):
    return _DTYPE


def _t(x):
    if isinstance(x, torch.Tensor):
        return x if x.dtype == _DTYPE or not x.is_floating_point() else x.to(_DTYPE)
    if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], torch.Tensor):
        return torch.stack([_t(v) for v in x])
    return torch.as_tensor(np.asarray(x), dtype=_DTYPE) if not np.isscalar(x) else torch.tensor(float(x), dtype=_DTYPE)


class _Logger:
    def setLevel(self, *_):
        pass


def get_logger():
    return _Logger()


class Module:
    def __init__(self, name=None):
        pass

    @property
    def trainable_variables(self):
        """tf.Module order: attributes sorted by name, depth first."""
        out, seen = [], set()

        def visit(obj):
            if id(obj) in seen:
                return
            seen.add(id(obj))
            if isinstance(obj, torch.Tensor):
                if getattr(obj, "_is_variable", False) and obj.requires_grad:
                    out.append(obj)
                return
            if isinstance(obj, Module):
                for k in sorted(vars(obj)):
                    visit(vars(obj)[k])
            elif isinstance(obj, (list, tuple)):
                for v in obj:
                    visit(v)

        visit(self)
        return tuple(out)


def Variable(initial_value=None, name=None, trainable=True, dtype=None, constraint=None):
    v = _t(initial_value).clone().detach()
    v.requires_grad_(bool(trainable))
    v._is_variable = True
    v.constraint = constraint
    return v


def _assign(self, value):
    with torch.no_grad():
        self.copy_(_t(value).reshape(self.shape))
    return self


torch.Tensor.assign = _assign


def constant(x, dtype=None):
    return _t(x)


def zeros(shape, dtype=None):
    return torch.zeros(shape, dtype=_DTYPE)


def zeros_like(x):
    return torch.zeros_like(_t(x))


def ones_like(x):
    return torch.ones_like(_t(x))


def clip_by_value(x, lo, hi):
    return torch.clamp(_t(x), lo, hi)


def sqrt(x):
    return torch.sqrt(_t(x))


def shape(x):
    return torch.tensor(list(x.shape))


def cast(x, dtype=None):
    return _t(x)


def expand_dims(x, axis):
    return _t(x).unsqueeze(axis)


def concat(values, axis):
    return torch.cat([_t(v) for v in values], dim=axis)


def transpose(x, perm=None):
    x = _t(x)
    return x.permute(*perm) if perm is not None else x.permute(*reversed(range(x.dim())))


def matmul(a, b):
    return torch.matmul(_t(a), _t(b))


class math:
    reciprocal = staticmethod(lambda x: torch.reciprocal(_t(x)))
    log = staticmethod(lambda x: torch.log(_t(x)))
    square = staticmethod(lambda x: torch.square(_t(x)))
    abs = staticmethod(lambda x: torch.abs(_t(x)))
    reduce_sum = staticmethod(lambda x: torch.sum(_t(x)))
    reduce_mean = staticmethod(lambda x: torch.mean(_t(x)))
    reduce_min = staticmethod(lambda x: torch.min(_t(x)))
    reduce_max = staticmethod(lambda x: torch.max(_t(x)))


class nn:
    tanh = staticmethod(torch.tanh)
    relu = staticmethod(torch.relu)


class TensorArray:
    def __init__(self, dtype=None, size=0, clear_after_read=True):
        self._items = [None] * size

    def write(self, i, value):
        self._items[i] = value
        return self

    def stack(self):
        return torch.stack(self._items)


class _Orthogonal:
    def __call__(self, shape):
        w = torch.empty(*shape, dtype=_DTYPE)
        torch.nn.init.orthogonal_(w)
        return w


class _Zeros:
    def __call__(self, shape):
        return torch.zeros(shape, dtype=_DTYPE)


class _MSE:
    def __call__(self, y_true, y_pred):
        return torch.mean(torch.square(_t(y_true) - _t(y_pred)))


class keras:
    class initializers:
        Orthogonal = _Orthogonal
        Zeros = _Zeros

    class losses:
        MeanSquaredError = _MSE
