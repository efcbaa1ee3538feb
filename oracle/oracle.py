"""ctypes front end of the CPU oracle (oracle/liboracle.so, oracle/_ref/libtoms917_ref.so).

TEST INFRASTRUCTURE ONLY.  Import this from tests/, from __graft_entry__.smoke() and from
the cpu_baseline leg of bench.py -- never from the product package.  See wdf_oracle.h for
what the oracle restates (reference file:line) and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NODE_RESISTOR, NODE_CAPACITOR, NODE_RES_VSOURCE, NODE_SERIES, NODE_PARALLEL, NODE_INVERTER = 1, 2, 3, 4, 5, 6
ROOT_IDEAL_VSOURCE, ROOT_DIODE_PAIR, ROOT_MLP = 1, 2, 3
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2
MAX_MLP_LAYERS = 8


class OrcNode(C.Structure):
    _fields_ = [("type", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32),
                ("param", C.c_int32), ("vin", C.c_int32), ("rin", C.c_int32)]


class OrcCircuit(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("top", C.c_int32), ("probe", C.c_int32),
                ("n_in", C.c_int32), ("root_kind", C.c_int32), ("root_vin", C.c_int32),
                ("p_is", C.c_int32), ("p_nvt", C.c_int32), ("n_up", C.c_int32),
                ("n_down", C.c_int32), ("mlp_off", C.c_int32), ("mlp_n_layers", C.c_int32),
                ("mlp_sizes", C.c_int32 * (MAX_MLP_LAYERS + 1)),
                ("mlp_act", C.c_int32 * MAX_MLP_LAYERS), ("fs", C.c_double)]


def build(ref=True):
    """(Re)build liboracle.so and -- when /root/reference exists -- oracle/_ref/."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_ref = None


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype)) if arr is not None else None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.oracle_wright_omega_f64.restype = C.c_double
        L.oracle_wright_omega_f64.argtypes = [C.c_double]
        L.oracle_wright_omega_f32.restype = C.c_float
        L.oracle_wright_omega_f32.argtypes = [C.c_float]
        L.oracle_wright_omega_ext_f64.restype = C.c_double
        L.oracle_wright_omega_ext_f64.argtypes = [C.c_double, C.POINTER(C.c_int)]
        L.oracle_wright_omega_ext_f32.restype = C.c_float
        L.oracle_wright_omega_ext_f32.argtypes = [C.c_float, C.POINTER(C.c_int)]
        L.oracle_diode_pair_f64.restype = C.c_double
        L.oracle_diode_pair_f64.argtypes = [C.c_double] * 5 + [C.c_int] * 2
        L.oracle_diode_pair_f32.restype = C.c_float
        L.oracle_diode_pair_f32.argtypes = [C.c_float] * 5 + [C.c_int] * 2
        L.oracle_max_threads.restype = C.c_int
        L.oracle_asym_root_f64.restype = C.c_double
        L.oracle_asym_root_f64.argtypes = [C.c_double] * 6
        _lib = L
    return _lib


def ref_lib():
    """The REAL reference toms917 build (None when oracle/_ref was never built)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libtoms917_ref.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_toms917_omega_real.restype = C.c_double
        L.ref_toms917_omega_real.argtypes = [C.c_double]
        L.ref_toms917_omega_ext.restype = C.c_double
        L.ref_toms917_omega_ext.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ref_toms917_diode_pair.restype = C.c_float
        L.ref_toms917_diode_pair.argtypes = [C.c_float] * 5
        _ref = L
    return _ref


# --------------------------------------------------------------------------- omega / diode
def wright_omega(x, dtype=np.float64):
    x = np.ascontiguousarray(x, dtype=dtype)
    w = np.empty_like(x)
    if dtype == np.float64:
        lib().oracle_wright_omega_vec_f64(_p(x, C.c_double), _p(w, C.c_double), C.c_int64(x.size))
    else:
        lib().oracle_wright_omega_vec_f32(_p(x, C.c_float), _p(w, C.c_float), C.c_int64(x.size))
    return w


def wright_omega_iters(x, dtype=np.float64):
    """omega and the number of FSC iterations the oracle ran, element-wise."""
    x = np.ascontiguousarray(x, dtype=dtype).ravel()
    w = np.empty_like(x)
    it = np.empty(x.size, dtype=np.int32)
    n = C.c_int(0)
    f = lib().oracle_wright_omega_ext_f64 if dtype == np.float64 else lib().oracle_wright_omega_ext_f32
    for i, xi in enumerate(x):
        w[i] = f(xi, C.byref(n))
        it[i] = n.value
    return w, it


def ref_wright_omega(x):
    L = ref_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libtoms917_ref.so not built (make -C oracle ref)")
    x = np.ascontiguousarray(x, dtype=np.float64)
    w = np.empty_like(x)
    L.ref_toms917_omega_vec(_p(x, C.c_double), _p(w, C.c_double), C.c_int64(x.size))
    return w


def diode_pair(a, R, Is, Vt, nabla, n_up=1, n_down=1, dtype=np.float64):
    a = np.atleast_1d(np.asarray(a, dtype=dtype))
    f = lib().oracle_diode_pair_f64 if dtype == np.float64 else lib().oracle_diode_pair_f32
    return np.array([f(ai, R, Is, Vt, nabla, n_up, n_down) for ai in a], dtype=dtype)


# --------------------------------------------------------------------------- tree programs
class Circuit:
    """A tree program + root for the oracle's generic interpreter."""

    def __init__(self, nodes, top, probe, n_in, root_kind, fs, root_vin=-1, p_is=-1, p_nvt=-1,
                 n_up=1, n_down=1, mlp_off=0, mlp_sizes=(), mlp_act=()):
        self.nodes = (OrcNode * len(nodes))(*[OrcNode(*n) for n in nodes])
        c = OrcCircuit()
        c.n_nodes, c.top, c.probe, c.n_in = len(nodes), top, probe, n_in
        c.root_kind, c.root_vin, c.p_is, c.p_nvt = root_kind, root_vin, p_is, p_nvt
        c.n_up, c.n_down, c.mlp_off = n_up, n_down, mlp_off
        c.mlp_n_layers = max(len(mlp_sizes) - 1, 0)
        for i, s in enumerate(mlp_sizes):
            c.mlp_sizes[i] = s
        for i, a in enumerate(mlp_act):
            c.mlp_act[i] = a
        c.fs = float(fs)
        self.c = c


def rc_lowpass_circuit(fs):
    """lpf.py:20-28: I1 = Inverter(Series(R1, C1)), root = IdealVoltageSource, probe = C1.
    theta = [R1, C1]."""
    nodes = [(NODE_RESISTOR, -1, -1, 0, -1, -1), (NODE_CAPACITOR, -1, -1, 1, -1, -1),
             (NODE_SERIES, 0, 1, -1, -1, -1), (NODE_INVERTER, 2, -1, -1, -1, -1)]
    return Circuit(nodes, top=3, probe=1, n_in=1, root_kind=ROOT_IDEAL_VSOURCE, fs=fs, root_vin=0)


def voltage_divider_circuit():
    """voltage_divider.py:17-25: I1 = Inverter(Series(R1, R2)), probe = R1. theta = [R1, R2]."""
    nodes = [(NODE_RESISTOR, -1, -1, 0, -1, -1), (NODE_RESISTOR, -1, -1, 1, -1, -1),
             (NODE_SERIES, 0, 1, -1, -1, -1), (NODE_INVERTER, 2, -1, -1, -1, -1)]
    return Circuit(nodes, top=3, probe=0, n_in=1, root_kind=ROOT_IDEAL_VSOURCE, fs=48000.0, root_vin=0)


def clipper_diode_circuit(fs, n_up=1, n_down=1, per_sample_r=False):
    """clipper_pot.py:94-101 topology P1 = Parallel(Vs, C) with the analytic diode-pair root.
    theta = [Is, nVt, R, C]; channel 0 = Vin, channel 1 = R (if per_sample_r)."""
    nodes = [(NODE_RES_VSOURCE, -1, -1, 2, 0, 1 if per_sample_r else -1),
             (NODE_CAPACITOR, -1, -1, 3, -1, -1), (NODE_PARALLEL, 0, 1, -1, -1, -1)]
    return Circuit(nodes, top=2, probe=1, n_in=2 if per_sample_r else 1, root_kind=ROOT_DIODE_PAIR,
                   fs=fs, p_is=0, p_nvt=1, n_up=n_up, n_down=n_down)


def clipper_mlp_circuit(fs, sizes, acts):
    """clipper_pot.py:94-127: P1 = Parallel(Vs, C), root = -DenseRootModel(b, log R), per-sample R.
    theta = [R (unused, channel 1 overrides), C, mlp weights...]."""
    nodes = [(NODE_RES_VSOURCE, -1, -1, 0, 0, 1), (NODE_CAPACITOR, -1, -1, 1, -1, -1),
             (NODE_PARALLEL, 0, 1, -1, -1, -1)]
    return Circuit(nodes, top=2, probe=1, n_in=2, root_kind=ROOT_MLP, fs=fs, mlp_off=2,
                   mlp_sizes=sizes, mlp_act=acts)


def mlp_theta_from_json(model_json):
    """Flatten a reference weights JSON (layers.py:51-70 loader semantics: skip non-dense
    entries) into (theta_tail, sizes, acts)."""
    sizes = [model_json["in_shape"][-1]]
    acts, flat = [], []
    for layer in model_json["layers"]:
        if layer["type"] != "dense":
            continue
        k = np.asarray(layer["weights"][0], dtype=np.float64)
        b = np.asarray(layer["weights"][1], dtype=np.float64)
        sizes.append(k.shape[1])
        acts.append({"tanh": ACT_TANH, "relu": ACT_RELU}.get(layer["activation"], ACT_NONE))
        flat += [k.ravel(), b.ravel()]
    return np.concatenate(flat), sizes, acts


def tree_fwd(circ, theta, x, dtype=np.float64, z0=None, return_state=False):
    """x [B,T] or [B,T,n_in] -> y [T,B]."""
    x = np.ascontiguousarray(x, dtype=dtype)
    if x.ndim == 2:
        x = x[:, :, None]
    B, T, n_in = x.shape
    assert n_in == circ.c.n_in
    theta = np.ascontiguousarray(theta, dtype=dtype)
    y = np.empty((T, B), dtype=dtype)
    zT = np.zeros((B, circ.c.n_nodes), dtype=dtype)
    z0a = None if z0 is None else np.ascontiguousarray(z0, dtype=dtype)
    ct = C.c_double if dtype == np.float64 else C.c_float
    f = lib().oracle_tree_fwd_f64 if dtype == np.float64 else lib().oracle_tree_fwd_f32
    rc = f(C.byref(circ.c), circ.nodes, _p(theta, ct), _p(x, ct), _p(y, ct), _p(z0a, ct), _p(zT, ct),
           C.c_int64(B), C.c_int64(T))
    if rc:
        raise RuntimeError(f"oracle_tree_fwd rc={rc}")
    return (y, zT) if return_state else y


def tree_dtheta(circ, theta, k, x):
    """Complex-step d y / d theta[k]: returns (y [T,B], dy [T,B])."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim == 2:
        x = x[:, :, None]
    B, T, _ = x.shape
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    y = np.empty((T, B))
    dy = np.empty((T, B))
    rc = lib().oracle_tree_dtheta_c64(C.byref(circ.c), circ.nodes, _p(theta, C.c_double), C.c_int(k),
                                      _p(x, C.c_double), _p(y, C.c_double), _p(dy, C.c_double),
                                      C.c_int64(B), C.c_int64(T))
    if rc:
        raise RuntimeError(f"oracle_tree_dtheta rc={rc}")
    return y, dy


def tree_grad(circ, theta, x, gy, params=None):
    """dL/dtheta[k] = sum gy * dy/dtheta[k] by complex step, for k in params."""
    params = range(len(theta)) if params is None else params
    return np.array([np.sum(gy * tree_dtheta(circ, theta, k, x)[1]) for k in params])


# --------------------------------------------------------------------------- clipper twin
def clipper_fwd(theta4, fs, x, r=None, n_up=1, n_down=1, dtype=np.float64, n_threads=0):
    x = np.ascontiguousarray(x, dtype=dtype)
    B, T = x.shape
    th = np.ascontiguousarray(theta4, dtype=dtype)
    ra = None if r is None else np.ascontiguousarray(r, dtype=dtype)
    y = np.empty((T, B), dtype=dtype)
    ct = C.c_double if dtype == np.float64 else C.c_float
    f = lib().oracle_clipper_fwd_f64 if dtype == np.float64 else lib().oracle_clipper_fwd_f32
    f(_p(th, ct), C.c_double(fs), n_up, n_down, _p(x, ct), _p(ra, ct), _p(y, ct),
      C.c_int64(B), C.c_int64(T), n_threads)
    return y


def clipper_fwd_bwd(theta4, fs, x, gy, r=None, n_up=1, n_down=1, dtype=np.float64, n_threads=0):
    """returns y [T,B], g[4] = dL/d{Is, nVt, R, C} (double)."""
    x = np.ascontiguousarray(x, dtype=dtype)
    B, T = x.shape
    th = np.ascontiguousarray(theta4, dtype=dtype)
    ra = None if r is None else np.ascontiguousarray(r, dtype=dtype)
    gya = np.ascontiguousarray(gy, dtype=dtype)
    assert gya.shape == (T, B)
    y = np.empty((T, B), dtype=dtype)
    g = np.zeros(4, dtype=np.float64)
    ct = C.c_double if dtype == np.float64 else C.c_float
    f = lib().oracle_clipper_fwd_bwd_f64 if dtype == np.float64 else lib().oracle_clipper_fwd_bwd_f32
    f(_p(th, ct), C.c_double(fs), n_up, n_down, _p(x, ct), _p(ra, ct), _p(gya, ct), _p(y, ct),
      _p(g, C.c_double), C.c_int64(B), C.c_int64(T), n_threads)
    return y, g


def clipper_mse_step(theta4, fs, x, target, n_up=1, n_down=1, dtype=np.float32, n_threads=0):
    """One fused fwd + MSE + bwd step; returns (loss, g[4], y)."""
    x = np.ascontiguousarray(x, dtype=dtype)
    B, T = x.shape
    th = np.ascontiguousarray(theta4, dtype=dtype)
    tg = np.ascontiguousarray(target, dtype=dtype)
    assert tg.shape == (T, B)
    y = np.empty((T, B), dtype=dtype)
    g = np.zeros(4, dtype=np.float64)
    loss = C.c_double(0.0)
    ct = C.c_double if dtype == np.float64 else C.c_float
    f = lib().oracle_clipper_mse_step_f64 if dtype == np.float64 else lib().oracle_clipper_mse_step_f32
    f(_p(th, ct), C.c_double(fs), n_up, n_down, _p(x, ct), _p(tg, ct), _p(y, ct), _p(g, C.c_double),
      C.byref(loss), C.c_int64(B), C.c_int64(T), n_threads)
    return loss.value, g, y


def asym_root(a, Rp, Is1, V1, Is2, V2):
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    return np.array([lib().oracle_asym_root_f64(ai, Rp, Is1, V1, Is2, V2) for ai in a])


def clipper_asym_fwd(theta6, fs, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    B, T = x.shape
    th = np.ascontiguousarray(theta6, dtype=np.float64)
    y = np.empty((T, B))
    lib().oracle_clipper_asym_fwd_f64(_p(th, C.c_double), C.c_double(fs), _p(x, C.c_double), _p(y, C.c_double),
                                      C.c_int64(B), C.c_int64(T))
    return y


def max_threads():
    return lib().oracle_max_threads()


# --------------------------------------------------------------------------- losses
def mse_loss(y_true, y_pred):
    """tf.keras.losses.MeanSquaredError (clipper_pot.py:176, lpf.py:78): mean over everything."""
    return float(np.mean((np.asarray(y_true, np.float64) - np.asarray(y_pred, np.float64)) ** 2))


def esr_loss(target_y, predicted_y):
    """clipper_pot.py:148-156 with the identity emphasis; eps = np.finfo(float).eps (:145)."""
    t = np.asarray(target_y, np.float64)
    p = np.asarray(predicted_y, np.float64)
    mse = np.sum((t - p) ** 2)
    energy = np.sum(t ** 2)
    n = t.shape[0] * t.shape[1]
    return float(np.sqrt(mse / (energy + np.finfo(float).eps) / n))
