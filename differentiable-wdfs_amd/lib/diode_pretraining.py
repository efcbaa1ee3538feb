"""Pre-training of the tanh-MLP diode root on a synthetic table: the first stage of the
reference's two-stage workflow (wdf_py/diode_clipper/diode_pretraining.py), on the MI355X.

Same steps as the reference script, as functions:
  diode_pair_func(x, R, diode)      eqn (45) reflected wave (diode_pretraining.py:39-60); arrays
                                    go through the HIP kernel wdf_diode_pair_f32, one lane per point
  synthetic_table(diode)            20 impedances 10^1..10^9 x 1000 incident waves in [-2.5, 2.5]
                                    (:64-74); inputs (a, log R) (:104), target -b (:100-101)
  build_model(n_layers, size)       2 -> size x (n_layers+1) tanh -> 1, orthogonal kernels (:113-126),
                                    as a layers.DenseRootModel so it drops into clipper_pot's circuit
  esr_loss / my_loss                (:136-155): MSE + ESR with the script's N = 1000 normaliser
  fit(model, x, y, epochs, ...)     Adam(2e-5), mini-batches of 32, shuffled each epoch
                                    (Keras fit defaults, :159-160)
  pretrain(diode, ...)              the script end to end; save with model_utils.save_model (:192)

The network is evaluated by wdf_mlp_eval and differentiated by wdf_clipper_mlp_wgrad (all
points of a batch in parallel); weights and Adam state stay on the device for the whole fit.
There is no CPU path.
"""
import numpy as np
import torch

from layers import DenseLayer, DenseRootModel
from wdf_hip import binding, mlp_root
from wdf_hip import compat_tf as tf

N = 1000                                   # points per impedance (diode_pretraining.py:64)
eps = float(np.finfo(np.float32).eps)      # :134


def _dev():
    binding.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def diode_pair_func(x, R, diode):
    """Reflected wave of the diode pair for incident wave(s) x at port impedance(s) R."""
    a = torch.as_tensor(np.atleast_1d(np.asarray(x, dtype=np.float32)), device=_dev())
    Rp = torch.as_tensor(np.broadcast_to(np.asarray(R, dtype=np.float32), a.shape).copy(), device=a.device)
    b = binding.diode_pair(a, Rp, float(diode.Is), float(diode.Vt * diode.nabla), int(diode.N_up), int(diode.N_down))
    out = b.cpu().numpy()
    return np.float32(out[0]) if np.isscalar(x) or np.ndim(x) == 0 else out


def synthetic_table(diode, n_points=N, R_orders=None):
    """-> test_x float32 [n_R * n_points, 2] = (a, log R), ideal_y float32 [n_R * n_points] = -b."""
    R_orders = np.linspace(1, 9, 20) if R_orders is None else np.asarray(R_orders, dtype=np.float64)
    a = np.tile(np.linspace(-2.5, 2.5, n_points), len(R_orders))
    R = np.repeat(10.0 ** R_orders, n_points)
    ideal_y = -diode_pair_func(a, R, diode)               # "multiply by -1 to make the data line up better"
    test_x = np.stack([a, np.log(R)], axis=1).astype(np.float32)
    return test_x, ideal_y.astype(np.float32)


def build_model(n_layers, layer_size, seed=None):
    """n_layers + 1 tanh layers of layer_size and a linear output, orthogonal kernels, zero bias."""
    if seed is not None:
        torch.manual_seed(int(seed))
    sizes = [2] + [int(layer_size)] * (n_layers + 1) + [1]
    layers = []
    for i in range(len(sizes) - 1):
        d = DenseLayer(sizes[i], sizes[i + 1])
        layers.append({"type": "dense", "shape": (None, sizes[i + 1]),
                       "weights": [d.kernel.numpy()[0], d.bias.numpy()[0]],
                       "activation": "tanh" if i < len(sizes) - 2 else ""})
    return DenseRootModel({"in_shape": (None, 2), "layers": layers})


class _MlpTableFn(torch.autograd.Function):
    """out [S] = MLP_w(a [S], lr [S]); backward gives dL/dw only (the table is data)."""

    @staticmethod
    def forward(ctx, w, a, lr, hidden, n_tanh):
        wd = w.detach().contiguous()
        ctx.save_for_backward(wd, a, lr)
        ctx.cfg = (hidden, n_tanh)
        return binding.mlp_eval(a, lr, wd, hidden, n_tanh)

    @staticmethod
    def backward(ctx, gout):
        wd, a, lr = ctx.saved_tensors
        hidden, n_tanh = ctx.cfg
        # wgrad returns -sum gb dMLP/dw (gb is dL/d(-MLP) on the circuit path): here gb = -dL/dout
        gw = binding.clipper_mlp_wgrad(a, lr, (-gout).contiguous(), None, wd, hidden, n_tanh, 1.0)
        return gw, None, None, None, None


def model_apply(model, x, w=None):
    """diode_model(test_x): x [S,2] = (a, log R) -> [S] on the device."""
    dense, hidden, n_tanh = mlp_root.describe(model)
    dev = _dev()
    xt = torch.as_tensor(np.asarray(x, dtype=np.float32), device=dev) if not isinstance(x, torch.Tensor) else x
    if w is None:
        w = mlp_root.flat_weights(dense).float().to(dev)
    return _MlpTableFn.apply(w, xt[:, 0].contiguous(), xt[:, 1].contiguous(), hidden, n_tanh)


def mse_loss(target_y, predicted_y):
    return torch.mean(torch.square(target_y - predicted_y))


def esr_loss(target_y, predicted_y, emphasis_func=lambda v: v):
    t, p = emphasis_func(target_y), emphasis_func(predicted_y)
    mse = torch.sum(torch.square(t - p))
    energy = torch.sum(torch.square(t))
    return torch.sqrt(mse / (energy + eps) / N)


def my_loss(target_y, predicted_y):
    return mse_loss(target_y, predicted_y) + esr_loss(target_y, predicted_y)


def _write_back(model, w):
    dense, _, _ = mlp_root.describe(model)
    o = 0
    wc = w.detach().cpu()
    for d in dense:
        n_in, n_out = int(d.kernel.shape[1]), int(d.kernel.shape[2])
        d.kernel.assign(wc[o:o + n_in * n_out].reshape(1, n_in, n_out).numpy())
        o += n_in * n_out
        d.bias.assign(wc[o:o + n_out].reshape(1, n_out).numpy())
        o += n_out


def fit(model, x, y, epochs, learning_rate=2e-5, batch_size=32, shuffle=True, seed=0, loss=my_loss, log=None,
        fused=True):
    """Adam on mini-batches (Keras fit semantics: reshuffle every epoch, last batch may be short).
    Returns the per-epoch mean batch loss.  The model's Variables are updated at the end.

    fused (and loss is my_loss, batch_size <= 64): each epoch is ONE launch of
    wdf_mlp_fit_epoch, the whole mini-batch loop inside a workgroup.  Otherwise one
    eval / weight-gradient / Adam launch sequence per mini-batch with any torch loss."""
    dense, hidden, n_tanh = mlp_root.describe(model)
    dev = _dev()
    xt = torch.as_tensor(np.asarray(x, dtype=np.float32), device=dev)
    yt = torch.as_tensor(np.asarray(y, dtype=np.float32), device=dev)
    S = xt.shape[0]
    w = mlp_root.flat_weights(dense).detach().float().to(dev).requires_grad_(True)
    adam = binding.Adam(w.numel(), learning_rate, device=dev)      # tf.keras.optimizers.Adam defaults (TF 2.5)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    history = []
    if fused and loss is my_loss and batch_size <= 64:
        w = w.detach()
        nb = -(-S // batch_size)
        sums = torch.zeros(int(epochs), dtype=torch.float64, device=dev)
        for epoch in range(int(epochs)):
            perm = torch.randperm(S, device=dev, generator=gen) if shuffle else torch.arange(S, device=dev)
            xa, xl, ys = xt[perm, 0].contiguous(), xt[perm, 1].contiguous(), yt[perm].contiguous()
            binding.mlp_fit_epoch(xa, xl, ys, batch_size, w, adam, hidden, n_tanh, N, eps, sums[epoch:epoch + 1])
            if log is not None:
                log(epoch, float(sums[epoch]) / nb)
        history = [float(v) / nb for v in sums.cpu()]
        _write_back(model, w)
        return history
    for epoch in range(int(epochs)):
        perm = torch.randperm(S, device=dev, generator=gen) if shuffle else torch.arange(S, device=dev)
        xa, xl, ys = xt[perm, 0].contiguous(), xt[perm, 1].contiguous(), yt[perm].contiguous()
        tot = torch.zeros((), device=dev)
        nb = 0
        for s in range(0, S, batch_size):
            e = min(S, s + batch_size)
            out = _MlpTableFn.apply(w, xa[s:e], xl[s:e], hidden, n_tanh)
            l = loss(ys[s:e], out)
            (g,) = torch.autograd.grad(l, [w])
            with torch.no_grad():
                adam.apply(w, g.contiguous())
            tot += l.detach()
            nb += 1
        history.append(float(tot) / nb)
        if log is not None:
            log(epoch, history[-1])
    _write_back(model, w)
    return history


def pretrain(diode, n_layers=2, layer_size=16, epochs=2000, learning_rate=2e-5, batch_size=32, seed=0, log=None,
             fused=True):
    """The reference script end to end: table, model, fit; returns (model, stats)."""
    test_x, ideal_y = synthetic_table(diode)
    model = build_model(n_layers, layer_size, seed=seed)
    dev = _dev()
    yt = torch.as_tensor(ideal_y, device=dev)
    before = (float(mse_loss(yt, model_apply(model, test_x))), float(esr_loss(yt, model_apply(model, test_x))))
    hist = fit(model, test_x, ideal_y, epochs, learning_rate, batch_size, seed=seed, log=log, fused=fused)
    after = (float(mse_loss(yt, model_apply(model, test_x))), float(esr_loss(yt, model_apply(model, test_x))))
    return model, {"name": f"{diode.name}_{n_layers}x{layer_size}_pretrained", "before": before, "after": after,
                   "history": hist}
