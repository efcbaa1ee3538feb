"""Host side of the MLP-root diode clipper (csrc/wdf_mlp.h): weight flattening, the autograd
Function, and the dense weight-gradient pass.

The per-lane reverse-sweep kernel returns g_b[n] = dL/d b[n] and the network inputs
(a[n], log R[n]); the weight gradient  dL/dW = -sum_n g_b[n] dMLP(a[n], lr[n])/dW  is then a
batched-MLP backward over B*T independent samples: plain GEMMs, run through torch
(hipBLASLt) in chunks.  That is the one place this package uses library GEMMs."""
import torch

from . import binding
from . import compat_tf as tf

_DENSE_CHUNK = 1 << 22        # samples per chunk of the dense pass (activations: chunk x 16 x 4 B x layers)


def describe(model):
    """(dense layers, hidden width, number of tanh layers) of a layers.DenseRootModel; raises
    for anything the kernels do not cover."""
    dense = [l for l in model.layers if type(l).__name__ == "DenseLayer"]
    acts = []
    for i, l in enumerate(model.layers):
        if type(l).__name__ == "DenseLayer":
            nxt = model.layers[i + 1] if i + 1 < len(model.layers) else None
            acts.append("tanh" if nxt is tf.nn.tanh else ("relu" if nxt is tf.nn.relu else ""))
    if len(dense) < 2 or acts[-1] != "" or any(a != "tanh" for a in acts[:-1]):
        raise binding.WdfHipError(f"MLP root: expected tanh hidden layers and a linear output, got {acts}")
    sizes = [int(dense[0].kernel.shape[1])] + [int(d.kernel.shape[2]) for d in dense]
    hidden = sizes[1]
    if sizes[0] != 2 or sizes[-1] != 1 or any(s != hidden for s in sizes[1:-1]):
        raise binding.WdfHipError(f"MLP root: expected 2 -> H -> ... -> H -> 1, got {sizes}")
    return dense, hidden, len(dense) - 1


def flat_weights(dense):
    """kernel[in][out] then bias[out] per layer (the JSON order, layers.py:31-36), differentiable."""
    parts = []
    for d in dense:
        parts += [d.kernel.as_subclass(torch.Tensor)[0].reshape(-1), d.bias.as_subclass(torch.Tensor)[0].reshape(-1)]
    return torch.cat(parts)


def _dense_forward(w, hidden, n_tanh, inp):
    """MLP(inp [N,2]) -> [N] with the flat weight vector w (same layout as the kernel)."""
    o = 0
    h = inp
    n_in = 2
    for _ in range(n_tanh):
        k = w[o:o + n_in * hidden].reshape(n_in, hidden)
        o += n_in * hidden
        b = w[o:o + hidden]
        o += hidden
        h = torch.tanh(h @ k + b)
        n_in = hidden
    k = w[o:o + hidden].reshape(hidden, 1)
    o += hidden
    return (h @ k)[:, 0] + w[o]


class _ClipperMlpFn(torch.autograd.Function):
    """y [T,B] = clipper_mlp(theta2 = {R, C}, w, x [B,T] (, r [B,T]))."""

    @staticmethod
    def forward(ctx, theta2, w, x, r, z0, fs, hidden, n_tanh, want_zT):
        need = theta2.requires_grad or w.requires_grad
        th, wd = theta2.detach().contiguous(), w.detach().contiguous()
        y, zs, zT = binding.clipper_mlp_fwd(x, th, wd, hidden, n_tanh, fs, r=r, want_stash=need, z0=z0,
                                            want_zT=want_zT)
        ctx.cfg = (fs, hidden, n_tanh, r is not None)
        if need:
            ctx.save_for_backward(th, wd, x, zs, *([r] if r is not None else []))
        if want_zT:
            ctx.mark_non_differentiable(zT)
        return y, zT

    @staticmethod
    def backward(ctx, gy, _gzT):
        fs, hidden, n_tanh, has_r = ctx.cfg
        saved = ctx.saved_tensors
        th, wd, x, zs = saved[:4]
        r = saved[4] if has_r else None
        gth, gb, ain, lrin = binding.clipper_mlp_bwd(x, th, wd, hidden, n_tanh, fs, zs, gy.contiguous(), r=r)
        # dense weight-gradient pass: dL/dw = -sum_n gb[n] dMLP(a[n], lr[n])/dw
        a_flat, g_flat = ain.reshape(-1), gb.reshape(-1)
        if lrin is not None:
            lr_flat = lrin.reshape(-1)
        else:
            R, C = th[0].double(), th[1].double()
            lr_const = torch.log(1.0 / (1.0 / R + 2.0 * C * fs)).float()
            lr_flat = None
        gw = torch.zeros_like(wd)
        n = a_flat.numel()
        with torch.enable_grad():
            for s in range(0, n, _DENSE_CHUNK):
                e = min(n, s + _DENSE_CHUNK)
                wv = wd.detach().requires_grad_(True)
                lr_c = lr_flat[s:e] if lr_flat is not None else lr_const.expand(e - s)
                out = _dense_forward(wv, hidden, n_tanh, torch.stack([a_flat[s:e], lr_c], dim=1))
                proxy = -(g_flat[s:e] * out).sum()
                gw += torch.autograd.grad(proxy, wv)[0]
        return gth, gw, None, None, None, None, None, None, None


def run_clipper_mlp(circ, x, z0, return_state):
    """Circuit.__call__ for root = layers.DenseRootModel (clipper_pot.py topology only)."""
    if not circ._is_clipper():
        raise binding.WdfHipError("the MLP root is supported on the diode-clipper topology "
                                  "Parallel(ResistiveVoltageSource, Capacitor) (clipper_pot.py:94-101)")
    model, vs, cap = circ.root, circ.top.P1, circ.top.P2
    dense, hidden, n_tanh = describe(model)
    dev = x.device
    Rv = vs.R if isinstance(vs.R, torch.Tensor) else torch.tensor(float(vs.R))
    theta2 = torch.stack([Rv.as_subclass(torch.Tensor).float().reshape(()),
                          cap.C.as_subclass(torch.Tensor).float().reshape(())]).to(dev)
    w = flat_weights(dense).float().to(dev)
    r = x[:, :, 1].contiguous() if circ.per_sample_R is not None else None
    xv = x[:, :, 0].contiguous()
    z0t = None if z0 is None else z0.as_subclass(torch.Tensor).to(dev).float().reshape(-1).contiguous()
    y, zT = _ClipperMlpFn.apply(theta2, w, xv, r, z0t, float(cap.FS), hidden, n_tanh, bool(return_state))
    y = y.as_subclass(tf.Tensor)
    return (y, zT.reshape(1, -1)) if return_state else y
