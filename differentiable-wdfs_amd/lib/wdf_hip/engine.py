"""Autograd glue: one torch.autograd.Function per fused sequence loop.

The forward call replaces the reference's whole per-sample Python loop
(clipper_pot.py:103-127) by one HIP launch; backward replaces tape.gradient over that loop
(clipper_pot.py:246-268) by one reverse-sweep launch.  torch only carries tensors between
the two and into the loss / optimizer.
"""
import torch

from . import binding


class _ClipperFn(torch.autograd.Function):
    """y[T,B] = clipper(theta = {Is, nVt, R, C}, x[B,T] (, r[B,T]))."""

    @staticmethod
    def forward(ctx, theta, x, r, fs, n_up, n_down, time_major):
        need_grad = theta.requires_grad
        th = theta.detach().contiguous()
        y, zs, _ = binding.clipper_fwd(x, th, fs, r=r, n_up=n_up, n_down=n_down, want_stash=need_grad,
                                       time_major=time_major)
        ctx.cfg = (fs, n_up, n_down, time_major)
        ctx.has_r = r is not None
        if need_grad:
            ctx.save_for_backward(th, x, zs, *([r] if r is not None else []))
        return y

    @staticmethod
    def backward(ctx, gy):
        fs, n_up, n_down, time_major = ctx.cfg
        saved = ctx.saved_tensors
        th, x, zs = saved[0], saved[1], saved[2]
        r = saved[3] if ctx.has_r else None
        gtheta, _ = binding.clipper_bwd(x, th, fs, zs, gy.contiguous(), r=r, n_up=n_up, n_down=n_down,
                                        time_major=time_major)
        return gtheta, None, None, None, None, None, None


def clipper(theta, x, fs, r=None, n_up=1, n_down=1, time_major=False):
    """Diode-clipper sequence loop on the GPU.  theta: float32[4] = {Is, nVt, R, C} on the
    device (may require grad); x, r: [B,T] (or [T,B] when time_major).  Returns y [T,B]."""
    return _ClipperFn.apply(theta, x, r, float(fs), int(n_up), int(n_down), bool(time_major))


class _ClipperStatefulFn(torch.autograd.Function):
    """Same loop with an explicit initial capacitor state z0 [B] and the final state returned
    (the reference carries Capacitor.z across forward() calls when a script never calls
    reset(), lpf.py:30-49)."""

    @staticmethod
    def forward(ctx, theta, x, r, z0, fs, n_up, n_down):
        need_grad = theta.requires_grad
        th = theta.detach().contiguous()
        y, zs, zT = binding.clipper_fwd(x, th, fs, r=r, n_up=n_up, n_down=n_down, want_stash=need_grad,
                                        z0=z0, want_zT=True)
        ctx.cfg = (fs, n_up, n_down)
        ctx.has_r = r is not None
        if need_grad:
            ctx.save_for_backward(th, x, zs, *([r] if r is not None else []))
        ctx.mark_non_differentiable(zT)
        return y, zT

    @staticmethod
    def backward(ctx, gy, _gzT):
        fs, n_up, n_down = ctx.cfg
        saved = ctx.saved_tensors
        th, x, zs = saved[0], saved[1], saved[2]
        r = saved[3] if ctx.has_r else None
        gtheta, _ = binding.clipper_bwd(x, th, fs, zs, gy.contiguous(), r=r, n_up=n_up, n_down=n_down)
        return gtheta, None, None, None, None, None, None


def clipper_stateful(theta, x, fs, r=None, n_up=1, n_down=1, z0=None):
    """Returns (y [T,B], zT [B])."""
    return _ClipperStatefulFn.apply(theta, x, r, z0, float(fs), int(n_up), int(n_down))
