"""Autograd glue: one torch.autograd.Function per fused sequence loop.

The forward call replaces the reference's whole per-sample Python loop
(clipper_pot.py:103-127) by one HIP launch; backward replaces tape.gradient over that loop
(clipper_pot.py:246-268) by one reverse-sweep launch.  torch only carries tensors between
the two and into the loss / optimizer.
"""
import math
import weakref
from collections import namedtuple

import torch

from . import binding

# ---- time-parallel plan -------------------------------------------------------------------
# k_fwd / warmup / tol: chunks, warm-up steps and verified tolerance of the speculative forward
# (csrc/wdf_clipper.h "Time-parallel variants"); k_bwd: chunks of the exact reverse sweep.
TpPlan = namedtuple("TpPlan", ["k_fwd", "warmup", "tol", "k_bwd"])

N_SIMD = 256 * 4          # MI355X: 256 CUs x 4 SIMDs


def plan_time_parallel(B, T, R, C, fs, tol=1.0e-6, time_major=False, R_min=None):
    """Choose chunk counts from the batch shape and the circuit's memory.  R: the source
    resistance; with a per-sample resistance pass the LARGEST value present and the smallest as
    R_min: the memory |1 - 2p| grows with R above Rc = 1/(2 C fs) and with 1/R below it, so the
    slowest sequence is at one of the two ends.

    Sequential mode gives ceil(B/64) waves for 1024 SIMDs, each a dependent chain (a dependent
    VALU op completes every ~3.3 ns on gfx950 where a SIMD could issue one every ~2 ns: tools/ubench).  The
    plan aims at ~2 waves per SIMD for the forward (every extra chunk costs a warm-up) and 4-8
    for the reverse sweep (no redundancy there; 8 when x is time-major, i.e. all loads coalesced).  The forward's warm-up must outlast the circuit's
    memory: with the diode off the state contracts by (1 - 2p) per sample (p = Rc/(R+Rc),
    Rc = 1/(2 C fs)); W is the number of steps that shrinks an O(10 V) error below 1e-7
    (measured miss at the headline circuit: 3e-8 against a verified tolerance of 1e-6).  If W would make chunks more than 2x redundant, fewer chunks are used; if
    even 2 chunks do not pay, the forward stays sequential (k_fwd = 1).  The verify kernel
    still guards every run, so a bad estimate costs time, never correctness.
    """
    waves = max(1, -(-B // 64))
    Rc = 1.0 / (2.0 * float(C) * float(fs))
    rho = max(abs(1.0 - 2.0 * Rc / (float(Rv) + Rc)) for Rv in ((R,) if R_min is None else (R, R_min)))
    if rho <= 0.0:
        W = 8
    elif rho >= 1.0 - 1e-9:
        W = T
    else:
        W = int(math.ceil(math.log(1.0e-8) / math.log(rho)))
    W = max(32, -(-W // 32) * 32)
    k_fwd = max(1, min((2 * N_SIMD) // waves, T // max(W, 64)))
    if k_fwd < 2:
        k_fwd = 1
    k_bwd = max(1, min(((8 if time_major else 4) * N_SIMD) // waves, T // 64))
    return TpPlan(k_fwd, W, float(tol), k_bwd)


_R_MAX_CACHE = {}      # id(tensor) -> (weakref to the tensor, version, max)


def resistance_max(r):
    """max of a per-sample resistance tensor: the planner needs the slowest sequence's memory.
    One device sync, cached per tensor OBJECT and version (a training set is the same tensor every
    epoch; a fresh tensor -- even one the allocator put at a recycled address -- is looked at again)."""
    hit = _R_MAX_CACHE.get(id(r))
    if hit is None or hit[0]() is not r or hit[1] != r._version:
        if len(_R_MAX_CACHE) > 256:
            for k in [k for k, v in _R_MAX_CACHE.items() if v[0]() is None]:
                del _R_MAX_CACHE[k]
        lo, hi = torch.aminmax(r)
        hit = (weakref.ref(r), r._version, float(hi), float(lo))
        _R_MAX_CACHE[id(r)] = hit
    return hit[2]


def resistance_min(r):
    """min of a per-sample resistance tensor (same cache entry as resistance_max)."""
    resistance_max(r)
    return _R_MAX_CACHE[id(r)][3]


_SPLIT_CACHE = {}      # id(x) -> (weakref to x, version, xv, r)


def split_channels(x, with_r, time_major=False, anchor=None):
    """The kernels' contiguous views of a [B,T,2] script input (Vin, R: clipper_pot.py:68-70):
    xv = x[..., 0] and r = x[..., 1], as [B,T] or -- time_major -- as [T,B].  Cached per input tensor
    object, version and layout: the reference feeds the same train_X every epoch
    (clipper_pot.py:245-248), so the de-interleaving (and transposing) copies and the planner's look
    at max R happen once, not per forward."""
    xt = x.as_subclass(torch.Tensor)
    if xt.dim() == 2:
        xt = xt.unsqueeze(-1)
    # the cache hangs on the object the CALLER holds (`anchor`: the tensor as the script passed it),
    # not on the normalised views made from it, which are new objects every call
    anchor = x if anchor is None else anchor
    key = (id(anchor), bool(time_major))
    hit = _SPLIT_CACHE.get(key)
    ver = getattr(anchor, "_version", None)
    if hit is None or hit[0]() is not anchor or hit[1] != ver or (with_r and hit[3] is None):
        if len(_SPLIT_CACHE) > 64:
            for k in [k for k, v in _SPLIT_CACHE.items() if v[0]() is None]:
                del _SPLIT_CACHE[k]
        def chan(c):
            v = xt[:, :, c]
            return v.t().contiguous() if time_major else v.contiguous()
        hit = (weakref.ref(anchor), ver, chan(0), chan(1) if with_r else None)
        _SPLIT_CACHE[key] = hit
    return hit[2], (hit[3] if with_r else None)


LAST_TP_STATUS = {"status": None}     # device status word of the most recent time-parallel forward


class _ClipperFn(torch.autograd.Function):
    """y[T,B] = clipper(theta = {Is, nVt, R, C}, x[B,T] (, r[B,T]))."""

    @staticmethod
    def forward(ctx, theta, x, r, fs, n_up, n_down, time_major, tp):
        need_grad = theta.requires_grad
        th = theta.detach().contiguous()
        if tp is not None and not time_major and tp.k_fwd > 1:
            y, zs, _, st = binding.clipper_fwd_tp(x, th, fs, tp.k_fwd, tp.warmup, tp.tol, r=r, n_up=n_up,
                                                  n_down=n_down, want_stash=need_grad)
            LAST_TP_STATUS["status"] = st
        else:
            y, zs, _ = binding.clipper_fwd(x, th, fs, r=r, n_up=n_up, n_down=n_down, want_stash=need_grad,
                                           time_major=time_major)
        ctx.tp = tp if (tp is not None and not time_major) else None
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
        if ctx.tp is not None and ctx.tp.k_bwd > 1:
            gtheta, _ = binding.clipper_bwd_tp(x, th, fs, zs, gy.contiguous(), ctx.tp.k_bwd, r=r, n_up=n_up,
                                               n_down=n_down)
        else:
            gtheta, _ = binding.clipper_bwd(x, th, fs, zs, gy.contiguous(), r=r, n_up=n_up, n_down=n_down,
                                            time_major=time_major)
        return gtheta, None, None, None, None, None, None, None


def clipper(theta, x, fs, r=None, n_up=1, n_down=1, time_major=False, tp=None):
    """Diode-clipper sequence loop on the GPU.  theta: float32[4] = {Is, nVt, R, C} on the
    device (may require grad); x, r: [B,T] (or [T,B] when time_major).  Returns y [T,B].
    tp: a TpPlan (plan_time_parallel) to run the time-parallel kernels, None = sequential."""
    return _ClipperFn.apply(theta, x, r, float(fs), int(n_up), int(n_down), bool(time_major), tp)


class _ClipperStatefulFn(torch.autograd.Function):
    """(y[T,B], zT[B]) = clipper(theta, x (, r), z0[B]): the sequential kernels with the capacitor
    state handed in and out -- lpf.py-style loops that never call reset() carry it from one
    forward() to the next -- differentiable w.r.t. theta and z0 (gradients flow back through zT)."""

    @staticmethod
    def forward(ctx, theta, x, r, z0, fs, n_up, n_down):
        th = theta.detach().contiguous()
        z0d = None if z0 is None else z0.detach().contiguous()
        need = theta.requires_grad or (z0 is not None and z0.requires_grad)
        y, zs, zT = binding.clipper_fwd(x, th, fs, r=r, n_up=n_up, n_down=n_down, want_stash=need, z0=z0d, want_zT=True)
        ctx.cfg = (fs, n_up, n_down)
        ctx.has_r, ctx.has_z0 = r is not None, z0 is not None
        if need:
            ctx.save_for_backward(th, x, zs, *([r] if r is not None else []))
        return y, zT

    @staticmethod
    def backward(ctx, gy, gzT):
        fs, n_up, n_down = ctx.cfg
        saved = ctx.saved_tensors
        th, x, zs = saved[0], saved[1], saved[2]
        r = saved[3] if ctx.has_r else None
        T, B = zs.shape
        gy = torch.zeros((T, B), dtype=torch.float32, device=x.device) if gy is None else gy.contiguous()
        gtheta, gz0 = binding.clipper_bwd(x, th, fs, zs, gy, r=r, n_up=n_up, n_down=n_down, want_gz0=True,
                                          gzT=None if gzT is None else gzT.contiguous())
        return gtheta, None, None, (gz0 if ctx.has_z0 else None), None, None, None


def clipper_stateful(theta, x, fs, r=None, n_up=1, n_down=1, z0=None):
    """Diode-clipper loop with explicit capacitor state: -> (y [T,B], zT [B]).  z0 [B] or None (reset)."""
    return _ClipperStatefulFn.apply(theta, x, r, z0, float(fs), int(n_up), int(n_down))


class _ClipperAsymFn(torch.autograd.Function):
    """y [T,B] = clipper with two different antiparallel diodes, differentiable w.r.t.
    theta6 = {Is_up, nVt_up, Is_down, nVt_down, R, C} (csrc/wdf_asym.h).  mode: fp64 Newton on the exact Shockley pair
    (default) or the fp32 Wright-omega closed form -- each differentiates its own forward.  The reverse sweep is the
    time-parallel one (no root re-solve, chunks composed exactly: wdf_clipper_asym_bwd_tp) unless tp says k_bwd = 0, which
    keeps the sequential Newton-re-solving sweep (wdf_clipper_asym_bwd; Newton mode only)."""

    @staticmethod
    def forward(ctx, theta6, x, fs, tol, max_iter, tp, mode):
        th = theta6.detach().contiguous()
        if tp is not None and tp.k_fwd > 1:      # time chunks, verified on the device (wdf_clipper_asym_fwd_tp)
            y, zT, zs, st = binding.clipper_asym_fwd_tp(x, th, fs, mode, tp.k_fwd, tp.warmup, tol=tol,
                                                        max_iter=max_iter, verify_tol=tp.tol, want_stash=True, want_zT=True)
            LAST_TP_STATUS["status"] = st
        else:
            y, zT, _, zs = binding.clipper_asym_fwd(x, th, fs, mode, tol=tol, max_iter=max_iter, want_stash=True, want_zT=True)
        B, T = x.shape
        if tp is not None:
            k_bwd = int(tp.k_bwd)
        else:                                    # as many chunks as give every SIMD ~2 waves, none shorter than 64 steps
            k_bwd = max(1, min(T // 64, (2 * N_SIMD) // max(1, -(-B // 64))))
        ctx.cfg = (fs, tol, max_iter, mode, k_bwd)
        ctx.save_for_backward(th, x, zs, zT)
        return y

    @staticmethod
    def backward(ctx, gy):
        fs, tol, max_iter, mode, k_bwd = ctx.cfg
        th, x, zs, zT = ctx.saved_tensors
        if k_bwd < 1 and mode == binding.ASYM_NEWTON_F64:
            g = binding.clipper_asym_bwd(x, th, fs, zs, gy.contiguous(), tol=tol, max_iter=max_iter)
        else:
            g = binding.clipper_asym_bwd_tp(x, th, fs, mode, zs, zT, gy.contiguous(), max(1, k_bwd))
        return g, None, None, None, None, None, None


def clipper_asym(theta6, x, fs, tol=1.0e-12, max_iter=50, tp=None, mode=None):
    """Two-different-diode clipper loop (BASELINE config 5) with gradients to all six parameters.
    mode: binding.ASYM_NEWTON_F64 (default: the exact model) or binding.ASYM_OMEGA_F32 (the fp32 closed form).
    tp: a TpPlan (plan_time_parallel with the circuit's R, C) -> the forward runs in k_fwd verified time chunks and the
    reverse sweep in k_bwd exact ones (k_bwd = 0: the sequential sweep)."""
    mode = binding.ASYM_NEWTON_F64 if mode is None else int(mode)
    return _ClipperAsymFn.apply(theta6, x, float(fs), float(tol), int(max_iter), tp, mode)


class MseStep:
    """Fused training step for the mean-squared-error loss (lpf.py:78, clipper_pot.py:176):
    forward, loss and reverse sweep in two kernel launches + three tiny ones, all buffers
    preallocated.  step() returns device tensors (sse[1], gtheta[4]): the sum of squared errors
    over THIS batch and d(mean over n_global)/d{Is, nVt, R, C}; nothing synchronises.

    time_major: x (and r) are [T,B] -- the layout an engine keeps its training inputs resident
    in (one transpose when the dataset is loaded; the reference trains on the same train_X
    for all 501 epochs, clipper_pot.py:245-248), giving fully coalesced loads."""

    def __init__(self, B, T, fs, tp, device, n_up=1, n_down=1, n_global=None, time_major=False, loss="mse",
                 skip=0, sums_allreduce=None, warm=False, max_warm_tiles=16):
        """loss: "mse" (mean over n_global samples) or "mse+esr", the training loss of
        clipper_pot.py:177 evaluated past `skip` samples (:232,248; n_global then counts the samples
        past skip over all ranks).  sums_allreduce: in-place SUM all-reduce for the two float64 loss
        sums when the batch is sharded (wdf_hip.dist.allreduce_sum_).
        warm: this stepper always sees the SAME resident x (a training set re-visited every epoch,
        clipper_pot.py:245-269): keep the forward's warm-start state between calls
        (binding.TpWarmState; call reset_warm() if x changes)."""
        if loss not in ("mse", "mse+esr"):
            raise binding.WdfHipError(f"unknown loss {loss!r}")
        if loss == "mse" and skip:
            raise binding.WdfHipError("skip is only meaningful with loss='mse+esr'")
        self.B, self.T, self.fs, self.tp = B, T, float(fs), tp
        self.n_up, self.n_down, self.time_major = n_up, n_down, bool(time_major)
        self.loss_kind, self.skip, self.sums_allreduce = loss, int(skip), sums_allreduce
        self.n_global = float(n_global if n_global is not None else B * (T - self.skip))
        self.gscale = 2.0 / self.n_global
        if loss == "mse+esr":
            self.ws_l = torch.empty((binding.lib().wdf_loss_sums_ws_bytes(),), dtype=torch.uint8, device=device)
            self.sums = torch.zeros((2,), dtype=torch.float64, device=device)
            self.gcoef = torch.zeros((2,), dtype=torch.float32, device=device)
            self.loss = torch.zeros((3,), dtype=torch.float32, device=device)      # mse, esr, mse + esr
            self.sums10 = torch.zeros((10,), dtype=torch.float32, device=device)   # one-pass step: {S, E, gP[4], gQ[4]}
        L = binding.lib()
        kf, kb = (tp.k_fwd, tp.k_bwd) if tp is not None else (1, 1)
        self.ws_f = torch.empty((max(16, L.wdf_clipper_fwd_tp_ws_bytes(B, kf)),), dtype=torch.uint8, device=device)
        self.ws_b = binding.bwd_tp_workspace(B, kb, device)
        self.status = torch.zeros((4,), dtype=torch.int32, device=device)
        # one fused buffer [sse, dIs, dnVt, dR, dC]: it is also the all-reduce payload
        self.out = torch.zeros((5,), dtype=torch.float32, device=device)
        self.sse, self.gtheta = self.out[0:1], self.out[1:5]
        self.y = self.zs = self.zT = None
        self.ws_s, self.ws_s_k = None, None          # workspace of step_fused (made on first use)
        self.warm = None
        if warm and tp is not None and tp.k_fwd > 1:
            self.warm = binding.TpWarmState(B, T, tp.k_fwd, max(max_warm_tiles, -(-tp.warmup // binding.warm_unit())), device)

    def reset_warm(self):
        if self.warm is not None:
            self.warm.reset()

    def forward(self, theta, x, r=None):
        tp = self.tp
        if tp is not None and tp.k_fwd > 1:
            self.y, self.zs, self.zT, _ = binding.clipper_fwd_tp(
                x, theta, self.fs, tp.k_fwd, tp.warmup, tp.tol, r=r, n_up=self.n_up, n_down=self.n_down,
                want_zT=True, ws=self.ws_f, status=self.status, time_major=self.time_major, state=self.warm)
        else:
            self.y, self.zs, self.zT = binding.clipper_fwd(x, theta, self.fs, r=r, n_up=self.n_up,
                                                           n_down=self.n_down, want_zT=True,
                                                           time_major=self.time_major)
        return self.y

    def backward(self, theta, x, target, r=None, adam=None):
        """adam: a binding.Adam(4, ...) to update theta in the sweep's own last kernel (single rank,
        MSE loss); otherwise the caller applies its optimizer to self.gtheta afterwards."""
        kb = self.tp.k_bwd if self.tp is not None else 1
        if adam is not None and self.loss_kind == "mse":
            binding.clipper_bwd_mse_tp_adam(x, theta, self.fs, self.zs, self.zT, target, self.gscale, kb, adam, r=r,
                                            n_up=self.n_up, n_down=self.n_down, gtheta=self.gtheta, sse=self.sse,
                                            ws=self.ws_b, time_major=self.time_major)
            return self.sse, self.gtheta
        if self.loss_kind == "mse+esr":
            # loss sums over this rank's y (one streaming pass), made global, then the two
            # coefficients of dL/dy = ga (y - t) + gb y; the sweep itself is the MSE one
            binding.loss_sums(self.y, target, self.skip, sums=self.sums, ws=self.ws_l)
            if self.sums_allreduce is not None:
                self.sums_allreduce(self.sums)
            binding.esr_coef(self.sums, self.n_global, float(torch.finfo(torch.float64).eps), self.gcoef, self.loss)
            binding.clipper_bwd_esr_tp(x, theta, self.fs, self.zs, self.zT, target, self.gcoef, self.skip, kb, r=r,
                                       n_up=self.n_up, n_down=self.n_down, gtheta=self.gtheta, sse=self.sse,
                                       ws=self.ws_b, time_major=self.time_major)
            return self.sse, self.gtheta
        binding.clipper_bwd_mse_tp(x, theta, self.fs, self.zs, self.zT, target, self.gscale, kb, r=r,
                                   n_up=self.n_up, n_down=self.n_down, gtheta=self.gtheta, sse=self.sse,
                                   ws=self.ws_b, time_major=self.time_major)
        return self.sse, self.gtheta

    def step(self, theta, x, target, r=None):
        self.forward(theta, x, r)
        return self.backward(theta, x, target, r)

    def step_fused(self, theta, x, target, r=None, adam=None):
        """The same step in ONE pass over the data (csrc/wdf_clipper_fused.h): forward, loss and gradient with x and target
        read once and y written once -- no state stash, one root solve per sample, the gradient carried forward as the
        state's tangent.  Chunking / verification / warm start as forward(); fills self.y, self.sse, self.gtheta (for
        "mse+esr" also self.loss = {mse, esr, mse + esr}) and, with `adam` on a single rank, updates theta in the same launch.
        With sums_allreduce set (sharded batch, "mse+esr") the ten sums are all-reduced between the pass and its finish."""
        tp = self.tp
        k = tp.k_fwd if tp is not None else 1
        if self.ws_s is None or self.ws_s_k != k:
            self.ws_s, self.ws_s_k = binding.step_mse_workspace(self.B, binding.lib().wdf_clipper_tp_chunks(self.T, k), x.device), k
            self.y = torch.empty((self.T, self.B), dtype=torch.float32, device=x.device)
            self.zs = self.zT = None
        W, tol = (tp.warmup, tp.tol) if tp is not None else (0, 1.0e-6)
        if self.loss_kind == "mse":
            binding.clipper_step_mse_tp(x, theta, self.fs, target, self.gscale, k, W, tol=tol, r=r, n_up=self.n_up,
                                        n_down=self.n_down, y=self.y, ws=self.ws_s, status=self.status, state=self.warm,
                                        gtheta=self.gtheta, sse=self.sse, opt=adam, time_major=self.time_major)
            return self.sse, self.gtheta
        eps = float(torch.finfo(torch.float64).eps)
        local = self.sums_allreduce is None
        binding.clipper_step_esr_tp(x, theta, self.fs, target, self.n_global, eps, self.skip, k, W, tol=tol, r=r, n_up=self.n_up,
                                    n_down=self.n_down, y=self.y, ws=self.ws_s, status=self.status, state=self.warm,
                                    sums10=self.sums10, gtheta=self.gtheta, loss3=self.loss, finish=local,
                                    opt=adam if local else None, time_major=self.time_major)
        if not local:
            self.sums_allreduce(self.sums10)
            binding.esr_finish(self.sums10, self.n_global, eps, gtheta=self.gtheta, loss3=self.loss)
        self.sse.copy_(self.sums10[0:1])
        return self.sse, self.gtheta


def autotune_time_parallel(theta, x, target, fs, plan, time_major=False, n_up=1, n_down=1, reps=7, r=None):
    """Refine a TpPlan by timing a few candidates on the actual batch (HIP events on the launch
    stream, a handful of launches each): the chunk counts, and the warm-up -- the planned W
    assumes a 10 V error at the chunk start; on real data the diodes clamp the state to ~1 V, so
    W - 32 is tried too and kept only if the verify kernel reports a miss 8x below the tolerance
    (every run is still verified, so a drifting circuit costs a repair, never a wrong result)."""
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if plan is None:
        return plan

    def timed(fn):
        """median of `reps` single-call timings (a mean lets one slow launch flip the plan)"""
        fn()
        e0, e1 = binding.Event(), binding.Event()
        ts = []
        for _ in range(reps):
            e0.record()
            fn()
            e1.record()
            ts.append(e0.elapsed_ms(e1))
        return sorted(ts)[len(ts) // 2]

    best_f, best_b = plan.k_fwd, plan.k_bwd
    if plan.k_fwd > 1:
        cands = sorted({k for k in (plan.k_fwd // 2, plan.k_fwd, plan.k_fwd * 2) if 2 <= k <= T // max(plan.warmup // 2, 64)})
        times = {}
        for k in cands:
            st = MseStep(B, T, fs, plan._replace(k_fwd=k), x.device, n_up=n_up, n_down=n_down, time_major=time_major)
            times[k] = timed(lambda: st.forward(theta, x, r))
        best_f = _pick(times, plan.k_fwd)
        t_now = times[best_f]
        for _ in range(8):                                  # shorter warm-ups, 32 steps at a time, while they pay
            if plan.warmup < 96:
                break
            shorter = plan._replace(k_fwd=best_f, warmup=plan.warmup - 32)
            st = MseStep(B, T, fs, shorter, x.device, n_up=n_up, n_down=n_down, time_major=time_major)
            t_short = timed(lambda: st.forward(theta, x, r))
            stat = binding.tp_status(st.status)
            if not (stat["n_bad"] == 0 and stat["max_miss"] <= plan.tol / 8.0 and t_short < 0.98 * t_now):
                break
            plan, t_now = shorter, t_short
    st = MseStep(B, T, fs, plan._replace(k_fwd=best_f), x.device, n_up=n_up, n_down=n_down, time_major=time_major)
    st.forward(theta, x, r)
    times = {}
    for k in sorted({k for k in (plan.k_bwd // 2, plan.k_bwd, plan.k_bwd * 2) if 1 <= k <= max(1, T // 32)}):
        st.tp = plan._replace(k_fwd=best_f, k_bwd=k)
        st.ws_b = binding.bwd_tp_workspace(B, k, x.device)
        times[k] = timed(lambda: st.backward(theta, x, target, r))
    # reverse sweep: among the candidates within 2 % of the fastest take the fewest chunks (less
    # combine work and workspace; also keeps the choice stable from run to run)
    t_best = min(times.values())
    best_b = min(k for k, t in times.items() if t <= 1.02 * t_best)
    return plan._replace(k_fwd=best_f, k_bwd=best_b)


def autotune_fused(theta, x, target, fs, plan, time_major=False, n_up=1, n_down=1, reps=7, r=None, warm=True):
    """The chunk count of the one-pass step (MseStep.step_fused), by timing the planned count, half and double
    on the actual batch (median of single-launch timings of the warm-started step, as the training loop will run it)."""
    if plan is None:
        return plan
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape

    def timed(fn):
        for _ in range(8):                   # past the cold call and the warm-start controller's descent (one tile per call)
            fn()
        e0, e1 = binding.Event(), binding.Event()
        ts = []
        for _ in range(reps):
            e0.record()
            fn()
            e1.record()
            ts.append(e0.elapsed_ms(e1))
        return sorted(ts)[len(ts) // 2]

    # plan_time_parallel counts one sequence per lane (two waves per SIMD at k_fwd chunks); with an even batch the one-pass
    # step runs two per lane, i.e. half the waves: the planned count is doubled to keep two waves per SIMD
    k_cap = T // max(plan.warmup // 2, 64)
    planned = plan.k_fwd * 2 if (B % 2 == 0 and not binding.ONE_SEQUENCE_PER_LANE and plan.k_fwd * 2 <= k_cap) else plan.k_fwd
    cands = {k for k in (max(1, planned // 2), planned, planned * 2) if k == 1 or k <= k_cap}
    if warm:
        # Round 6: a stepper that keeps its warm-start state pays the cold warm-up ONCE, then runs 0-16 warm-up steps per chunk
        # -- so its chunk count is not bounded by the cold warm-up's redundancy but by the chip: enough chunks for two waves
        # on every SIMD, down to 32-step chunks (a few thousand sequences: the per-rank share of a strong-scaling run).  The
        # step's tail no longer grows with the chunk count on one wave (clipper_fused_finish_kernel: 8 waves per tile).
        per_tile = 128 if (B % 2 == 0 and not binding.ONE_SEQUENCE_PER_LANE) else 64
        fill = (2 * N_SIMD) // max(1, -(-B // per_tile))
        for k in (fill // 2, fill):
            k = min(k, T // 32)
            if k >= 2 and k > max(cands):
                cands.add(int(binding.lib().wdf_clipper_tp_chunks(T, k)))
    times = {}
    for k in sorted(cands):
        st = MseStep(B, T, fs, plan._replace(k_fwd=k), x.device, n_up=n_up, n_down=n_down, time_major=time_major, warm=warm)
        times[k] = timed(lambda: st.step_fused(theta, x, target, r))
        if binding.tp_status(st.status)["n_bad"]:
            del times[k]                     # a chunking whose speculation fails on this data is not a candidate
    return plan._replace(k_fwd=_pick(times, planned)) if times else plan


class _ClipperMseFn(torch.autograd.Function):
    """mean((clipper(theta, x) - target)^2) as the one-pass training step (MseStep.step_fused: forward, loss
    and gradient in one kernel, one stepper per shape, buffers reused); backward only scales the stored
    gradient -- no y-sized loss temporaries, no dL/dy array, no state stash."""
    _steppers = {}

    @staticmethod
    def forward(ctx, theta, x, r, target, fs, n_up, n_down, tp, time_major):
        B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
        key = (B, T, float(fs), n_up, n_down, tp, x.device, time_major)
        st = _ClipperMseFn._steppers.get(key)
        if st is None:
            if len(_ClipperMseFn._steppers) > 8:
                _ClipperMseFn._steppers.clear()
            st = _ClipperMseFn._steppers[key] = MseStep(B, T, fs, tp, x.device, n_up=n_up, n_down=n_down,
                                                        time_major=time_major)
        th = theta.detach().contiguous()
        LAST_TP_STATUS["status"] = st.status
        if not theta.requires_grad:          # validation / no_grad losses: forward + one streaming pass for the sum
            st.forward(th, x, r)
            d = st.y - target
            return torch.sum(d * d) / float(B * T)
        sse, g = st.step_fused(th, x, target, r)      # forward, loss and gradient in one pass over the data
        ctx.save_for_backward(g.clone())
        return sse[0] / float(B * T)

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        return gl * g, None, None, None, None, None, None, None, None     # (never reached when theta needs no grad)


class _ResidentMseFn(torch.autograd.Function):
    """_ClipperMseFn for component values resident in a parameter block (lowering.Circuit.to_device): the one-pass step
    reads theta straight from the block; the Variables that own its entries are the differentiable inputs, each handed
    its entry of the gradient as a device scalar.  One stepper per (x, target) pair, with the warm-start state the
    repeated visits of a training set allow."""

    @staticmethod
    def forward(ctx, st, block, x, r, target, inv_n, idx, *live):
        st.step_fused(block, x, target, r)
        # {sse | mse + esr, gtheta[4]} of THIS call (a validation pass may run before backward)
        out = st.out.clone() if st.loss_kind == "mse" else torch.cat((st.loss[2:3], st.gtheta))
        ctx.save_for_backward(out)
        ctx.idx = idx
        ctx.mark_non_differentiable(out)
        return out[0] * inv_n, out            # (out: what GradientTape.gradient hands out directly, compat_tf)

    @staticmethod
    def backward(ctx, gl, _):
        (out,) = ctx.saved_tensors
        g = gl * out
        return (None,) * 7 + tuple(g[1 + i] for i in ctx.idx)


def clipper_mse(theta, x, target, fs, r=None, n_up=1, n_down=1, tp=None, time_major=False):
    """Scalar mean-squared error of the clipper output against target [T,B], differentiable w.r.t.
    theta = {Is, nVt, R, C} (float32[4] on the device); the fused path of lpf.py:87-90-style loops.
    time_major: x (and r) are [T,B]."""
    if tp is None:
        tp = TpPlan(1, 32, 1.0e-6, 1)
    return _ClipperMseFn.apply(theta, x, r, target, float(fs), int(n_up), int(n_down), tp, bool(time_major))


_TUNED = {}      # (B, T, n_up, n_down, per-sample R?, R, C to 2 digits, fs, device, layout, consumer) -> TpPlan


def tuned_plan(theta, x, r, fs, R_plan, C, n_up=1, n_down=1, min_samples=1 << 22, time_major=False, R_min=None, fused=False):
    """plan_time_parallel, refined once per (shape, circuit) by timing candidates on the batch when it is large enough for
    the ~0.1 s of set-up to pay (>= 4 M samples); later calls with the same shape and (to two digits) the same R and C
    reuse the result.  fused: the consumer is the one-pass training step (clipper_mse) -- its chunk count is what is
    tuned (autotune_fused); otherwise the forward / reverse-sweep pair (autotune_time_parallel).
    The key is built from values the caller holds on the HOST (R_plan, C: floats); theta is a device tensor and is only
    handed to the kernels on a miss -- a hit costs no device-to-host copy, no synchronisation.  The diode's own
    parameters are not part of the key: they do not enter the warm-up estimate, training moves them slowly, and every
    forward is verified whatever the plan."""
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    plan = plan_time_parallel(B, T, R_plan, C, fs, time_major=time_major, R_min=R_min)
    if B * T < min_samples or plan.k_fwd < 2:
        return plan
    key = (B, T, n_up, n_down, r is not None, f"{R_plan:.1e}", f"{C:.1e}", float(fs), str(x.device), time_major, bool(fused))
    hit = _TUNED.get(key)
    if hit is None:
        if len(_TUNED) > 32:
            _TUNED.clear()
        with torch.no_grad():
            zero_target = torch.zeros((T, B), dtype=torch.float32, device=x.device)
            tune = autotune_fused if fused else autotune_time_parallel
            hit = _TUNED[key] = tune(theta.detach(), x, zero_target, fs, plan, time_major=time_major,
                                     n_up=n_up, n_down=n_down, reps=5, r=r)
    return hit


def _pick(times, planned):
    """The fastest candidate, but the planned one unless another is at least 3 % faster (timing
    noise should not flip the plan between runs)."""
    best = min(times, key=times.get)
    if planned in times and times[best] > 0.97 * times[planned]:
        return planned
    return best
