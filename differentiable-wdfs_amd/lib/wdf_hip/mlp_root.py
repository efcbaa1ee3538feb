"""Host side of the MLP-root diode clipper (csrc/wdf_mlp.h): weight flattening, the autograd
Function and the data-level time-parallel segmentation.

The per-lane reverse-sweep kernel returns g_b[n] = dL/d b[n] and the network inputs
(a[n], log R[n]); the weight gradient  dL/dW = -sum_n g_b[n] dMLP(a[n], lr[n])/dW  is a sum
over B*T independent samples, done by wdf_clipper_mlp_wgrad (fully parallel HIP kernel)."""
import os
import math
import weakref
from collections import namedtuple

import torch

from . import binding
from . import engine
from . import warmstart
from . import compat_tf as tf


def describe(model, with_activation=False):
    """(dense layers, hidden width, number of tanh layers) of a layers.DenseRootModel; raises
    for anything the kernels do not cover."""
    dense = [l for l in model.layers if type(l).__name__ == "DenseLayer"]
    acts = []
    for i, l in enumerate(model.layers):
        if type(l).__name__ == "DenseLayer":
            nxt = model.layers[i + 1] if i + 1 < len(model.layers) else None
            acts.append("tanh" if nxt is tf.nn.tanh else ("relu" if nxt is tf.nn.relu else ""))
    if len(dense) < 2 or acts[-1] != "" or any(a != "tanh" for a in acts[:-1]):
        if with_activation and len(dense) >= 2 and acts[-1] == "" and all(a == "relu" for a in acts[:-1]):
            pass                                                  # relu hidden layers (layers.py:63-65): the resident step's kernels carry them
        else:
            raise binding.WdfHipError(f"MLP root: expected tanh hidden layers and a linear output, got {acts}")
    sizes = [int(dense[0].kernel.shape[1])] + [int(d.kernel.shape[2]) for d in dense]
    hidden = sizes[1]
    if sizes[0] != 2 or sizes[-1] != 1 or any(s != hidden for s in sizes[1:-1]):
        raise binding.WdfHipError(f"MLP root: expected 2 -> H -> ... -> H -> 1, got {sizes}")
    if with_activation:
        return dense, hidden, len(dense) - 1, acts[0]
    return dense, hidden, len(dense) - 1


def flat_weights(dense):
    """kernel[in][out] then bias[out] per layer (the JSON order, layers.py:31-36), differentiable."""
    parts = []
    for d in dense:
        parts += [d.kernel.as_subclass(torch.Tensor)[0].reshape(-1), d.bias.as_subclass(torch.Tensor)[0].reshape(-1)]
    return torch.cat(parts)


def slope_range(dense, log_r, a_max, n=513):
    """(min, max) of db/da of the root b = -MLP(a, log R) over a in [-a_max, a_max] at the given log R values: the
    network evaluated in float64 on the host (609 weights), its derivative by autograd.  What the warm-up estimate of the
    streamed-coefficient kernels needs instead of a diode's |db/da| <= 1 (lowering._plan_dyn): a learned network's slope is
    whatever its weights say."""
    ws = [(d.kernel.as_subclass(torch.Tensor)[0].detach().double().cpu(), d.bias.as_subclass(torch.Tensor)[0].detach().double().cpu())
          for d in dense]
    lo, hi = float("inf"), float("-inf")
    with torch.enable_grad():
        for lr in log_r:
            a = torch.linspace(-float(a_max), float(a_max), int(n), dtype=torch.float64, requires_grad=True)
            h = torch.stack([a, torch.full_like(a, float(lr))], dim=1)
            for i, (k, b) in enumerate(ws):
                h = h @ k + b
                if i + 1 < len(ws):
                    h = torch.tanh(h)
            (g,) = torch.autograd.grad(-h.sum(), a)
            lo, hi = min(lo, float(g.min())), max(hi, float(g.max()))
    return lo, hi


MlpTpPlan = namedtuple("MlpTpPlan", ["k_fwd", "warmup", "warmup_per_wave", "tol", "k_bwd"])
LAST_TP_STATUS = {"status": None}
WARM_START = os.environ.get("WDF_MLP_WARM_START", "1") != "0"        # 0: every call warms its chunks up from z = 0
_TRACE_WARMUP = [] if os.environ.get("WDF_MLP_TRACE_WARMUP") else None    # (probing) per call: (warm-up steps, warm-started?)
_TRACE_VERDICTS = []   # (probing, with _TRACE_WARMUP) per verdict read back: (warm-up, n_bad, max miss, gated waves, sequential waves)
_WARM_START = {}       # (batch object, version, shape, chunks, planned warm-up, r) -> warmstart.WarmUpController + the previous call's states
KAPPA_FROM_FORWARD = os.environ.get("WDF_MLP_KAPPA_FROM_FORWARD", "1") != "0"   # 0: the reverse sweep recomputes kappa
_WARMUP_ADAPT = {}     # (x shape, forward chunks, planned warm-up) -> {"warmup": steps in use, "calls": n}


class _ClipperMlpFn(torch.autograd.Function):
    """y [T,B] = clipper_mlp(theta2 = {R, C}, w, x [B,T] (, r [B,T])).  tp: an MlpTpPlan -> the
    in-kernel time-parallel forward (verified) and the all-steps-parallel exact reverse sweep."""

    @staticmethod
    def forward(ctx, theta2, w, x, r, z0, fs, hidden, n_tanh, want_zT, want_stash=False, tp=None):
        need = theta2.requires_grad or w.requires_grad
        th, wd = theta2.detach().contiguous(), w.detach().contiguous()
        kap = None
        if tp is not None and tp.k_fwd > 1 and not binding.MLP_LANE_PER_SEQUENCE:
            # The warm-up estimate knows the RC network, not the learned root: the first calls of a shape (and
            # every 16th later) look at the verification's verdict -- one 16-byte read-back -- and lengthen
            # the warm-up by half when a wave had to be re-run (a re-run is a whole sequential pass of that wave).
            ad = _WARMUP_ADAPT.setdefault((x.shape, tp.k_fwd, tp.warmup), {"warmup": tp.warmup, "calls": 0})
            # a training call also takes kappa from the forward (the reverse sweep then skips its first pass)
            use_kappa = need and tp.k_bwd > 1 and KAPPA_FROM_FORWARD
            # ... and, the second time it sees a batch, starts every chunk from the state the previous call had there
            # (the weights moved by one optimizer step): a fraction of the warm-up closes that gap.  Keyed by the
            # batch's address and shape; other data at the same address is caught by the verification like any miss.
            warm = None
            if use_kappa and WARM_START and tp.warmup_per_wave is None and z0 is None:
                # keyed on the batch OBJECT and its version (a weak reference: the entry dies with the tensor, another
                # tensor that lands at the same address later is another key, in-place changes of x restart cold)
                wkey = (id(x), x._version, tuple(x.shape), tp.k_fwd, tp.warmup, None if r is None else (id(r), r._version))
                warm = _WARM_START.get(wkey)
                if warm is not None and (warm["xref"]() is not x or (r is not None and warm["rref"]() is not r)):
                    warm = None                                  # (an id reused by another tensor)
                if warm is None:
                    for k in [k for k, v in _WARM_START.items() if v["xref"]() is None or (v["rref"] is not None and v["rref"]() is None)]:
                        del _WARM_START[k]                       # entries whose batch has been freed
                    if len(_WARM_START) >= 8:
                        _WARM_START.pop(next(iter(_WARM_START)))     # oldest out
                    # (a wave or two re-run now and then is the fp32 floor of this path crossing the tolerance --
                    #  plan_mlp_time_parallel's note -- which no warm-up cures: only 8 or more per verdict count as a
                    #  warm-up too short; the weights swing with periods of ~16 calls, so 32 clean calls before less is tried)
                    ctl = warmstart.WarmUpController(ad["warmup"], int(os.environ.get("WDF_MLP_WARM_W", ad["warmup"] // 3)),
                                                     unit=16, floor=32, miss_waves=8, wait_calls=32)
                    warm = _WARM_START[wkey] = {"ctl": ctl, "rows": {}, "idx": None, "idx_key": None,
                                                "xref": weakref.ref(x), "rref": None if r is None else weakref.ref(r)}
            zinit, w_used = None, ad["warmup"]
            if warm is not None:
                # the previous call's verified states at this call's chunk starts (the weights moved by one optimizer step).
                # Order 0 on purpose: with Adam(1e-4, beta_1 0.5) on these weights the trajectory jumps by 1e-2 ... 4e-1 per
                # call and the jumps change sign irregularly -- secant, parabola and fitted AR(1) predictors all do worse
                # (tools/mlp_start_probe.py, profiles/r03_mlp_start_probe.txt); the warm-up's contraction closes the gap.
                warm["ctl"].cold = ad["warmup"]
                w_warm = warm["ctl"].begin(warm["rows"])
                if w_warm is not None:
                    zinit, w_used = warm["rows"][w_warm], w_warm
            hot = zinit is not None
            out = binding.clipper_mlp_fwd_tp(x, th, wd, hidden, n_tanh, fs, tp.k_fwd, w_used, r=r,
                                             warmup_per_wave=tp.warmup_per_wave, tol=tp.tol,
                                             want_stash=need or want_stash, z0=z0, want_zT=want_zT,
                                             want_kappa=use_kappa, zinit=zinit)
            y, zs, zT, st = out[:4]
            kap = out[4] if use_kappa else None
            LAST_TP_STATUS["status"] = st
            LAST_TP_STATUS["warmup_used"] = w_used
            if _TRACE_WARMUP is not None:
                _TRACE_WARMUP.append((w_used, hot))
                if warm is not None:
                    _TRACE_VERDICTS[:] = list(warm["ctl"].verdicts)
            if hot:
                warm["ctl"].end(st, w_used)
            else:
                ad["calls"] += 1
                if tp.warmup_per_wave is None and (ad["calls"] <= 4 or ad["calls"] % 16 == 0):
                    if binding.mlp_tp_status(st)["gated_waves"] > 0:
                        ad["warmup"] = min(-(-int(1.5 * ad["warmup"]) // 16) * 16, int(x.shape[1]))
            if warm is not None:
                # the states the NEXT call may start from: this call's verified trajectory at the chunk starts of every
                # warm-up the controller can ask for -- one gather, [candidates][chunks, B] floats; the stash is not kept alive
                cands = warm["ctl"].candidates()
                if warm["idx_key"] != tuple(cands):
                    starts = [binding.mlp_tp_starts(int(x.shape[1]), tp.k_fwd, w_) for w_ in cands]
                    warm["idx"] = torch.tensor([t for s_ in starts for t in s_], dtype=torch.int64, device=x.device)
                    warm["idx_key"] = tuple(cands)
                rows = zs.index_select(0, warm["idx"]).view(len(cands), -1, zs.shape[1])
                warm["rows"] = {w_: rows[i] for i, w_ in enumerate(cands)}
        else:
            y, zs, zT = binding.clipper_mlp_fwd(x, th, wd, hidden, n_tanh, fs, r=r, want_stash=need or want_stash,
                                                z0=z0, want_zT=want_zT)
        ctx.tp = tp
        ctx.cfg = (fs, hidden, n_tanh, r is not None)
        ctx.kappa = None
        if need:
            ctx.save_for_backward(th, wd, x, zs, *([r] if r is not None else []))
            ctx.kappa = kap                                  # (not an input or output: a plain attribute)
        if want_zT:
            ctx.mark_non_differentiable(zT)
        if want_stash:
            zs_out = zs.detach()
            ctx.mark_non_differentiable(zs_out)
            return y, zT, zs_out
        return y, zT

    @staticmethod
    def backward(ctx, gy, *_unused):
        fs, hidden, n_tanh, has_r = ctx.cfg
        saved = ctx.saved_tensors
        th, wd, x, zs = saved[:4]
        r = saved[4] if has_r else None
        if ctx.tp is not None and ctx.tp.k_bwd > 1 and not binding.MLP_LANE_PER_SEQUENCE:
            gth, gw = binding.clipper_mlp_bwd_w_tp(x, th, wd, hidden, n_tanh, fs, zs, gy.contiguous(), ctx.tp.k_bwd, r=r,
                                                   kappa=ctx.kappa)
        elif binding.MLP_LANE_PER_SEQUENCE:
            gth, gb, ain, lrin = binding.clipper_mlp_bwd(x, th, wd, hidden, n_tanh, fs, zs, gy.contiguous(), r=r)
            # weight gradient: dL/dw = -sum_n gb[n] dMLP(a[n], lr[n])/dw, all B*T samples in parallel
            gw = binding.clipper_mlp_wgrad(ain, lrin, gb, th, wd, hidden, n_tanh, fs)
        else:
            # one 16-lane row per sequence; the weight gradient is accumulated in the same sweep
            gth, gw = binding.clipper_mlp_bwd_w(x, th, wd, hidden, n_tanh, fs, zs, gy.contiguous(), r=r)
        return gth, gw, None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------ time-parallel segments
# The MLP kernels are sequential in time; with the reference's training set (1340 sequences,
# clipper_pot.py:58) that is 21 waves on a 1024-SIMD chip.  More parallelism is obtained at the
# DATA level, with no extra device code: the time axis is cut into K chunks and every chunk but
# the first becomes its own "sequence" that starts W steps early from z = 0 (the circuit forgets
# its state: |dz'/dz| < 1; the reference itself discards the first 50 outputs, clipper_pot.py:232).
# The outputs of the W warm-up steps are dropped, so autograd sends no gradient there, and the
# adjoint that would cross a chunk boundary flows through the next chunk's warm-up copy of the
# same samples instead (overlapped truncated BPTT; the truncation error decays like the
# forward's).  The state each chunk arrives with is compared with the state the previous chunk
# ended in; if any differ by more than tol the call is redone sequentially.
def segment_plan(B, T, R_slowest, C, fs, tol=1.0e-6):
    """(K, L, W) or None.  W outlasts the slowest (largest-R) sequence's memory.  The row kernels
    put 4 sequences in a wave, so a batch is ceil(B/4) waves; segments are added only until there is
    about one wave per SIMD (1024), and only while a chunk is at least 2 W long (<= 1.5x the work):
    beyond that the redundant warm-up and the gather / scatter copies cost more than they hide."""
    import math
    Rc = 1.0 / (2.0 * float(C) * float(fs))
    rho = abs(1.0 - 2.0 * Rc / (float(R_slowest) + Rc))
    if rho >= 1.0 - 1e-9:
        return None
    W = 8 if rho <= 0.0 else int(math.ceil(math.log(0.01 * tol) / math.log(rho)))
    W = max(8, -(-W // 8) * 8)
    per_wave = 64 if binding.MLP_LANE_PER_SEQUENCE else 4
    waves = max(1, -(-B // per_wave))
    if 4 * waves > engine.N_SIMD:           # measured at 335 waves (1340 sequences): segmenting no longer pays
        return None
    K = min(T // max(2 * W, 64), -(-engine.N_SIMD // waves))
    if K < 2:
        return None
    L = -(-(-(-T // K)) // 8) * 8
    K = -(-T // L)
    return (K, L, W) if K >= 2 and L >= W else None


def clipper_mlp_segmented(theta2, w, x, r, fs, hidden, n_tanh, plan, tol=1.0e-6, z0=None):
    """y [T,B] and the final state [B] through K overlapping segments; redone sequentially if the
    verification fails.  Returns (y, zT, max_miss)."""
    K, L, W = plan
    B, T = x.shape
    Tp = K * L
    xp, rp = x, r
    if Tp > T:                                               # pad the tail (outputs dropped below)
        xp = torch.nn.functional.pad(x, (0, Tp - T))
        if r is not None:
            rp = torch.cat([r, r[:, -1:].expand(B, Tp - T)], dim=1)
    y0, zT0, _ = _ClipperMlpFn.apply(theta2, w, xp[:, :L].contiguous(), None if r is None else rp[:, :L].contiguous(),
                                     z0, fs, hidden, n_tanh, True, True)

    def segments(v):
        return torch.stack([v[:, k * L - W:(k + 1) * L] for k in range(1, K)], dim=0).reshape((K - 1) * B, W + L)

    ys, zTs, zss = _ClipperMlpFn.apply(theta2, w, segments(xp).contiguous(), None if r is None else segments(rp).contiguous(),
                                       None, fs, hidden, n_tanh, True, True)
    # verification: the state each segment has after its warm-up vs the state its predecessor ended in
    arrive = zss[W].reshape(K - 1, B)
    ended = torch.cat([zT0.reshape(1, B), zTs.reshape(K - 1, B)[:-1]], dim=0)
    miss = float((arrive.detach() - ended.detach()).abs().max())
    if not miss <= tol:
        y, zT = _ClipperMlpFn.apply(theta2, w, x.contiguous(), None if r is None else r.contiguous(),
                                    z0, fs, hidden, n_tanh, True, False)
        return y, zT, miss
    y = torch.cat([y0, ys[W:].reshape(L, K - 1, B).permute(1, 0, 2).reshape((K - 1) * L, B)], dim=0)[:T]
    last = T - (K - 1) * L                                   # true steps in the last segment
    zT = zTs.reshape(K - 1, B)[-1] if last == L else zss[W + last].reshape(K - 1, B)[-1]
    return y, zT, miss


_WROW_CACHE = {}       # id(r) -> (weakref, version, (C, fs, tol), int32 tensor, max)


def _warmup_steps(rho, tol):
    rho = min(max(float(rho), 0.0), 1.0 - 1e-9)
    if rho <= 0.0:
        return 16
    return max(16, -(-int(math.ceil(math.log(0.01 * tol) / math.log(rho))) // 16) * 16)


def warmup_per_wave(r, C, fs, tol=1.0e-6):
    """(Experimental; the planner does not use it -- see plan_mlp_time_parallel.)
    int32[ceil(B/4)] warm-up steps for the time-parallel forward, one per 4 consecutive sequences, from the
    slowest memory |1 - 2p| among them (p = Rc/(R+Rc) at the smallest and the largest pot value of the four
    sequences); cached per resistance tensor (a training set is the same tensor every epoch).  Also the max."""
    hit = _WROW_CACHE.get(id(r))
    key = (float(C), float(fs), float(tol))
    if hit is None or hit[0]() is not r or hit[1] != r._version or hit[2] != key:
        if len(_WROW_CACHE) > 64:
            for k in [k for k, v in _WROW_CACHE.items() if v[0]() is None]:
                del _WROW_CACHE[k]
        B = r.shape[0]
        Rc = 1.0 / (2.0 * float(C) * float(fs))
        lo, hi = r.amin(dim=1), r.amax(dim=1)
        pad = (-B) % 4
        if pad:
            lo, hi = torch.cat([lo, lo[-1:].expand(pad)]), torch.cat([hi, hi[-1:].expand(pad)])
        lo, hi = lo.reshape(-1, 4).amin(dim=1).double(), hi.reshape(-1, 4).amax(dim=1).double()
        rho = torch.maximum((1.0 - 2.0 * Rc / (lo + Rc)).abs(), (1.0 - 2.0 * Rc / (hi + Rc)).abs()).clamp(1e-6, 1.0 - 1e-9)
        W = torch.ceil(math.log(0.01 * tol) / torch.log(rho))
        W = (torch.ceil(W / 16.0) * 16.0).clamp(min=16.0).to(torch.int32).contiguous()
        hit = (weakref.ref(r), r._version, key, W, int(W.max()))
        _WROW_CACHE[id(r)] = hit
    return hit[3], hit[4]


def plan_mlp_time_parallel(B, T, r, R_static, C, fs, tol=4.0e-6, hidden=None, n_tanh=None):
    """MlpTpPlan for the in-kernel time-parallel MLP-root kernels, or None when the batch already fills the
    chip or the circuit remembers more than a chunk.  The row kernels are bound by VALU issue (~100 / ~250
    instructions per step and wave, forward / reverse), so chunks pay only until every SIMD has work: about
    two waves per SIMD (measured at 1340 x 2048: 6 chunks best for both sweeps, profiles/r02_mlp_bench.txt).
    Warm-up: ONE value for the batch, 1.15 x the RC network's diode-off memory at the largest / smallest pot
    value -- with the reference's trained 2x16 root the slowest sequences still miss by 1.5e-6 after that
    estimate's 416 steps (the learned root conducts less sharply than the diode it imitates), and a per-wave
    warm-up from the pot value alone (warmup_per_wave) is wrong for SMALL pots: there the slow mode is the
    conducting one, d z'/d z -> -1, not the off-state 1 - 2p (measured: 55 of 335 waves re-run).  The device
    verification covers whatever the estimate gets wrong.  tol 4e-6: two fp32 evaluations of this path (a learned root is
    not a strict contraction the way the diode pair is) differ by 1-2e-6 on their own whatever the warm-up, and a tolerance
    AT that floor re-runs a wave now and then for nothing (a sequential pass of that wave: 0.6 ms) and drives the adaptive
    warm-up up (464 -> 1056 steps at 2e-6 in the bench loop, 672 at 4e-6)."""
    waves = max(1, -(-B // 4))
    if waves >= 2 * engine.N_SIMD or binding.MLP_LANE_PER_SEQUENCE:
        return None
    Rc = 1.0 / (2.0 * float(C) * float(fs))
    ends = (engine.resistance_max(r), engine.resistance_min(r)) if r is not None else (float(R_static),)
    wmax = max(_warmup_steps(abs(1.0 - 2.0 * Rc / (Rv + Rc)), tol) for Rv in ends)
    wmax = -(-int(1.15 * wmax) // 16) * 16
    k_fwd = max(1, min(2 * engine.N_SIMD // waves, T // max(wmax // 2, 64)))
    # The matrix-core forward carries 16 sequences per wave: one wave per SIMD reaches twice the chunks (shorter chunks,
    # and with warm-started chunks the warm-up no longer dominates) -- the library picks that kernel by itself when the
    # row kernel could not cover the chunk count with two waves per SIMD (csrc/wdf_capi_mlp.hip, mlp_fwd_on_matrix_cores).
    # Width 16, three tanh layers only (narrower nets are padded there, deeper ones lengthen the MFMA chain: measured worse).
    k_mfma = min(engine.N_SIMD // max(1, -(-B // 16)), T // 128)
    if (hidden, n_tanh) == (16, 3) and k_mfma > k_fwd and waves * k_mfma > 2 * engine.N_SIMD \
            and os.environ.get("WDF_MLP_FWD_ROW") != "1":
        k_fwd = k_mfma
    k_bwd = max(1, min(2 * engine.N_SIMD // waves, T // 64))
    if os.environ.get("WDF_MLP_K_FWD"):                      # (probing)
        k_fwd = int(os.environ["WDF_MLP_K_FWD"])
    if os.environ.get("WDF_MLP_K_BWD"):
        k_bwd = int(os.environ["WDF_MLP_K_BWD"])
    if k_fwd < 2 and k_bwd < 2:
        return None
    return MlpTpPlan(k_fwd, wmax, None, float(tol), k_bwd)


def clipper_mlp(theta2, w, x, r, z0, fs, hidden, n_tanh, C, R_static=None, time_parallel="auto"):
    """The MLP-root clipper over a batch.  time_parallel: "auto" / an MlpTpPlan -> the in-kernel time-parallel
    kernels (verified forward, exact reverse sweep) when the batch leaves the chip idle; None -> sequential;
    "segments" -> the older data-level segmentation (forward verified, backward truncated: evaluation only).
    Returns (y [T,B], zT [B])."""
    if time_parallel == "auto":
        time_parallel = plan_mlp_time_parallel(x.shape[0], x.shape[1], r, R_static, C, fs, hidden=hidden, n_tanh=n_tanh)
    if isinstance(time_parallel, MlpTpPlan):
        return _ClipperMlpFn.apply(theta2, w, x, r, z0, fs, hidden, n_tanh, True, False, time_parallel)
    if time_parallel != "segments":
        return _ClipperMlpFn.apply(theta2, w, x, r, z0, fs, hidden, n_tanh, True)
    return _clipper_mlp_segments(theta2, w, x, r, z0, fs, hidden, n_tanh, C, R_static)


def _clipper_mlp_segments(theta2, w, x, r, z0, fs, hidden, n_tanh, C, R_static=None):
    """Data-level segmentation (the round-1 path, kept for A/B): the forward is verified to 1e-6, the
    backward is truncated BPTT through the W warm-up steps -- W comes from the RC network's diode-off
    contraction, not from the learned root -- so training uses the in-kernel path above instead."""
    plan = segment_plan(x.shape[0], x.shape[1], engine.resistance_max(r) if r is not None else float(R_static), float(C), fs)
    if plan is not None:
        y, zT, miss = clipper_mlp_segmented(theta2, w, x, r, fs, hidden, n_tanh, plan, z0=z0)
        LAST_SEGMENT_MISS["miss"] = miss
        return y, zT
    return _ClipperMlpFn.apply(theta2, w, x, r, z0, fs, hidden, n_tanh, True)


LAST_SEGMENT_MISS = {"miss": None}


def run_clipper_mlp(circ, x, z0, return_state):
    """Circuit.__call__ for root = layers.DenseRootModel (clipper_pot.py topology only)."""
    if not circ._is_clipper():
        raise binding.WdfHipError("the MLP root is supported on the diode-clipper topology "
                                  "Parallel(ResistiveVoltageSource, Capacitor) (clipper_pot.py:94-101)")
    model, vs, cap = circ.root, circ.top.P1, circ.top.P2
    dense, hidden, n_tanh, act = describe(model, with_activation=True)
    if act == "relu":
        return _run_clipper_relu(circ, x, z0, return_state, dense, hidden, n_tanh)
    dev = x.device
    Rv = vs.R if isinstance(vs.R, torch.Tensor) else torch.tensor(float(vs.R))
    theta2 = torch.stack([Rv.as_subclass(torch.Tensor).float().reshape(()),
                          cap.C.as_subclass(torch.Tensor).float().reshape(())]).to(dev)
    w = flat_weights(dense).float().to(dev)
    xv, r = engine.split_channels(x, circ.per_sample_R is not None, anchor=getattr(circ, "_anchor", None))
    z0t = None if z0 is None else z0.as_subclass(torch.Tensor).to(dev).float().reshape(-1).contiguous()
    y, zT = clipper_mlp(theta2, w, xv, r, z0t, float(cap.FS), hidden, n_tanh, float(cap.C), R_static=Rv,
                        time_parallel=getattr(circ, "time_parallel", None))
    y = y.as_subclass(tf.Tensor)
    return (y, zT.reshape(1, -1)) if return_state else y


def _run_clipper_relu(circ, x, z0, return_state, dense, hidden, n_layers):
    """Circuit.__call__ on a ReLU network (layers.py:63-67): the forward phase of the resident step's kernels (the row /
    matrix-core kernels of the plain path are tanh-only).  Outputs only: the gradient of a ReLU network comes from
    Circuit.mse_esr (to_device), the loss those kernels differentiate."""
    if z0 is not None or return_state or int(x.shape[1]) % 16:
        raise binding.WdfHipError("a ReLU DenseRootModel runs on the resident step's kernels: zero initial state, no returned state, "
                                  "T a multiple of 16")
    if torch.is_grad_enabled() and any(d.kernel.requires_grad or d.bias.requires_grad for d in dense):
        raise binding.WdfHipError("a ReLU DenseRootModel is differentiated by Circuit.to_device() + Circuit.mse_esr(); call the "
                                  "circuit under tf.stop_gradient / torch.no_grad for its outputs alone")
    vs, cap = circ.top.P1, circ.top.P2
    dev = x.device
    if circ.per_sample_R is None and isinstance(vs.R, torch.Tensor) and vs.R.numel() != 1:
        raise binding.WdfHipError("a ReLU DenseRootModel on the resident step's kernels: the source resistance is a scalar, or a "
                                  f"per-sample channel declared with per_sample_R (got a tensor of shape {tuple(vs.R.shape)})")
    # one forward-only stepper per input (clipper_pot.py:251-262 validates on the same val_X every epoch): its buffers, its
    # plan and its warm-start snapshots are kept; only the weights are refreshed
    anchor = getattr(circ, "_anchor", None)
    key = None if anchor is None else (id(anchor), anchor._version, tuple(x.shape))
    cache = circ.__dict__.setdefault("_relu_fwd", {})
    ent = cache.get(key) if key is not None else None
    with torch.no_grad():
        w = flat_weights(dense).float().to(dev).contiguous()
        if ent is None or ent[0]() is not anchor:
            xv, r = engine.split_channels(x, circ.per_sample_R is not None, anchor=anchor)
            Rv = None if circ.per_sample_R is not None else float(vs.R)
            st = MlpTrainStep(xv, r, torch.zeros((int(x.shape[1]), int(x.shape[0])), dtype=torch.float32, device=dev), w.clone(), hidden,
                              n_layers, float(cap.FS), float(cap.C), R_static=Rv, skip=0, adam=None, activation="relu")
            if key is not None:
                if len(cache) >= 2:
                    cache.pop(next(iter(cache)))
                import weakref
                cache[key] = (weakref.ref(anchor), st)
        else:
            st = ent[1]
            st.w.copy_(w)
        st.forward_only()
    return st.y.clone().as_subclass(tf.Tensor)


# ------------------------------------------------------------------------------ the resident training step
# csrc/wdf_mlp_step.h: clipper_pot.py's epoch (forward over the whole training set, MSE + ESR past skip_samples, gradient
# to the DenseRootModel weights, Adam) as five launches steered on the device.
import ctypes as _C

import numpy as _np

PHASE_FWD, PHASE_SUMS, PHASE_BWD, PHASE_GLOBAL_SUMS = 1, 2, 4, 8
WARM_COST = float(os.environ.get("WDF_MLP_WARM_COST", 0.55))   # a warm-up step (no input Jacobian, no stores, no loss) in owned-step units:
#   swept on the reference shape (0.45 .. 0.9): 0.55 gives the shortest step (0.397 ms against 0.414 at 0.75)


def _column_bounds(T, K, W, cost=None):
    """Chunk boundaries [0, t1, ..., T] of one column for K chunks and a warm-up of W steps: chunk 0 has no warm-up, so it
    is longer by about the warm-up's cost and every wave of the column runs about the same number of steps (lengths are
    multiples of 16: the L that minimises the longer of the two)."""
    if K <= 1:
        return [0, T]
    cost = WARM_COST if cost is None else cost
    best = None
    Lc = int((T - cost * W) / K / 16.0) * 16
    for L in (Lc - 16, Lc, Lc + 16, Lc + 32):
        L = max(16, min(L, T // K // 16 * 16))                # (chunk 0 is never the shortest: L0 >= L)
        L0 = T - (K - 1) * L
        c = max(L0, L + cost * W)
        if best is None or c < best[0]:
            best = (c, L, L0)
    _, L, L0 = best
    return [0] + [L0 + i * L for i in range(K)]


def _column_cost(T, K, W, cost=None):
    cost = WARM_COST if cost is None else cost
    b = _column_bounds(T, K, W, cost)
    return max(b[1], (b[2] - b[1] + cost * W) if K > 1 else 0)


def plan_step_items(T, wcol_steps, n_items, min_chunk=32):
    """The work list of the chunked forward: every column (16 sequences) gets its OWN chunk count -- the smallest common
    bound tau on a wave's steps (owned + the warm-up's cost) such that the columns' chunk counts add up to at most n_items
    (bisection on tau; a column takes the fewest chunks that meet it).  wcol_steps: warm-up steps per column.
    -> int32 [n_items, 4] = {column, chunk, t0, t1}."""
    ncol = len(wcol_steps)
    kmax = max(1, (T - 16) // min_chunk)
    table = {}

    def chunks_for(w, tau):
        key = int(w)
        if key not in table:
            table[key] = [_column_cost(T, k, key) for k in range(1, kmax + 1)]
        for k, c in enumerate(table[key], start=1):
            if c <= tau:
                return k
        return None

    def total(tau):
        ks = [chunks_for(w, tau) for w in wcol_steps]
        return None if any(k is None for k in ks) else ks

    lo, hi = 16, T
    best = total(hi)
    while hi - lo > 4:
        mid = (lo + hi) // 2
        ks = total(mid)
        if ks is not None and sum(ks) <= n_items:
            hi, best = mid, ks
        else:
            lo = mid
    # the waves left over go, one at a time, to the column that is slowest now: the plan always has n_items items (the
    # launch grid of a captured step does not change when the plan does)
    import heapq
    heap = [(-table[int(w)][best[c] - 1], c) for c, w in enumerate(wcol_steps)]
    heapq.heapify(heap)
    left = int(n_items) - sum(best)
    while left > 0 and heap:
        _, c = heapq.heappop(heap)
        if best[c] < kmax:
            best[c] += 1
            left -= 1
            heapq.heappush(heap, (-table[int(wcol_steps[c])][best[c] - 1], c))
    items = []
    for c in range(ncol):
        b = _column_bounds(T, best[c], wcol_steps[c])
        items += [(c, k, b[k], b[k + 1]) for k in range(best[c])]
    return _np.asarray(items, dtype=_np.int32).reshape(-1, 4)


class MlpTrainStep:
    """The resident training step of the MLP-root pot clipper (include/wdf_hip.h: wdf_clipper_mlp_step).

        st = MlpTrainStep(x, r, target, w, hidden, n_layers, fs, C, skip=50, adam=binding.Adam(...))
        for epoch in ...: st.step()          # w and the Adam moments are updated in place; st.loss3 = {mse, esr, loss}

    x, r [B,T] (r None: static R), target [T,B]: the resident training set.  w: the flat weights (float32, device, updated
    in place).  Everything a step reads or writes lives in buffers that do not move: capture-safe (HIP graph)."""

    def __init__(self, x, r, target, w, hidden, n_layers, fs, C, R_static=None, skip=50, adam=None, n_global=None,
                 eps_energy=None, activation="tanh", n_items=None, wgrad_chunks=None, tol=4.0e-6, sums_allreduce=None,
                 grad_allreduce=None):
        binding.require_gpu()
        self.lib = binding.lib()
        f32 = binding._f32_dev
        self.x, self.r, self.target, self.w = f32(x, "x"), f32(r, "r"), f32(target, "target"), f32(w, "w")
        self.B, self.T = (int(v) for v in x.shape)
        B, T = self.B, self.T
        if tuple(target.shape) != (T, B):
            raise binding.WdfHipError(f"target: expected [{T}, {B}] (time-major), got {tuple(target.shape)}")
        self.hidden, self.n_layers, self.fs = int(hidden), int(n_layers), float(fs)
        self.act = {"tanh": 0, "relu": 1}[activation]
        if self.w.numel() != self.lib.wdf_mlp_weight_count(self.hidden, self.n_layers):
            raise binding.WdfHipError("weight count does not match the network")
        dev = x.device
        self.skip = int(skip)
        self.n_global = float(n_global if n_global is not None else B * (T - self.skip))
        self.eps_energy = float(_np.finfo(float).eps if eps_energy is None else eps_energy)
        self.adam, self.sums_allreduce, self.grad_allreduce = adam, sums_allreduce, grad_allreduce
        self.ncol = (B + 15) // 16
        Rv = float(R_static) if R_static is not None else 45.0e3
        self.theta2 = torch.tensor([Rv, float(C)], dtype=torch.float32, device=dev)
        Rc = 1.0 / (2.0 * float(C) * self.fs)
        if r is not None:
            self.p, self.lr = torch.empty_like(self.r), torch.empty_like(self.r)
            binding._check(self.lib.wdf_clipper_mlp_step_prepare(binding._ptr(self.r), binding._ptr(self.theta2), self.fs, B, T,
                                                                 binding._ptr(self.p), binding._ptr(self.lr), binding._stream()),
                           "wdf_clipper_mlp_step_prepare")
            rmax = self.r.amax(dim=1)
            pad = (-B) % 16
            if pad:
                rmax = torch.cat([rmax, rmax[-1:].expand(pad)])
            rcol = rmax.reshape(-1, 16).amax(dim=1).double().cpu().numpy()
        else:
            self.p = self.lr = None
            rcol = _np.full(self.ncol, Rv)
        # the first guess of a column's warm-up: the RC network's diode-off memory |1 - 2p| at its largest pot value, down
        # to ~1e-5 of the call-to-call change; the device controller takes it from there
        rho = _np.clip(_np.abs(1.0 - 2.0 * Rc / (rcol + Rc)), 1e-6, 1.0 - 1e-9)
        self.cold = int(-(-int(1.15 * _np.max(_np.ceil(_np.log(0.01 * tol) / _np.log(rho)))) // 16) * 16)
        self.cold = min(self.cold, T)
        self.w_max = max(2, min(self.cold // 16, 40))
        w0 = _np.clip(_np.ceil(_np.log(1.0e-5) / _np.log(rho) / 16.0), 2, self.w_max).astype(_np.int32)
        # one wave per SIMD (the forward is a dependent chain per wave; a second wave on a SIMD would just share its issue slots)
        self.n_items_max = int(n_items) if n_items else max(self.ncol, min(engine.N_SIMD, self.ncol * (T // 64)))
        self.wgrad_chunks = int(wgrad_chunks) if wgrad_chunks else max(1, min(2 * engine.N_SIMD // self.ncol, T // 64))
        self.tol = float(tol)
        self.y = torch.empty((T, B), dtype=torch.float32, device=dev)
        self.zstash, self.kappa = torch.empty_like(self.y), torch.empty_like(self.y)
        self.sums = torch.zeros(2, dtype=torch.float64, device=dev)
        self.gw = torch.zeros(self.w.numel(), dtype=torch.float32, device=dev)
        self.loss3 = torch.zeros(3, dtype=torch.float32, device=dev)
        self.gcoef = torch.zeros(2, dtype=torch.float32, device=dev)
        self.state = None
        self._install_plan(w0, reset=True)

    # -- plan ---------------------------------------------------------------------------------------------------
    def _geom(self):
        return (self.hidden, self.n_layers, self.B, self.T, self.n_items, self.wgrad_chunks)

    def _install_plan(self, wcol_units, reset, items=None):
        if items is None:
            items = plan_step_items(self.T, [16 * int(v) for v in wcol_units], self.n_items_max)
        n_items = int(items.shape[0])
        if self.state is None or n_items != getattr(self, "n_items", None):
            if self.state is not None and not reset:
                raise binding.WdfHipError("a re-plan must keep the number of work items")   # (unreachable: see replan)
            self.n_items = n_items
            nbytes = self.lib.wdf_clipper_mlp_step_state_bytes(*self._geom())
            if nbytes == 0:
                raise binding.WdfHipError(self.lib.wdf_last_error().decode())
            self.state = torch.zeros((nbytes,), dtype=torch.uint8, device=self.x.device)
        self.items = items
        arr = _np.ascontiguousarray(items, dtype=_np.int32)
        rc = self.lib.wdf_clipper_mlp_step_plan(binding._ptr(self.state), *self._geom(),
                                                arr.ctypes.data_as(_C.POINTER(_C.c_int32)), 1 if reset else 0,
                                                0 if wcol_units is None else int(max(wcol_units)), self.cold // 16, 1, self.w_max,
                                                self.tol, binding._stream())
        binding._check(rc, "wdf_clipper_mlp_step_plan")
        if reset:
            self.set_wcol(wcol_units)

    def set_wcol(self, wcol_units):
        arr = _np.ascontiguousarray(wcol_units, dtype=_np.int32)
        binding._check(self.lib.wdf_clipper_mlp_step_set_wcol(binding._ptr(self.state), *self._geom(),
                                                              arr.ctypes.data_as(_C.POINTER(_C.c_int32)), binding._stream()),
                       "wdf_clipper_mlp_step_set_wcol")

    def read(self):
        """(controller dict, per-column warm-up units) -- synchronises."""
        ctl = (_C.c_int32 * 32)()
        wc = (_C.c_int32 * self.ncol)()
        binding._check(self.lib.wdf_clipper_mlp_step_read(binding._ptr(self.state), *self._geom(), ctl, wc, None, None, None, binding._stream()),
                       "wdf_clipper_mlp_step_read")
        c = _np.frombuffer(ctl, dtype=_np.int32).copy()
        last = c[16 + 4 * ((int(c[0]) - 1) & 1):][:4] if c[0] > 0 else c[16:20]
        d = {"calls": int(c[0]), "have_snap": int(c[2]), "cold_steps": 16 * int(c[3]),
             "n_bad": int(last[0]), "max_miss": float(last[1:2].view(_np.float32)[0]), "flagged_columns": int(last[2]),
             "sequential_columns": int(last[3]), "total_flagged": int(c[24]), "total_sequential": int(c[25])}
        return d, _np.frombuffer(wc, dtype=_np.int32).copy()

    def placement(self):
        """Where the dispatcher put the forward's waves in the last call: int array [n_items, 3] = (XCD, CU within the XCD
        (shader engine, array, CU), SIMD).  Diagnostics."""
        hw = (_C.c_int32 * (2 * self.n_items))()
        binding._check(self.lib.wdf_clipper_mlp_step_read(binding._ptr(self.state), *self._geom(), None, None, None, hw, None, binding._stream()),
                       "wdf_clipper_mlp_step_read")
        a = _np.frombuffer(hw, dtype=_np.uint32).reshape(-1, 2)
        hid, xcc = a[:, 0], a[:, 1] & 0xf
        simd, cu = (hid >> 4) & 3, (hid >> 8) & 0xff           # cu_id [11:8], sh_id [12], se_id [15:13]
        return _np.stack([xcc, cu, simd], axis=1).astype(_np.int64)

    def column_misses(self):
        """float [ncol, 4]: per column the last verification's arrival miss and the misses 16, 32, 48 steps before arrival."""
        cm = (_C.c_float * (4 * self.ncol))()
        binding._check(self.lib.wdf_clipper_mlp_step_read(binding._ptr(self.state), *self._geom(), None, None, None, None, cm, binding._stream()),
                       "wdf_clipper_mlp_step_read")
        return _np.frombuffer(cm, dtype=_np.float32).reshape(-1, 4).copy()

    def warmup_peaks(self):
        wp = (_C.c_int32 * self.ncol)()
        binding._check(self.lib.wdf_clipper_mlp_step_read(binding._ptr(self.state), *self._geom(), None, None, wp, None, None, binding._stream()),
                       "wdf_clipper_mlp_step_read")
        return _np.frombuffer(wp, dtype=_np.int32).copy()

    def replan(self, by=None):
        """Re-distribute the chunks over the columns with the warm-ups the controller has been running (host round trip;
        the number of work items -- the captured grid -- stays).  by: "peak" (default: the largest warm-up of each column
        since the last plan: the plan is balanced for the calls in which the weights swing) or "current"."""
        by = by or os.environ.get("WDF_MLP_PLAN_BY", "peak")
        _, wc = self.read()
        wp = _np.maximum(self.warmup_peaks(), wc) if by == "peak" else wc + 1
        items = plan_step_items(self.T, [16 * int(v) for v in wp], self.n_items_max)
        if int(items.shape[0]) != self.n_items:
            return False
        self._install_plan(wp, reset=False)
        return True

    def autotune(self, calls=16, verbose=False):
        """Pick the chunk plan by MEASURING: a handful of candidate plans (the columns' peak / current warm-ups, with and
        without head room, three values of the planner's warm-up cost) are each run for `calls` training steps -- one
        swing of the weights under Adam(beta_1 0.5) -- and the one with the shortest mean step is installed.  The steps are
        real training steps (part of a loop's warm-up).  -> (label, ms per step) of the winner."""
        global WARM_COST
        e0, e1 = binding.Event(), binding.Event()
        cost0 = WARM_COST
        results = []
        _, wc = self.read()
        wp = _np.maximum(self.warmup_peaks(), wc)
        cands = [("peak", wp, cost0), ("peak+1", wp + 1, cost0), ("current+1", wc + 1, cost0),
                 ("peak, cost 0.45", wp, 0.45), ("peak, cost 0.65", wp, 0.65), ("peak+2, cost 0.45", wp + 2, 0.45)]
        for label, wprof, cost in cands:
            WARM_COST = cost
            items = plan_step_items(self.T, [16 * int(v) for v in wprof], self.n_items_max)
            if int(items.shape[0]) != self.n_items:
                continue
            self._install_plan(wprof, reset=False, items=items)
            e0.record()
            for _ in range(calls):
                self.step()
            e1.record()
            torch.cuda.synchronize()
            results.append((e0.elapsed_ms(e1) / calls, label, items))
            if verbose:
                print(f"  plan {label}: {results[-1][0]:.4f} ms per step")
        WARM_COST = cost0
        if not results:
            return None
        ms, label, items = min(results, key=lambda r_: r_[0])
        self._install_plan(None, reset=False, items=items)
        return label, ms

    def freeze(self, on=True):
        binding._check(self.lib.wdf_clipper_mlp_step_set(binding._ptr(self.state), 12, 1 if on else 0, binding._stream()),
                       "wdf_clipper_mlp_step_set")

    # -- the step -------------------------------------------------------------------------------------------------
    def _call(self, phase, adam):
        a = adam
        rc = self.lib.wdf_clipper_mlp_step(
            binding._ptr(self.x), binding._ptr(self.p), binding._ptr(self.lr), binding._ptr(self.theta2), binding._ptr(self.w),
            self.hidden, self.n_layers, self.act, self.fs, binding._ptr(self.target), self.skip, self.n_global, self.eps_energy,
            binding._ptr(self.y), binding._ptr(self.zstash), binding._ptr(self.kappa), binding._ptr(self.state), self.B, self.T,
            self.n_items, self.wgrad_chunks, int(phase), binding._ptr(self.sums), binding._ptr(self.gw), binding._ptr(self.loss3),
            binding._ptr(self.gcoef),
            None if a is None else binding._ptr(a.m), None if a is None else binding._ptr(a.v),
            None if a is None else binding._ptr(a.step), None if a is None else binding._ptr(a.lr),
            0.0 if a is None else a.b1, 0.0 if a is None else a.b2, 0.0 if a is None else a.eps, binding._stream())
        binding._check(rc, "wdf_clipper_mlp_step")

    def step(self):
        """One training step.  Single rank: five launches, Adam inside.  With all-reduce hooks (data-parallel ranks):
        forward -> the rank's two loss sums -> all-reduce -> reverse sweep with the global sums -> all-reduce of the
        gradient -> Adam."""
        if self.sums_allreduce is None and self.grad_allreduce is None:
            self._call(PHASE_FWD | PHASE_BWD, self.adam)
            return self.loss3
        from . import dist as wdist

        def fwd():
            self._call(PHASE_FWD | PHASE_SUMS, None)
            return self.sums

        def bwd(_sums):
            self._call(PHASE_BWD | PHASE_GLOBAL_SUMS, None)
            return self.gw

        ar_s, ar_g = self.sums_allreduce or (lambda t: t), self.grad_allreduce or (lambda t: t)
        wdist.esr_two_exchange(fwd, bwd, allreduce=lambda t: (ar_s if t is self.sums else ar_g)(t))
        if self.adam is not None:
            self.adam.apply(self.w, self.gw)
        return self.loss3

    def forward_only(self):
        self._call(PHASE_FWD, None)

    def backward_only(self):
        self._call(PHASE_BWD, self.adam)


# ------------------------------------------------------------------------------ the resident step behind the element API
class MlpResident:
    """tf_wdf.Circuit(P1, DenseRootModel, C, per_sample_R=Vs).to_device(): the network's kernels and biases move into ONE
    flat device vector (each Variable becomes a view of its slice: same object, same shape, same autograd leaf), and
    circ.mse_esr(x, target, skip) -- the loss of clipper_pot.py:141-177,248 -- is one resident training step
    (MlpTrainStep, csrc/wdf_mlp_step.h) whose weight gradient tape.gradient hands out; tf.keras.optimizers.Adam updates
    the flat vector with one launch (compat_tf._Adam).  The loop of clipper_pot.py:245-269 then runs at the speed of
    `bench.py --root mlp2x16`, written the way the script writes it."""

    def __init__(self, circ, device):
        self.circ = circ
        self.dense, self.hidden, self.n_layers, self.act = describe(circ.root, with_activation=True)
        cap = circ.top.P2
        if getattr(cap.C, "requires_grad", False):
            raise binding.WdfHipError("Circuit.to_device with a DenseRootModel root trains the network's weights; the capacitor must "
                                      "not be trainable (clipper_pot.py:96-101)")
        self.fs, self.C = float(cap.FS), float(cap.C)
        vs = circ.top.P1
        self.R_static = None if circ.per_sample_R is not None else float(vs.R)
        flat = flat_weights(self.dense).detach().float().to(device).contiguous()
        self.w = flat
        self.vars, o = [], 0
        for d in self.dense:
            for v in (d.kernel, d.bias):
                n = v.numel()
                if getattr(v, "_wdf_flat", None) is not None:
                    raise binding.WdfHipError("Circuit.to_device: this network's weights already live in another circuit's vector")
                with torch.no_grad():
                    v.data = self.w[o:o + n].view(v.shape)
                v._wdf_flat = (self, o, n)
                self.vars.append(v)
                o += n
        self.spec = {id(v): v._wdf_flat[1:] + (tuple(v.shape),) for v in self.vars}
        from . import lowering
        self.cache = lowering.EntryCache()

    def entry(self, x, target, skip):
        from . import lowering
        with torch._C.DisableTorchFunctionSubclass():
            key = (lowering.tensor_key(x), lowering.tensor_key(target), int(skip))
        ent = self.cache.get(key)
        if ent is None:
            circ, dev = self.circ, self.w.device
            xd = x.as_subclass(torch.Tensor).to(dev).float()
            xv, r = engine.split_channels(xd, circ.per_sample_R is not None, anchor=x)
            B, T = xv.shape
            tgt = target.as_subclass(torch.Tensor).to(dev).float().reshape(T, B).contiguous()
            st = MlpTrainStep(xv, r, tgt, self.w, self.hidden, self.n_layers, self.fs, self.C, R_static=self.R_static,
                              skip=int(skip), adam=None, activation=self.act)
            ent = self.cache.put(key, {"st": st, "hold": (x, target), "calls": 0}, 8 * tgt.numel() * 4)
        return ent


class MlpResidentFn(torch.autograd.Function):
    """loss = MSE + ESR of the resident MLP-root clipper; the step that produced it produced d loss / d weights."""

    @staticmethod
    def forward(ctx, res, ent, *variables):
        st = ent["st"]
        st.step()
        ent["calls"] += 1
        if ent["calls"] == 24:                                   # the controller has settled: one measured re-plan
            st.replan()
        out = torch.cat((st.loss3[2:3], st.gw))                  # (a validation pass may run before backward)
        ctx.save_for_backward(out)
        ctx.spec = [res.spec[id(v)] for v in variables]
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gl, _):
        (out,) = ctx.saved_tensors
        return (None, None) + tuple((gl * out[1 + o:1 + o + n]).view(shape) for o, n, shape in ctx.spec)
